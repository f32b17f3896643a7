cd $GRAFT_REPO_ROOT
for N in 2 4; do
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 2960$N bench.py --gpus $N --backend gloo --points 3000000 --steps 6 --warmup 2 --no-cpu-baseline 2>gpurun_out/mr$N.err
done
