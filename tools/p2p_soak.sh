cd $GRAFT_REPO_ROOT
for cfg in "3 640 480 400000 300 room_shell 1" "2 1920 1080 2000000 150 room_shell 1" "3 208 120 200000 300 uniform_box 1"; do
set -- $cfg
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $1 --master-addr 127.0.0.1 --master-port 2970$1 tests/p2p_worker.py $2 $3 $4 $5 $6 $7 2>gpurun_out/soak.err | tail -1 | cut -c1-600
done
