set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/p2
cd $R && timeout -k 10 400 python bench.py > gpurun_out/p2/bench.json 2> gpurun_out/p2/bench.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/p2/trace -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra > $R/gpurun_out/p2/bench_prof.json 2>/dev/null
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/p2/pmc_f -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra --no-parity > /dev/null 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/p2/pmc_w -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra --no-parity > /dev/null 2>&1
cat $R/gpurun_out/p2/bench.json
