# Round profile (GPU box): bench line, rocprofv3 kernel trace + the two PMC passes of the same command.
# usage: bash tools/profile_round.sh <tag>      -> gpurun_out/<tag>/
set -e
R=$GRAFT_REPO_ROOT
TAG=${1:-p2}
mkdir -p $R/gpurun_out/$TAG
cd $R && timeout -k 10 300 python bench.py > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 20 --warmup 3 --no-cpu-baseline --no-extra --no-parity"
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/trace -- python $R/bench.py $ARGS > $R/gpurun_out/$TAG/bench_prof.json 2>/dev/null
timeout -k 10 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/$TAG/pmc_f -- python $R/bench.py $ARGS > /dev/null 2>&1
timeout -k 10 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/$TAG/pmc_w -- python $R/bench.py $ARGS > /dev/null 2>&1
# instruction-issue counters of the same command (the packed point kernel is issue-bound, not HBM-bound)
timeout -k 10 120 rocprofv3 --pmc VALUBusy SALUBusy MemUnitStalled --output-format csv -d $R/gpurun_out/$TAG/pmc_i1 -- python $R/bench.py $ARGS > /dev/null 2>&1
timeout -k 10 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM --output-format csv -d $R/gpurun_out/$TAG/pmc_i2 -- python $R/bench.py $ARGS > /dev/null 2>&1
# the incoherent scene as uploaded (auto_reorder off is what bench's uniform_box.as_uploaded leg uses)
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/trace_ubox -- python $R/bench.py $ARGS --scene uniform_box > $R/gpurun_out/$TAG/bench_prof_ubox.json 2>/dev/null
python3 $R/tools/summarize_profile.py $R/gpurun_out/$TAG
cat $R/gpurun_out/$TAG/bench.json | tail -c 600
