# Round profile (GPU box): the bench line, a rocprofv3 kernel trace and the PMC passes of the same command.
# usage: bash tools/profile_round.sh <tag>      -> gpurun_out/<tag>/   (then tools/summarize_profile.py fills profiles/)
R=$GRAFT_REPO_ROOT
TAG=${1:-r03}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R && timeout -k 10 420 python bench.py > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 20 --warmup 3 --no-cpu-baseline --no-extra --no-parity"
run() {  # name, rocprofv3 options, extra bench args
  timeout -k 10 150 rocprofv3 $2 --output-format csv -d $O/$1 -- python $R/bench.py $ARGS $3 > $O/bench_$1.json 2> $O/err_$1.txt || { echo "pass $1 failed"; return 1; }
}
run trace "--kernel-trace --stats" "" &&
run pmc_f "--pmc FETCH_SIZE" "" &&
run pmc_w "--pmc WRITE_SIZE" "" &&
run pmc_f0 "--pmc FETCH_SIZE" "--set pack=0" &&
run pmc_w0 "--pmc WRITE_SIZE" "--set pack=0" &&
run pmc_i1 "--pmc VALUBusy SALUBusy MemUnitStalled" "" &&
run pmc_sq1 "--pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "" &&
run pmc_sq2 "--pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "" &&
run trace_ubox "--kernel-trace --stats" "--scene uniform_box" &&
run trace_fp32 "--kernel-trace --stats" "--set pack=0"
python3 $R/tools/summarize_profile.py $O
tail -c 400 $O/bench.json
