# Round-3 first probe (GPU box): where does T1's time go?  (a) the RTR_EXPERIMENT build with parts of T1 switched
# off (option xp), (b) SQ busy / wait counters of the unmodified library.  usage: bash tools/r3_probe1.sh [tag]
R=$GRAFT_REPO_ROOT
TAG=${1:-r3p1}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
RTR_LIB_VARIANT=xp timeout -k 10 240 python tools/kbench.py --scenes room_shell --frames 24 \
  --options "xp=0;xp=128;xp=256;xp=64;xp=8;xp=4;xp=0;pack=0,xp=0;pack=0,xp=64;pack=0,xp=8;pack=0,xp=0" > $O/xp.jsonl 2> $O/xp.err || exit 1
cat $O/xp.jsonl
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 6 --warmup 2 --no-cpu-baseline --no-extra --no-parity"
pass() {  # name, counters
  timeout -k 10 150 rocprofv3 --pmc $2 --output-format csv -d $O/pmc_$1 -- python $R/bench.py $ARGS > $O/bench_$1.json 2> $O/err_$1.txt || return 1
}
pass a "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" &&
pass b "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_LEVEL_VMEM" &&
pass c "SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_THREAD_CYCLES_VALU SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL GRBM_GUI_ACTIVE" &&
pass d "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_IFETCH"
python3 $R/tools/pmc_summary.py $O
