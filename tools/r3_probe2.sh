# Round-3 second probe (GPU box): LDS-DMA semantics, parity of the ring form, A/B against the register pipeline
R=$GRAFT_REPO_ROOT
TAG=${1:-r3p2}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout -k 5 60 tools/bin/glds_probe > $O/glds.txt 2>&1; rc=$?; cat $O/glds.txt; [ $rc -ge 124 ] && exit 1
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/pytest_parity.txt 2>&1; rc=$?; tail -15 $O/pytest_parity.txt; [ $rc -ge 124 ] && exit 1
L=real-time-neural-rendering-of-lidar-point-clouds_amd/lib
timeout -k 10 300 python tools/ab_frame.py $L/librtr_hip_reg.so $L/librtr_hip.so $L/librtr_hip_d3.so > $O/ab.txt 2>&1; rc=$?; cat $O/ab.txt; [ $rc -ge 124 ] && exit 1
RTR_LIB_VARIANT=xp timeout -k 10 240 python tools/kbench.py --scenes room_shell --frames 24 \
  --options "xp=0;xp=128;xp=256;xp=64;xp=8;xp=4;xp=0" > $O/xp.jsonl 2> $O/xp.err; rc=$?
python3 -c "
import sys,json
for l in open('$O/xp.jsonl'):
    d=json.loads(l); print(d['opts'].ljust(16), 'T1', d['min_depth'], 'tile', d['tile'], 'filter', d['filter'], 'probe', d['probe'])
"
