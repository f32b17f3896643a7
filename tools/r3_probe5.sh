R=$GRAFT_REPO_ROOT
TAG=${1:-r3p5}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; rc=$?; tail -6 $O/pytest.txt; [ $rc -ge 124 ] && exit 1
L=real-time-neural-rendering-of-lidar-point-clouds_amd/lib
timeout -k 10 300 python tools/ab_frame.py $L/librtr_hip_prev.so $L/librtr_hip.so > $O/ab.txt 2>&1; rc=$?; cat $O/ab.txt; [ $rc -ge 124 ] && exit 1
timeout -k 10 120 python tools/c2_probe.py "" > $O/c2.txt 2>&1; cat $O/c2.txt
RTR_LIB_VARIANT=prev timeout -k 10 120 python tools/c2_probe.py "" > $O/c2_prev.txt 2>&1; cat $O/c2_prev.txt
RTR_LIB_VARIANT=xp timeout -k 10 100 python tools/stamps.py 2>&1 | tail -4
