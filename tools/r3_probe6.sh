R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r3p6}
mkdir -p $O
cd $R
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/pytest_parity.txt 2>&1; rc=$?; tail -4 $O/pytest_parity.txt; [ $rc -ge 124 ] && exit 1
L=real-time-neural-rendering-of-lidar-point-clouds_amd/lib
timeout -k 10 300 python tools/ab_frame.py $L/librtr_hip_prev.so $L/librtr_hip.so > $O/ab.txt 2>&1; rc=$?; cat $O/ab.txt; [ $rc -ge 124 ] && exit 1
timeout -k 10 120 python tools/c2_probe.py "" > $O/c2.txt 2>&1; cat $O/c2.txt
RTR_LIB_VARIANT=prev timeout -k 10 120 python tools/c2_probe.py "" > $O/c2_prev.txt 2>&1; cat $O/c2_prev.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra --no-parity > $O/bench_prof.json 2>/dev/null
python3 - <<PY
import csv, glob
for f in glob.glob("$O/trace/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:8]:
        print(r["Name"].replace("void ","").replace("rtr::","")[:50].ljust(50), r["Calls"].rjust(5), "avg_us", round(float(r["AverageNs"])/1e3, 1))
PY
