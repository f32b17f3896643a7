R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r3sh}
mkdir -p $O
cd $R
timeout -k 10 200 python tools/sharded_probe.py 8 100 > $O/probe.txt 2>&1; rc=$?; cat $O/probe.txt; [ $rc -ge 124 ] && exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $R/tools/sharded_probe.py 8 30 > $O/probe_prof.txt 2>/dev/null
python3 - <<PY
import csv, glob
for f in glob.glob("$O/trace/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:16]:
        print(r["Name"].replace("void ","").replace("rtr::","")[:60].ljust(60), r["Calls"].rjust(5), "avg_us", round(float(r["AverageNs"])/1e3, 1))
PY
