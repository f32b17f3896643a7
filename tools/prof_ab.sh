# per-kernel trace of tools/ab_frame.py for the given library files (GPU box)
set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/ab
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ab/trace -- python $R/tools/ab_frame.py "$@" > $R/gpurun_out/ab/out.txt 2>/dev/null
grep round $R/gpurun_out/ab/out.txt
python3 - <<PY
import csv, glob
for f in glob.glob("$R/gpurun_out/ab/trace/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows[:16]:
        print(r["Name"][:70].ljust(70), r["Calls"].rjust(5), "avg_us", round(float(r["AverageNs"])/1e3, 1))
PY
