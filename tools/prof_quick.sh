# quick per-kernel trace of a short bench run (GPU box): tools/prof_quick.sh <tag> [bench args]
set -e
R=$GRAFT_REPO_ROOT
TAG=$1; shift
mkdir -p $R/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
timeout -k 10 90 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/trace -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra --no-parity "$@" > $R/gpurun_out/$TAG/bench_prof.json 2>/dev/null
python3 - <<PY
import csv, glob
for f in glob.glob("$R/gpurun_out/$TAG/trace/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows[:12]:
        print(r["Name"][:60].ljust(60), r["Calls"].rjust(5), "avg_us", round(float(r["AverageNs"])/1e3, 1), "min", round(float(r["MinNs"])/1e3,1), "max", round(float(r["MaxNs"])/1e3,1))
PY
