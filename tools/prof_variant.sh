# per-kernel times of a short bench run with a library variant (GPU box): bash tools/prof_variant.sh <tag> [bench args]
R=$GRAFT_REPO_ROOT
TAG=$1; shift
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $R/bench.py --steps 16 --warmup 3 --no-cpu-baseline --no-extra --no-parity "$@" > $O/bench_prof.json 2>/dev/null
python3 - <<PY
import csv, glob
for f in glob.glob("$O/trace/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:6]:
        print(r["Name"].replace("void ","").replace("rtr::","")[:56].ljust(56), r["Calls"].rjust(5), "avg_us", round(float(r["AverageNs"])/1e3, 1))
PY
