# hardware counters of the point kernel (GPU box): tools/pmc_t1.sh <tag> "<counters>" [bench args]
set -e
R=$GRAFT_REPO_ROOT
TAG=$1; CTR=$2; shift; shift
mkdir -p $R/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
timeout -k 10 150 rocprofv3 --pmc $CTR --output-format csv -d $R/gpurun_out/$TAG/pmc -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra --no-parity "$@" > $R/gpurun_out/$TAG/bench.json 2>$R/gpurun_out/$TAG/err.txt
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$R/gpurun_out/$TAG/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:40]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if "project_bin" in k or "k_tile<0>" in k:
        print(k, {c: round(sum(v) / len(v), 2) for c, v in d.items()}, "n", len(next(iter(d.values()))))
PY
