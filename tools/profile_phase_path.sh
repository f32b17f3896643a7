R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/p3 -- python $R/bench.py --force-exchange --backend gloo --exchange p2p --steps 20 --warmup 3 --no-cpu-baseline --pipeline 1 --no-parity > $R/gpurun_out/p3.json 2>/dev/null
cd $R; python - <<'PY'
import csv,glob
f=sorted(glob.glob('gpurun_out/p3/*/*kernel_stats.csv'))[-1]
for r in csv.DictReader(open(f)):
    if int(r['Calls'])>=20: print(r['Name'][:50].ljust(52), r['Calls'], round(float(r['AverageNs'])/1e3,1))
PY
python -c "import json;d=json.loads(open('gpurun_out/p3.json').read().strip().splitlines()[-1]);print(d['ms_per_step'])"
