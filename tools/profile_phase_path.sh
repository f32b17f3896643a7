# kernel trace of the sharded frame's local path with ONE rank (force-exchange, p2p form), GPU box
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/p3
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/p3 -- python $R/bench.py --force-exchange --backend gloo --exchange p2p --steps 20 --warmup 3 --no-cpu-baseline --pipeline 1 --no-parity --no-extra > $R/gpurun_out/p3.json 2>/dev/null
python3 - <<PY
import csv, glob
tot = 0.0
for f in glob.glob("$R/gpurun_out/p3/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Name"].replace("void ", "").replace("rtr::", "").split("(")[0]
        if int(r["Calls"]) >= 20 and not n.startswith(("k_generate", "k_chunk")):
            per_frame = float(r["TotalDurationNs"]) / 1e3 / 24.0
            print(n.ljust(28), r["Calls"].rjust(5), "avg_us", round(float(r["AverageNs"]) / 1e3, 1), "per_frame_us", round(per_frame, 1))
            tot += per_frame
print("sum per frame (24 frames incl. the verification frame)", round(tot, 1))
PY
