R=$GRAFT_REPO_ROOT
TAG=${1:-r3p4}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/pytest_parity.txt 2>&1; rc=$?; tail -8 $O/pytest_parity.txt; [ $rc -ge 124 ] && exit 1
L=real-time-neural-rendering-of-lidar-point-clouds_amd/lib
timeout -k 10 300 python tools/ab_frame.py $L/librtr_hip_reg.so $L/librtr_hip.so > $O/ab.txt 2>&1; rc=$?; cat $O/ab.txt; [ $rc -ge 124 ] && exit 1
timeout -k 10 120 python tools/c2_probe.py "" "overlap=1" "overlap=0,pack=0" > $O/c2.txt 2>&1; rc=$?; cat $O/c2.txt; [ $rc -ge 124 ] && exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c2 -- python $R/tools/c2_probe.py > $O/c2_prof.txt 2>/dev/null
python3 - <<PY
import csv, glob
for f in glob.glob("$O/trace_c2/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:10]:
        print(r["Name"][:70].ljust(70), r["Calls"].rjust(5), "avg_us", round(float(r["AverageNs"])/1e3, 1))
PY
cd $R
timeout -k 10 500 python bench.py > $O/bench.json 2> $O/bench.err; rc=$?; tail -c 800 $O/bench.err; [ $rc -ge 124 ] && exit 1
python3 - <<PY
import json
d = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
def show(name, o):
    if not o: print(name, o); return
    r = o.get("roofline") or {}
    print(name.ljust(28), "ms %.4f" % o["ms_per_step"], "Gpts/s %.1f" % (o["value"] / 1e3), "| T1 ms", r.get("avg_launch_ms") and round(r["avg_launch_ms"], 4), "B/pt", r.get("resident_stream_bytes_per_point"), "frac", r.get("frac") and round(r["frac"], 3), "src", r.get("bytes_source"), "vs_fp32", r.get("vs_fp32_stream") and round(r["vs_fp32_stream"], 3), "parity", o.get("parity_vs_oracle"))
show("headline", d); show("fp32_soa", d.get("fp32_soa")); show("c2", d.get("c2")); show("host_outputs", d.get("host_outputs"))
show("rotated", d.get("rotated_noisy_scene")); show("ubox as uploaded", (d.get("uniform_box") or {}).get("as_uploaded")); show("ubox default", (d.get("uniform_box") or {}).get("default_upload_policy"))
show("pipelined", d.get("pipelined")); show("cull", d.get("with_chunk_culling"))
PY
