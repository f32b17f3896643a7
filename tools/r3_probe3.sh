R=$GRAFT_REPO_ROOT
TAG=${1:-r3p3}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout -k 5 120 tools/bin/align_probe > $O/align.txt 2>&1; rc=$?; cat $O/align.txt; [ $rc -ge 124 ] && exit 1
timeout -k 10 600 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; rc=$?; tail -12 $O/pytest.txt; [ $rc -ge 124 ] && exit 1
timeout -k 10 500 python bench.py > $O/bench.json 2> $O/bench.err; rc=$?; tail -c 1500 $O/bench.err; [ $rc -ge 124 ] && exit 1
python3 - <<PY
import json
d = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
def show(name, o):
    if not o: print(name, o); return
    r = o.get("roofline") or {}
    print(name.ljust(28), "ms %.4f" % o["ms_per_step"], "Gpts/s %.1f" % (o["value"] / 1e3), "| T1 ms", r.get("avg_launch_ms"), "frac", r.get("frac") and round(r["frac"], 3), "src", r.get("bytes_source"), "vs_fp32", r.get("vs_fp32_stream") and round(r["vs_fp32_stream"], 3), "parity", o.get("parity_vs_oracle"))
show("headline", d); show("fp32_soa", d.get("fp32_soa")); show("c2", d.get("c2")); show("host_outputs", d.get("host_outputs"))
show("rotated", d.get("rotated_noisy_scene")); show("ubox as uploaded", (d.get("uniform_box") or {}).get("as_uploaded")); show("ubox default", (d.get("uniform_box") or {}).get("default_upload_policy"))
show("pipelined", d.get("pipelined")); show("cull", d.get("with_chunk_culling")); print("cpu", d.get("cpu_baseline")); print("stats", (d.get("roofline") or {}).get("frame_stats"))
PY
