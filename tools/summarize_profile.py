"""Condenses a tools/profile_round.sh output directory into the small files kept under profiles/:
kernel_stats.csv (name, calls, avg/min/max us) and pmc_fetch_write_summary.json (per kernel: FETCH_SIZE
and WRITE_SIZE in KiB per dispatch, and the corrected HBM bytes 1024 * (2 * FETCH_SIZE + WRITE_SIZE))."""
import csv, glob, json, os, sys

d = sys.argv[1]


def short(name):
    name = name.replace("void ", "").replace("rtr::", "")
    return name.split("(")[0]


def kernel_stats(sub, out):
    rows = []
    for f in glob.glob(os.path.join(d, sub, "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((short(r["Name"]), int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3,
                         float(r["MaxNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3))
    with open(os.path.join(d, out), "w") as f:
        f.write("kernel,calls,avg_us,min_us,max_us,total_us\n")
        for r in rows:
            f.write("%s,%d,%.1f,%.1f,%.1f,%.1f\n" % r)
    return rows


def pmc(sub, counter):
    acc = {}
    for f in glob.glob(os.path.join(d, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            k = short(r["Kernel_Name"])
            a = acc.setdefault(k, [0.0, 0])
            a[0] += float(r["Counter_Value"])
            a[1] += 1
    return {k: v[0] / max(v[1], 1) for k, v in acc.items()}


rows = kernel_stats("trace", "kernel_stats.csv")
if os.path.isdir(os.path.join(d, "trace_ubox")):
    kernel_stats("trace_ubox", "kernel_stats_uniform_box.csv")
if os.path.isdir(os.path.join(d, "trace_fp32")):
    kernel_stats("trace_fp32", "kernel_stats_fp32_soa.csv")


def traffic(fdir, wdir, out):
    fetch, write = pmc(fdir, "FETCH_SIZE"), pmc(wdir, "WRITE_SIZE")
    summ = {}
    for k in sorted(set(fetch) | set(write)):
        f_, w_ = fetch.get(k, 0.0), write.get(k, 0.0)
        summ[k] = {"FETCH_SIZE_KiB_per_dispatch": f_, "WRITE_SIZE_KiB_per_dispatch": w_,
                   "hbm_bytes_per_dispatch": 1024.0 * (2.0 * f_ + w_)}
    json.dump(summ, open(os.path.join(d, out), "w"), indent=1)
    return summ


summ = traffic("pmc_f", "pmc_w", "pmc_fetch_write_summary.json")
if os.path.isdir(os.path.join(d, "pmc_f0")):
    summ0 = traffic("pmc_f0", "pmc_w0", "pmc_fetch_write_summary_fp32_soa.json")
    for k, v in summ0.items():
        if k.startswith("k_project_bin"):
            print("fp32 SoA (pack = 0):", k, round(v["hbm_bytes_per_dispatch"] / 1e9, 4), "GB")
# every other counter pass (pmc_i*, pmc_sq*), per kernel and dispatch
issue = {}
for sub in sorted(os.listdir(d)):
    if not (sub.startswith("pmc_i") or sub.startswith("pmc_sq")):
        continue
    for f in glob.glob(os.path.join(d, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if not k.startswith(("k_project_bin", "k_tile", "k_filter4")):
                continue
            a = issue.setdefault(k, {}).setdefault(r["Counter_Name"], [0.0, 0])
            a[0] += float(r["Counter_Value"])
            a[1] += 1
if issue:
    out = {k: {c: v[0] / max(v[1], 1) for c, v in cs.items()} for k, cs in issue.items()}
    for k, cs in out.items():  # where a wave's cycles go (the three are disjoint: MI355X_MICROARCH.md, PMC section)
        wc = cs.get("SQ_WAVE_CYCLES")
        if wc:
            cs["share_waiting_on_memory_or_barriers"] = cs.get("SQ_WAIT_ANY", 0.0) / wc
            cs["share_waiting_to_issue"] = cs.get("SQ_WAIT_INST_ANY", 0.0) / wc
            cs["share_issuing"] = cs.get("SQ_ACTIVE_INST_ANY", 0.0) / wc
    json.dump(out, open(os.path.join(d, "pmc_issue_summary.json"), "w"), indent=1)
frame = [r for r in rows if r[0].startswith(("k_project_bin", "k_tile", "k_filter4"))]
print("frame kernels:", [(r[0], round(r[2], 1)) for r in frame], "sum_us", round(sum(r[2] for r in frame), 1))
print("frame traffic GB:", round(sum(v["hbm_bytes_per_dispatch"] for k, v in summ.items()
                                     if k.startswith(("k_project_bin", "k_tile", "k_filter4"))) / 1e9, 3))
for k, v in summ.items():
    if k.startswith(("k_project_bin", "k_tile", "k_filter4")):
        print("  ", k, round(v["hbm_bytes_per_dispatch"] / 1e9, 4), "GB")
