"""Averages every counter of the rocprofv3 --pmc passes under <dir>/pmc_* per kernel and dispatch -> <dir>/pmc_summary.json."""
import collections, csv, glob, json, os, sys

d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(d, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void ", "").replace("rtr::", "").split("(")[0]
        if k.startswith(("k_project_bin", "k_tile", "k_filter4", "k_p2p", "k_min_depth", "k_accumulate")):
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: sum(v) / len(v) for c, v in sorted(cs.items())} for k, cs in acc.items()}
for k, cs in out.items():
    cs["dispatches_per_pass"] = len(next(iter(acc[k].values())))
json.dump(out, open(os.path.join(d, "pmc_summary.json"), "w"), indent=1)
for k, cs in out.items():
    print(k, json.dumps({c: round(v, 1) for c, v in cs.items()}))
