"""Prints the kernel timeline (start offset / duration in us) of a few frames from a rocprofv3 kernel trace."""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "").replace("rtr::", "").split("(")[0], r.get("Stream_Id", r.get("Queue_Id", "?"))))
rows.sort()
sel = [r for r in rows if r[2].startswith(("k_project_bin", "k_tile", "k_filter4"))]
mid = len(sel) // 2
t0 = sel[mid][0]
for s, e, n, q in sel[mid:mid + 14]:
    print("%9.1f %9.1f  dur %7.1f  q=%s  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, n))
