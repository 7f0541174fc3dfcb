#!/usr/bin/env python3
"""Print the kernel timeline of the last bench step from a rocprofv3 kernel_trace.csv (any --mode: a step starts with the
encode call's counter fills followed by its first window-control / transform kernel, or - decode only - with the walk)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].replace("void ", "")
first = ("k_xf_spec", "k_wc_ef", "k_wc_energy")
starts = []
for i, r in enumerate(rows):
    if name(r).startswith(first) and (i == 0 or "fillBuffer" in rows[i - 1]["Kernel_Name"] or not name(rows[i - 1]).startswith("k_")):
        j = i
        while j > 0 and "fillBuffer" in rows[j - 1]["Kernel_Name"]: j -= 1
        starts.append(j)
if not starts:
    starts = [i for i, r in enumerate(rows) if name(r).startswith(("k_dscan", "k_dscan_packed"))]
idx = starts[-1] if starts else 0
t0 = int(rows[idx]["Start_Timestamp"]); prev_end = t0
for r in rows[idx:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-28s start %8.3f  dur %7.3f  gap %7.3f  q=%s" % (name(r)[:28], (s - t0) / 1e6, (e - s) / 1e6, (s - prev_end) / 1e6, r.get("Queue_Id", "?")))
    prev_end = max(prev_end, e)
