#!/usr/bin/env python3
"""Print the kernel timeline of the last bench step from a rocprofv3 kernel_trace.csv."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last step: find last k_wc_energy
ends = [i for i, r in enumerate(rows) if "k_dsyn" in r["Kernel_Name"] or "k_dgen" in r["Kernel_Name"]]
idx = ends[-2] + 1 if len(ends) >= 2 else 0           # first kernel after the previous step's last decode kernel
t0 = int(rows[idx]["Start_Timestamp"]); prev_end = t0
for r in rows[idx:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-28s start %8.3f  dur %7.3f  gap %7.3f  q=%s" % (r["Kernel_Name"][:28], (s - t0) / 1e6, (e - s) / 1e6, (s - prev_end) / 1e6, r.get("Queue_Id", "?")))
    prev_end = max(prev_end, e)
