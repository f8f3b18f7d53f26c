#!/usr/bin/env python3
"""Last N kernels of a rocprofv3 kernel trace CSV as a timeline (start, duration, queue, name): what overlaps what.  usage: trace_tail.py <kernel_trace.csv> [N]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f us  +%7.1f us  q%-3s grid %-7s %s" % ((s - t0) * 1e-3, (e - s) * 1e-3, r.get("Queue_Id", "?"), r.get("Grid_Size") or r.get("Grid_Size_X"), r["Kernel_Name"].split("(")[0].replace("hb::", "")))
