#!/usr/bin/env python3
"""Timeline summary of a rocprofv3 kernel trace CSV: per kernel name count / mean duration, and how much of the wall span had
0, 1, 2+ kernels in flight (do the two pipelined segments actually overlap?).  usage: trace_overlap.py <kernel_trace.csv> [last N rows]"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows)
rows = rows[-n:]
ev = []
per = collections.defaultdict(list)
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("hb::", "") + ":" + (r.get("Grid_Size") or r.get("Grid_Size_X"))
    per[name].append(e - s)
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
t0, t1 = ev[0][0], ev[-1][0]
depth, last, hist = 0, t0, collections.Counter()
for t, d in ev:
    hist[min(depth, 3)] += t - last
    last, depth = t, depth + d
print("span %.3f ms, %d kernels" % ((t1 - t0) * 1e-6, len(rows)))
for k in sorted(hist): print("  %s kernels in flight: %5.1f %%" % (("%d" % k) if k < 3 else "3+", 100.0 * hist[k] / (t1 - t0)))
for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
    print("  %-50s n %5d mean %8.1f us total %8.2f ms" % (k, len(v), 1e-3 * sum(v) / len(v), 1e-6 * sum(v)))
