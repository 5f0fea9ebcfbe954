#!/usr/bin/env python3
"""Timeline of ONE fit step from a rocprofv3 kernel trace of scripts/prof_objective.py:
   rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 scripts/prof_objective.py --ops 0 ;  python scripts/step_timeline.py DIR
Prints every dispatch of the last complete step (k_setup to k_setup) with its start offset, duration, queue and the idle
gap before it, so that launch gaps and what overlaps with the two objective calls become visible."""
import csv, glob, os, sys
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:100], r.get("Queue_Id", "?")))
rows.sort()
starts = [i for i, r in enumerate(rows) if r[2] == "k_setup" or r[2].startswith("void k_setup<")]
a, b = starts[-2], starts[-1]
t0 = rows[a][0]
busy_end = t0
print("step: %.3f ms between two k_setup launches, %d dispatches" % ((rows[b][0] - t0) / 1e6, b - a))
for s, e, n, q in rows[a:b]:
    gap = (s - busy_end) / 1e3
    if (e - s) > 3000 or gap > 5 or "--all" in sys.argv:
        print("%9.1f us  +%8.1f us  gap %7.1f  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, q, n))
    busy_end = max(busy_end, e)
