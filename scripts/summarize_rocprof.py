#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel_stats / kernel_trace / counter_collection) into a small text summary
that is committed under profiles/.   python scripts/summarize_rocprof.py <rocprof_out_dir> [more dirs] > profiles/x.txt"""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    return name.split("(")[0][:70]


for d in sys.argv[1:]:
    print(f"== {d}")
    for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
        rows = list(csv.DictReader(open(f)))
        print(f"-- kernel stats ({os.path.basename(f)}): name, calls, total_ms, avg_us, pct")
        for r in rows[:90]:
            print(f"{short(r['Name']):72s} {int(r['Calls']):6d} {float(r['TotalDurationNs']) / 1e6:10.3f} "
                  f"{float(r['AverageNs']) / 1e3:10.1f} {float(r['Percentage']):6.2f}")
    # list kernels are launched at the full grid until the launch hints of the first call have landed (one or two of a run's first
    # steps), afterwards at the hinted size: their mean over ALL launches above mixes the two (a dispatch-bound 587 k-workgroup launch
    # can take ten times as long).  The steady state -- what bench.py's timed region sees -- is the mean over the hinted launches.
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        per = defaultdict(list)
        for r in csv.DictReader(open(f)):
            g = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0) * max(int(r.get("Grid_Size_Y", 1) or 1), 1) * max(int(r.get("Grid_Size_Z", 1) or 1), 1)
            per[short(r["Kernel_Name"])].append((g, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
        rows = []
        for k, v in per.items():
            gmax = max(g for g, _ in v)
            small = [t for g, t in v if 2 * g < gmax]
            if small and ("_list<" in k or "_list" in k):
                rows.append((sum(small), k, len(small), sum(small) / len(small), len(v) - len(small)))
        if rows:
            print(f"-- list kernels, HINTED launches only ({os.path.basename(f)}): name, hinted calls, avg_us, (full-grid calls left out)")
            for _, k, n, avg, rest in sorted(rows, reverse=True):
                print(f"{k:72s} {n:6d} {avg:10.1f}   ({rest})")
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        rows = list(csv.DictReader(open(f)))
        agg = defaultdict(lambda: defaultdict(list))
        for r in rows:
            agg[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
        print(f"-- counters ({os.path.basename(f)}): kernel, counter, dispatches, mean value per dispatch")
        for k, cs in sorted(agg.items()):
            if not any(x in k for x in ("k_", "Cijk")):
                continue
            for c, v in cs.items():
                print(f"{k:72s} {c:14s} {len(v):5d} {sum(v) / len(v):16.1f} (max {max(v):.1f})")
