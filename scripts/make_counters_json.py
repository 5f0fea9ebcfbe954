#!/usr/bin/env python3
"""rocprofv3 --pmc passes (counter_collection.csv + kernel_trace.csv) -> one JSON: per kernel, the mean of every counter per
dispatch and the mean duration in that pass.   python scripts/make_counters_json.py DIR... > counters.json
Dispatches up to and including the first k_objective_finish (the last kernel of the one-pass objective; k_render_aa_bwd for the two-call form) are ignored (set-up, the targets rendered in small chunks, and the first
fit step, whose list kernels run without launch hints at the full grid); of the rest, only dispatches whose grid lies within a factor two of the
median grid of each kernel are averaged (the list kernels' grids follow the hints from step to step)."""
import csv, glob, json, os, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_source_sha16      # the hash bench.py compares before it pairs these counters with its own timings

# git_head: the commit the passes ran on (the GPU box has no .git: scripts/measure_round.sh passes git_head=... through)
META = {"workload": "cfg3", "images": 288, "channels": 1, "git_head": os.environ.get("FPCDR_GIT_HEAD", "unknown"),
        "kernel_source_sha16": kernel_source_sha16()}
ARGS = [a for a in sys.argv[1:] if "=" not in a]
for a in sys.argv[1:]:
    if "=" in a:
        k, v = a.split("=", 1)
        META[k] = int(v) if v.isdigit() else v


def short(name):
    return name.replace("(anonymous namespace)::", "").split("(")[0]


out = defaultdict(lambda: {"counters": {}, "duration_us": {}})
for d in ARGS:
    tag = os.path.basename(d.rstrip("/"))
    dur = defaultdict(list)
    first = None    # dispatch id of the first backward call
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_render_aa_bwd" in r["Kernel_Name"] or "k_objective_finish" in r["Kernel_Name"]:      # (two-call form / one-pass form)
                i = int(r["Dispatch_Id"])
                first = i if first is None else min(first, i)
    first = first or 0
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if int(r["Dispatch_Id"]) <= first:
                continue
            g = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0) * max(int(r.get("Grid_Size_Y", 1) or 1), 1) * max(int(r.get("Grid_Size_Z", 1) or 1), 1)
            dur[short(r["Kernel_Name"])].append((g, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    def steady(v):
        """values of the dispatches whose grid lies within a factor two of the MEDIAN grid (a list kernel's first launch of a run has
        no hint yet and dispatches one workgroup per bin of the batch)"""
        gs = sorted(g for g, _ in v)
        gmed = gs[len(gs) // 2]
        return [t for g, t in v if g <= 2 * gmed and 2 * g >= gmed]

    for k, v in dur.items():
        sel = steady(v)
        out[k]["duration_us"][tag] = sum(sel) / len(sel)
        out[k]["dispatches"] = len(sel)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        vals = defaultdict(lambda: defaultdict(list))
        for r in csv.DictReader(open(f)):
            if int(r["Dispatch_Id"]) <= first:
                continue
            g = int(r.get("Grid_Size", 0) or 0)
            vals[short(r["Kernel_Name"])][r["Counter_Name"]].append((g, float(r["Counter_Value"])))
        for k, cs in vals.items():
            for c, v in cs.items():
                sel = steady(v)
                out[k]["counters"][c] = sum(sel) / len(sel)
keep = {k: v for k, v in out.items() if k.startswith(("k_", "void k_"))}
print(json.dumps(dict(META, _comment="rocprofv3 --pmc passes of scripts/prof_objective.py (scripts/measure_round.sh): per kernel the mean "
                       "counter value per dispatch (after the first fit step; dispatches of at least half the largest grid) and the mean duration in each pass; FETCH_SIZE / "
                       "WRITE_SIZE in KB", kernels=keep), indent=1, sort_keys=True))
