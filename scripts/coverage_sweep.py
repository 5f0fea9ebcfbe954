#!/usr/bin/env python3
"""frames/s of the cfg3 bench at several coverages (share of the frame the head covers): the sparse objective skips empty
regions, so its speed is a function of coverage.  The synthetic rig's default framing (head = 60 % of the image height) covers
11.5 % of a 1080p frame; the reference's own rig (f ~ 9.8 deg FOV, head filling a 1600 x 1200 frame) is the
40 % + case.   python scripts/coverage_sweep.py > profiles/rNN_coverage_sweep.json"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
points = []
for fill in (0.6, 1.37, 2.45):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--fill", str(fill), "--steps", "10", "--warmup", "3", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=600)
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if not line:
        points.append({"fill": fill, "error": out.stderr[-500:]})
        continue
    d = json.loads(line[-1])
    pt = {"fill": fill, "coverage": d["config"]["coverage"], "frames_per_s": d["value"], "ms_per_step": d["ms_per_step"],
          "occupied_bins": d["config"].get("occupied_bins"),
          "kernels_ms": {k: v["avg_ms"] for k, v in d.get("kernels", {}).items()},
          "roofline": {k: d["roofline"].get(k) for k in ("kernel", "achieved", "frac", "algorithmic_bytes", "dense_equivalent_GBps")} if "roofline" in d else None,
          "drop_in_path": d.get("drop_in_path"),
          "kernels_standalone_ops_ms": {k: v["avg_ms"] for k, v in d.get("kernels_standalone_ops", {}).items()} if isinstance(d.get("kernels_standalone_ops"), dict) and "error" not in d.get("kernels_standalone_ops", {}) else None}
    points.append(pt)
    print(f"fill {fill}: coverage {pt['coverage']:.3f}  {pt['frames_per_s']:.0f} frames/s  {pt['ms_per_step']:.2f} ms/step", file=sys.stderr)
print(json.dumps({"workload": "cfg3: 9-view 1920x1080, T=30000, K=150, 32 frames per step, one MI355X", "points": points}, indent=1))
