#!/usr/bin/env python3
"""Time every C-ABI entry point of one fit step at a chosen batch size (HIP events), for kernel tuning.
    python scripts/profile_ops.py [--frames 4] [--reps 5] [--workload cfg3]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from fpc_diffrend_amd import _lib, fit, scene  # noqa: E402
from bench import algorithmic_bytes_per_px  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=4)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--workload", default="cfg3")
ap.add_argument("--mip", action="store_true")
args = ap.parse_args()
sc = scene.cfg(args.workload, n_frames=args.frames)
cfg = fit.FitConfig(init_texture="random", enable_mip=args.mip)
ft = fit.Fitter(sc, cfg, device="cuda")
for _ in range(2):
    ft.step()
_lib.TIMER = _lib.KernelTimer()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(args.reps):
    ft.step()
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / args.reps
summ = _lib.TIMER.summary()
_lib.TIMER = None
H, W = ft.resolution
npix = args.frames * len(ft.cam_idxs) * H * W
bpp = algorithmic_bytes_per_px(sc.texture.shape[2], args.mip)
tot = 0.0
print(f"{args.workload}: {args.frames} frames x {len(ft.cam_idxs)} views, {npix/1e6:.1f} Mpx, wall {wall*1e3:.2f} ms/step")
for k, (n, ms) in sorted(summ.items(), key=lambda kv: -kv[1][1]):
    per = ms / n
    tot += per * n / args.reps
    gb = f"{bpp[k] * npix / per / 1e6:8.0f} GB/s" if k in bpp else ""
    print(f"  {k:28s} {per:9.3f} ms  {gb}")
print(f"  sum of fpcdr calls: {tot:.2f} ms/step")
