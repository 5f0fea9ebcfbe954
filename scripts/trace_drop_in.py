#!/usr/bin/env python3
"""Three fit steps at cfg3 through the DROP-IN surface only (the four nvdiffrast-style operators + the reference's torch loss),
for a kernel trace of what a user who only switches the import runs:
    rocprofv3 --kernel-trace --stats --output-format csv -d DIR -- python3 scripts/trace_drop_in.py ; python scripts/summarize_rocprof.py DIR"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from fpc_diffrend_amd import fit, scene  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 32
sc = scene.cfg("cfg3", n_frames=frames)
cfg = fit.FitConfig(max_iter=80000, frames_per_step=0, init_texture="random", fused_objective=False, fused_render=False, fused_loss=False)
ft = fit.Fitter(sc, cfg, device="cuda")
for _ in range(3):
    ft.step()
torch.cuda.synchronize()
