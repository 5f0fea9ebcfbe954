"""Seven cfg3 fit steps with FitConfig.enable_mip for rocprofv3 --kernel-trace --stats (scripts/summarize_rocprof.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fpc_diffrend_amd import fit, scene
sc = scene.cfg('cfg3', n_frames=32)
ft = fit.Fitter(sc, fit.FitConfig(max_iter=80000, init_texture="random", enable_mip=True, max_mip_level=6), device="cuda")
for _ in range(7):
    ft.step()
torch.cuda.synchronize()
