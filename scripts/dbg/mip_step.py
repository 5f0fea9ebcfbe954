import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fpc_diffrend_amd import fit, scene, _lib
sc = scene.cfg('cfg3', n_frames=32)
ft = fit.Fitter(sc, fit.FitConfig(max_iter=80000, init_texture="random", enable_mip=True, max_mip_level=6), device="cuda")
for _ in range(5):
    ft.step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    ft.step()
e1.record(); torch.cuda.synchronize()
_lib.TIMER = _lib.KernelTimer(names=["fpcdr_objective_fwd"])
for _ in range(3):
    ft.step()
s = _lib.TIMER.summary(); _lib.TIMER = None
print("mip step %.3f ms; objective %.3f ms" % (e0.elapsed_time(e1) / 20, s["fpcdr_objective_fwd"][1] / s["fpcdr_objective_fwd"][0]))
