"""How much host time does one Fitter.step() take to ENQUEUE (no synchronisation inside the loop)?  If this approaches the
GPU time of a step, the loop is launch-bound."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fpc_diffrend_amd import fit, scene
sc = scene.cfg('cfg3', n_frames=32)
ft = fit.Fitter(sc, fit.FitConfig(max_iter=80000, frames_per_step=0, init_texture="random"), device="cuda")
for _ in range(3):
    ft.step()
torch.cuda.synchronize()
for n in (5, 10, 20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        ft.step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{n} steps: enqueue {1e3 * (t1 - t0) / n:.3f} ms/step, complete {1e3 * (t2 - t0) / n:.3f} ms/step", flush=True)
