"""Spread of the cfg2 graph-vs-eager comparison of tests/test_gpu_large.py over repetitions (float-atomic order + Adam)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from fpc_diffrend_amd import fit, scene
def rel_l2(a, b): return float((a - b).norm() / b.norm().clamp_min(1e-30))
for rep in range(6):
    out = {}
    for graph in (False, True):
        sc = scene.cfg('cfg2', n_frames=1)
        cfg = fit.FitConfig(max_iter=80000, frames_per_step=0, init_texture="random", shading='vertex', optimize_texture=False, hip_graph=graph)
        ft = fit.Fitter(sc, cfg, device='cuda')
        losses = [float(ft.step()) for _ in range(10)]
        out[graph] = (np.asarray(losses), [p.detach().double().cpu().clone() for p in ft.params])
    a, b = out[False][0], out[True][0]
    print(rep, "max rel loss diff %.2e" % float(np.max(np.abs(a - b) / np.abs(a))),
          "params (rel_l2, max abs):", [(round(rel_l2(pg, pe), 5), float((pg - pe).abs().max())) for pe, pg in zip(out[False][1], out[True][1])], flush=True)
