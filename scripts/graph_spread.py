"""Spread of the cfg2 ten-step trajectory of tests/test_gpu_large.py over repetitions, WITH CONTROLS: eager against eager, graph against
graph and eager against graph.  If the three spreads are alike, the difference between a replayed and an eager trajectory is the order of
the float atomics amplified by Adam, not the replay (whose single step is compared at 1e-5 in the test itself)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from fpc_diffrend_amd import fit, scene


def rel_l2(a, b):
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def run(graph):
    sc = scene.cfg('cfg2', n_frames=1)
    cfg = fit.FitConfig(max_iter=80000, frames_per_step=0, init_texture="random", shading='vertex', optimize_texture=False, hip_graph=graph)
    ft = fit.Fitter(sc, cfg, device='cuda')
    losses = [float(ft.step()) for _ in range(10)]
    return np.asarray(losses), [p.detach().double().cpu().clone() for p in ft.params]


reps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
runs = {g: [run(g) for _ in range(reps)] for g in (False, True)}
for name, xs, ys in (("eager vs eager", runs[False], runs[False]), ("graph vs graph", runs[True], runs[True]), ("eager vs graph", runs[False], runs[True])):
    worst_loss, worst_rel, worst_abs = 0.0, 0.0, 0.0
    for i, (la, pa) in enumerate(xs):
        for j, (lb, pb) in enumerate(ys):
            if xs is ys and j <= i:
                continue
            worst_loss = max(worst_loss, float(np.max(np.abs(la - lb) / np.abs(la))))
            for a, b in zip(pa, pb):
                if float(a.abs().max()) > 0:
                    worst_rel, worst_abs = max(worst_rel, rel_l2(b, a)), max(worst_abs, float((a - b).abs().max()))
    print(f"{name}: max rel loss diff {worst_loss:.2e}  params: max rel-L2 {worst_rel:.2e}  max abs {worst_abs:.2e}", flush=True)
