"""ms per fit step at cfg3 for a few stream-overlap variants (HIP events around 40 steps)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fpc_diffrend_amd import fit, scene
import fpc_diffrend_amd.ops as dr
sc = scene.cfg("cfg3", n_frames=32)
res = {}
orig_side = dr._side_stream
for name, kw, nosil in (("base", {}, False), ("lap_on_main", {"overlap_regularisers": False}, False), ("sil_on_main", {}, True),
                        ("both_on_main", {"overlap_regularisers": False}, True), ("base2", {}, False)):
    dr._side_stream = (lambda dev: None) if nosil else orig_side
    ft = fit.Fitter(sc, fit.FitConfig(max_iter=80000, init_texture="random", **kw), device="cuda")
    for _ in range(6):
        ft.step()
    torch.cuda.synchronize()
    ts = []
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            ft.step()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 30)
    res[name] = [round(t, 4) for t in ts]
    print(name, res[name], flush=True)
    del ft
