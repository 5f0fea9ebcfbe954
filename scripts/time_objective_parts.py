"""cfg3: HIP-event time of fpcdr_objective_fwd with both gradients, one of them, or none (the kernel's gradient work is gated by
run-time flags): what the gradient half of the one-pass shading kernel costs.  usage: python scripts/time_objective_parts.py [--two-call]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fpc_diffrend_amd import _lib, fit, scene, camera
import fpc_diffrend_amd.ops as dr
sc = scene.cfg("cfg3", n_frames=32)
ft = fit.Fitter(sc, fit.FitConfig(max_iter=80000, init_texture="random"), device="cuda")
ids = slice(0, ft.n_frames)
verts = ft.vertices(ids).reshape(ft.n_frames, -1, 3).detach()
pos0 = camera.transform_clip(ft.mvp(ids).detach(), verts).contiguous()
tex0 = ft.tex_opt.detach().clone()
ref = ft.targets.reshape(-1, *ft.resolution)
bg = ft.target_bg_sumsq.sum()
out = {}
for name, gp, gt in (("both", True, True), ("pos_only", True, False), ("tex_only", False, True), ("value_only", False, False)):
    p, t = pos0.clone().requires_grad_(gp), tex0.clone().requires_grad_(gt)
    for it in range(6):
        if it == 2:
            torch.cuda.synchronize()
            tm = _lib.KernelTimer(names=["fpcdr_objective_fwd"]); _lib.TIMER = tm
        loss = dr.pixel_objective(ft.glctx, p, ft.pos_idx, ft.uv, ft.uv_idx, t, ref, ft.resolution, ref_bg_sumsq=bg, unit_upstream=True)
        if gp or gt:
            loss.backward()
    _lib.TIMER = None
    out[name] = {k: round(v[1] / v[0], 4) for k, v in tm.summary().items()}
print(json.dumps(out))
