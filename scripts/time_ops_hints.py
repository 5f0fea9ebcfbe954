"""HIP-event time of the nine operator calls of the reference's render() + backward at cfg3 (288 images of 1920x1080), with
and without region hints."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fpc_diffrend_amd import _lib, camera, fit, scene
import fpc_diffrend_amd.ops as dr
sc = scene.cfg('cfg3', n_frames=int(os.environ.get("FRAMES", 32)))
ft = fit.Fitter(sc, fit.FitConfig(max_iter=80000, init_texture="random"), device="cuda")
ids = slice(0, ft.n_frames)
for hints in (True, False, True, 'empty'):   # 'empty': hint map zeroed after rasterize = the floor of writing the outputs
    dr.region_hints = bool(hints)
    timer = _lib.KernelTimer()
    for rep in range(3):
        if rep == 1:
            torch.cuda.synchronize(); _lib.TIMER = timer
        verts = ft.vertices(ids).reshape(ft.n_frames, -1, 3).detach()
        pos = camera.transform_clip(ft.mvp(ids).detach(), verts).requires_grad_(True)
        tex = ft.tex_opt.detach().clone().requires_grad_(True)
        ctx = dr.RasterizeGLContext(output_db=False, device=ft.device)
        rast, _ = dr.rasterize(ctx, pos, ft.pos_idx, ft.resolution)
        if hints == 'empty':
            dr._hint_of(rast, 'rast')[0].zero_()
        texc, _ = dr.interpolate(ft.uv[None], rast, ft.uv_idx)
        col = dr.texture(tex[None], texc, filter_mode='linear')
        aa = dr.antialias(col, rast, pos, ft.pos_idx)
        tagged = [dr._hint_of(rast, 'rast') is not None, dr._hint_of(texc, 'zero') is not None, dr._hint_of(col, 'const') is not None]
        img = torch.where(rast[..., 3:] > 0, aa, torch.tensor(fit.BACKGROUND, device=ft.device))
        loss = torch.mean((ft.targets.reshape(-1, *ft.resolution)[..., None].float() - img * 255) ** 2)
        loss.backward()
        del rast, texc, col, aa, img, loss, pos, tex
    summ = timer.summary(); _lib.TIMER = None
    print("hints", hints, tagged, {k.replace("fpcdr_", ""): round(v[1] / v[0], 3) for k, v in summ.items() if v[1] / v[0] > 0.1},
          "sum", round(sum(v[1] / v[0] for v in summ.values()), 2))
