"""A few fit steps at cfg3 (or --workload) for rocprofv3: `rocprofv3 --kernel-trace --stats ... -- python3 scripts/prof_objective.py`.
Prints the HIP-event times of the two objective calls as well."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fpc_diffrend_amd import _lib, fit, scene
ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="cfg3")
ap.add_argument("--frames", type=int, default=0, help="frames per step (0: the bench's default for the workload: 32 at cfg3, 1 at cfg2, 4 at cfg5)")
ap.add_argument("--mip", action="store_true", help="the reference's enable_mip branch (bench.py --mip)")
ap.add_argument("--channels", type=int, default=1)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--fill", type=float, default=0.0, help="head height as a fraction of the image height (default: the scene's 0.6)")
ap.add_argument("--frames-per-step", type=int, default=0, help="frames drawn per step (0: all)")
ap.add_argument("--views-per-step", type=int, default=0, help="cameras drawn per step (0: all); 1 + --frames-per-step 1 = the reference's run shape")
ap.add_argument("--graph", type=int, default=0, help="1: FitConfig.hip_graph=True (the step replayed as HIP graphs; no per-call timer then)")
ap.add_argument("--ops", type=int, default=1, help="also run the eight operator calls of render() + backward once at the full batch (their kernels' counters)")
a = ap.parse_args()
import bench      # (the workloads are defined ONCE, in bench.workload_config: these passes are paired with the bench's own timings)
cfg, fpg, _ = bench.workload_config(a.workload, a.mip)
a.frames = a.frames or fpg
sc = scene.cfg(a.workload, n_frames=a.frames)
if a.fill:
    sc.cams = scene.make_cameras(sc.resolution, fill=a.fill)
if a.channels != 1:
    import numpy as np
    sc.texture = np.repeat(sc.texture, a.channels, axis=2)[:, :, :a.channels].copy()
if a.frames_per_step or a.views_per_step:
    cfg.frames_per_step, cfg.views_per_step = a.frames_per_step, a.views_per_step
cfg.hip_graph = bool(a.graph)
ft = fit.Fitter(sc, cfg, device="cuda")
for _ in range(2 + (fit.Fitter.GRAPH_WARMUP + 3 if a.graph else 0)):
    ft.step()
torch.cuda.synchronize()
t = _lib.KernelTimer(names=list(bench.algorithmic_bytes_per_px(1, False).keys()))
_lib.TIMER = None if a.graph else t
for _ in range(a.steps):
    ft.step()
_lib.TIMER = None
print(json.dumps({k: v[1] / v[0] for k, v in t.summary().items()}))
import fpc_diffrend_amd.ops as _dr
if a.ops and a.workload != "cfg2" and not a.mip:
    from fpc_diffrend_amd import camera
    ids = slice(0, ft.n_frames)
    for _ in range(2):
        verts = ft.vertices(ids).reshape(ft.n_frames, -1, 3).detach()
        pos = camera.transform_clip(ft.mvp(ids).detach(), verts).requires_grad_(True)
        tex = ft.tex_opt.detach().clone().requires_grad_(True)
        ctx = _dr.RasterizeGLContext(output_db=False, device=ft.device)
        rast, _ = _dr.rasterize(ctx, pos, ft.pos_idx, ft.resolution)
        texc, _ = _dr.interpolate(ft.uv[None], rast, ft.uv_idx)
        col = _dr.texture(tex[None], texc, filter_mode='linear')
        aa = _dr.antialias(col, rast, pos, ft.pos_idx)
        (aa * (rast[..., 3:] > 0)).sum().backward()
        del rast, texc, col, aa, pos, tex
    torch.cuda.synchronize()
for k, h in _dr._list_hints.items():
    print("hints", k, "host counts [bwd,-,bins,fix]:", h.host.tolist(), "caps", h.caps, file=sys.stderr)
