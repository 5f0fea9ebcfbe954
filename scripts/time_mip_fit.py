"""Fit steps at cfg3 with FitConfig.enable_mip (the reference's other render() branch): the fused objective against the four
separate operators.   python scripts/time_mip_fit.py [frames]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fpc_diffrend_amd import fit, scene
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 32
sc = scene.cfg('cfg3', n_frames=nf)
targets = None
for name, kw in (("fused objective", {}), ("four operators + torch loss", dict(fused_objective=False, fused_render=False, fused_loss=False)),
                 ("fused objective, no mip", dict(enable_mip=False))):
    cfg = fit.FitConfig(**dict(dict(max_iter=80000, init_texture="random", enable_mip=True, max_mip_level=6), **kw))
    ft = fit.Fitter(sc, cfg, device="cuda", targets=targets)
    targets = ft.targets
    for _ in range(3):
        ft.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        loss = ft.step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5 * 1e3
    from fpc_diffrend_amd import _lib
    _lib.TIMER = _lib.KernelTimer()
    ft.step()
    summ = _lib.TIMER.summary(); _lib.TIMER = None
    print("%-32s %.2f ms/step (loss %.3f)" % (name, dt, float(loss)), {k.replace("fpcdr_", ""): round(v[1] / v[0], 2) for k, v in summ.items() if v[1] / v[0] > 0.3})
    del ft
    torch.cuda.empty_cache()
