import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fpc_diffrend_amd import fit, scene, _lib
sc = scene.cfg('cfg3', n_frames=32)
targets = None
for qb in (False, True, False, True):
    ft = fit.Fitter(sc, fit.FitConfig(max_iter=80000, init_texture="random", queued_backward=qb), device="cuda", targets=targets)
    targets = ft.targets
    for _ in range(4): ft.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): ft.step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10 * 1e3
    _lib.TIMER = _lib.KernelTimer(names=["fpcdr_render_loss_fwd", "fpcdr_render_aa_bwd"])
    for _ in range(3): ft.step()
    s = _lib.TIMER.summary(); _lib.TIMER = None
    print("queued_backward", qb, "%.3f ms/step" % dt, {k: round(v[1] / v[0], 3) for k, v in s.items()})
    del ft
