"""A few fit steps at cfg3 (or --workload) for rocprofv3: `rocprofv3 --kernel-trace --stats ... -- python3 scripts/prof_objective.py`.
Prints the HIP-event times of the two objective calls as well."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fpc_diffrend_amd import _lib, fit, scene
ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="cfg3")
ap.add_argument("--frames", type=int, default=32)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--fill", type=float, default=0.0, help="head height as a fraction of the image height (default: the scene's 0.6)")
a = ap.parse_args()
sc = scene.cfg(a.workload, n_frames=a.frames)
if a.fill:
    sc.cams = scene.make_cameras(sc.resolution, fill=a.fill)
ft = fit.Fitter(sc, fit.FitConfig(max_iter=80000, init_texture="random"), device="cuda")
for _ in range(2):
    ft.step()
torch.cuda.synchronize()
t = _lib.KernelTimer(names=["fpcdr_render_loss_fwd", "fpcdr_render_aa_bwd"])
_lib.TIMER = t
for _ in range(a.steps):
    ft.step()
_lib.TIMER = None
print(json.dumps({k: v[1] / v[0] for k, v in t.summary().items()}))
import fpc_diffrend_amd.ops as _dr
for k, h in _dr._list_hints.items():
    print("hints", k, "host counts [bwd,-,bins,fix]:", h.host.tolist(), "caps", h.caps, file=sys.stderr)
