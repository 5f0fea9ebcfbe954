import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
import numpy as np, torch
from fpc_diffrend_amd import fit, scene
from oracle import fit as ofit, ops as O
from helpers import rel_l2
O.build()
mode = sys.argv[1] if len(sys.argv) > 1 else 'prior'
cams = (0, 4)
sc = scene.cfg('cfg1', n_frames=2)
targets = fit.smoke_targets(sc, cams)
hp = dict(max_iter=100, lr_base=2e-3, lr_tex_coef=0.5, lr_ramp=0.005, lr_t=1e-3, lr_q=1e-4, weight_laplacian=float(os.environ.get('WL', 300.0)),
          weight_meshedge=float(os.environ.get('WE', 0.5)), weight_normalconsistency=float(os.environ.get('WN', 0.2)))
st, F = ofit.perturbed_state(sc, cams, mode=mode)
start = [p.detach().clone() for p in st.params()]
ft = fit.Fitter(sc, fit.FitConfig(cam_idxs=cams, mode=mode, **hp), device='cuda', targets=targets.cuda())
with torch.no_grad():
    for p, v in zip(ft.params, start):
        p.copy_(v.cuda())
tr = ofit.Trainer(st, **hp)
for i in range(3):
    lo = tr.step(torch.arange(F), targets)
    lg = float(ft.step())
    print(i, 'loss', lo, lg)
    for name, p, q in zip(ofit.State.NAMES, ft.params, st.params()):
        if q.grad is None or p.grad is None:
            continue
        print('   ', name, 'grad rel', f"{rel_l2(p.grad, q.grad):.2e}", 'param rel', f"{rel_l2(p, q):.2e}",
              'gmax', float(q.grad.abs().max()), (p.grad.cpu().flatten()[:4].tolist(), q.grad.flatten()[:4].tolist()) if q.numel() <= 4 else '')
