import os, sys, time
sys.path.insert(0, os.getcwd())
import torch, numpy as np
from fpc_diffrend_amd import fit, scene
sc = scene.cfg('cfg3', n_frames=32)
sc.q_gt[:] = (0.0, 0.0, 0.0, 1.0)
cfg = fit.FitConfig(max_iter=600, frames_per_step=0, init_texture="random", lr_base=2e-3, lr_t=2e-3, lr_q=1e-6)
ft = fit.Fitter(sc, cfg, device="cuda")
ft.init_near_truth(0.7)
act = sc.weights_gt > 0
w0 = np.abs(ft.weights().cpu().numpy() - sc.weights_gt)[act].mean()
t0 = time.perf_counter()
for i in range(600):
    l = ft.step()
    if i % 100 == 0 or i == 599:
        print(i, float(l), flush=True)
torch.cuda.synchronize()
print("600 steps in %.2f s" % (time.perf_counter() - t0), "weight err", w0, "->", np.abs(ft.weights().cpu().numpy() - sc.weights_gt)[act].mean(),
      "finite params:", all(torch.isfinite(p).all().item() for p in ft.params), "steps skipped on the device:", ft.skipped_steps)
