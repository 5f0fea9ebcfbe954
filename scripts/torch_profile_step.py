import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from fpc_diffrend_amd import fit, scene
sc = scene.cfg('cfg3', n_frames=int(sys.argv[1]) if len(sys.argv) > 1 else 32)
ft = fit.Fitter(sc, fit.FitConfig(init_texture="random"), device="cuda")
for _ in range(2): ft.step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(3): ft.step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=60, max_name_column_width=60))
