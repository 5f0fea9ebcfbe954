import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from fpc_diffrend_amd import fit, scene
sc = scene.cfg('cfg3', n_frames=int(sys.argv[1]) if len(sys.argv) > 1 else 32)
# DROPIN=1: the step through the four separate operators + the reference's torch loss chain
drop = dict(fused_objective=False, fused_render=False, fused_loss=False) if os.environ.get("DROPIN") else {}
ft = fit.Fitter(sc, fit.FitConfig(init_texture="random", **drop), device="cuda")
for _ in range(2): ft.step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(3): ft.step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=60, max_name_column_width=60))
if os.environ.get("DROPIN"):    # which operators launch the large copies / elementwise kernels
    rows = []
    for e in prof.events():
        ks = getattr(e, "kernels", [])
        for k in ks:
            if k.duration > 200:
                rows.append((e.name, k.name[:70], k.duration, e.input_shapes if hasattr(e, "input_shapes") else None))
    for r in rows[: len(rows) // 3]:
        print("%-28s %-72s %8.0f us" % r[:3])
