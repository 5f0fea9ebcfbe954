import sys, torch
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import fpc_diffrend_amd.ops as dr
from helpers import random_soup, rel_l2
dev='cuda'
res=(97,131)
pos, tri = random_soup(4, 5, seed=21, spread=0.7, size=0.9)
tri = tri.to(dev)
g0 = torch.Generator().manual_seed(8)
uv = (torch.rand(15, 2, generator=g0) * 1.2 - 0.1).to(dev)
uv_idx = tri.clone()
g = torch.Generator().manual_seed(1)
tex0 = (torch.rand(48, 64, 1, generator=g) * 0.5).to(dev)
ref = torch.randint(0, 141, (pos.shape[0], res[0], res[1]), generator=g, dtype=torch.uint8).to(dev)
ctx = dr.RasterizeGLContext(device=dev)
pos = pos.to(dev)
for b in range(4):
    for bm in ('wrap','clamp'):
        l = []
        for one in (True, False):
            with torch.no_grad():
                l.append(float(dr.pixel_objective(ctx, pos[b:b+1].contiguous(), tri, uv, uv_idx, tex0, ref[b:b+1].contiguous(), res, boundary_mode=bm, one_pass=one, launch_hints=False)))
        print(b, bm, l, l[0]-l[1])
# per-pixel: compare ids from rasterize with ... 
rast,_ = dr.rasterize(ctx, pos, tri, res)
print("covered", (rast[...,3]>0).sum().item())
