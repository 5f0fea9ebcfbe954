import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from fpc_diffrend_amd import _lib, scene
import fpc_diffrend_amd.ops as dr
from helpers import clip_positions
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 32
sc = scene.cfg('cfg3', n_frames=nf)
pos, _ = clip_positions(sc, list(range(9)), frames=list(range(nf)))
pos = pos.cuda().requires_grad_(True); tri = torch.tensor(sc.pos_idx).cuda()
ctx = dr.RasterizeGLContext(output_db=False)
with torch.no_grad():
    rast, _ = dr.rasterize(ctx, pos, tri, sc.resolution)
col = torch.rand(rast.shape[0], rast.shape[1], rast.shape[2], 1, device='cuda', requires_grad=True)
out = dr.antialias(col, rast, pos, tri)
gy = torch.randn_like(out)
t = _lib.KernelTimer()
for i in range(6):
    if i == 1: torch.cuda.synchronize(); _lib.TIMER = t
    col.grad = None; pos.grad = None
    out.backward(gy, retain_graph=True)
s = t.summary(); _lib.TIMER = None
n, ms = s['fpcdr_antialias_bwd']
px = rast.shape[0] * rast.shape[1] * rast.shape[2]
print(f"antialias_bwd {ms/n:.3f} ms  {8*px/(ms/n)/1e6:.0f} GB/s algorithmic ({100*8*px/(ms/n)/1e6/8000:.1f} % of 8 TB/s)")
