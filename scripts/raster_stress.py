import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import fpc_diffrend_amd.ops as dr
dev='cuda'
ctx = dr.RasterizeGLContext(device=dev)
def run(name, pos, tri, res, reps=5):
    pos, tri = pos.to(dev), tri.to(dev)
    for _ in range(2): rast,_ = dr.rasterize(ctx, pos, tri, res)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(reps): rast,_ = dr.rasterize(ctx, pos, tri, res)
    torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/reps
    cov = float((rast[...,3]>0).float().mean())
    print(f"{name}: {dt*1e3:.3f} ms, coverage {cov:.3f}", flush=True)
B=32
# two full-screen triangles
quad = torch.tensor([[-1,-1,0,1],[1,-1,0,1],[1,1,0,1],[-1,1,0,1]], dtype=torch.float32)[None].repeat(B,1,1)
run("fullscreen quad 1080p x32", quad, torch.tensor([[0,1,2],[0,2,3]], dtype=torch.int32), (1080,1920))
# 200 large overlapping triangles
g = torch.Generator().manual_seed(0)
T=200
xy = (torch.rand(B,T,3,2,generator=g)*2-1)*1.2
z = (torch.rand(B,T,3,1,generator=g)*2-1)*0.9
pos = torch.cat([xy, z, torch.ones(B,T,3,1)], -1).reshape(B,T*3,4).contiguous()
run("200 huge random triangles 1080p x32", pos, torch.arange(T*3, dtype=torch.int32).reshape(T,3), (1080,1920))
# 100k tiny triangles in one image region (dense overdraw)
T=100000
c = (torch.rand(B,T,1,2,generator=g)*2-1)*0.3
xy = c + (torch.rand(B,T,3,2,generator=g)*2-1)*0.004
z = (torch.rand(B,T,3,1,generator=g)*2-1)*0.9
pos = torch.cat([xy, z, torch.ones(B,T,3,1)], -1).reshape(B,T*3,4).contiguous()
run("100k tiny triangles in a 576x324 patch x32", pos, torch.arange(T*3, dtype=torch.int32).reshape(T,3), (1080,1920))
