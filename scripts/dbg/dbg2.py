import sys, ctypes, torch
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import fpc_diffrend_amd.ops as dr
from fpc_diffrend_amd import _lib, fit
from fpc_diffrend_amd.ops import _ptr, _stream, _cached_tri_uv, _cached_topology
from helpers import random_soup, rel_l2
dev='cuda'
res=(97,131); H,W=res
pos, tri = random_soup(4, 5, seed=21, spread=0.7, size=0.9)
tri = tri.to(dev)
g0 = torch.Generator().manual_seed(8)
uv = (torch.rand(15, 2, generator=g0) * 1.2 - 0.1).to(dev)
uv_idx = tri.clone()
g = torch.Generator().manual_seed(1)
tex = (torch.rand(48, 64, 1, generator=g) * 0.5).to(dev)
ref = torch.randint(0, 141, (pos.shape[0], res[0], res[1]), generator=g, dtype=torch.uint8).to(dev)
ctx = dr.RasterizeGLContext(device=dev)
b = int(sys.argv[1]) if len(sys.argv) > 1 else 2
pos = pos.to(dev)[b:b+1].contiguous(); ref = ref[b:b+1].contiguous()
lib = _lib.load()
B,V,_ = pos.shape; T = tri.shape[0]; Ht,Wt,C = tex.shape
adj = _cached_topology(tri)
u8 = lambda n: torch.zeros(n, dtype=torch.uint8, device=dev)
scratch = u8(lib.fpcdr_rasterize_scratch_bytes(B, T)); sil=u8(B*T); idp=u8(lib.fpcdr_idplane_bytes(B,H,W))
occ=u8(lib.fpcdr_occ_bytes(B,H,W)); cmask=u8(lib.fpcdr_cmask_bytes(B,H,W))
rec=torch.zeros(B,H,W,4,device=dev); color=torch.full((B,H,W,C),-1.0,device=dev); g_aa=torch.full((B,H,W,C),-7.0,device=dev)
ecol=torch.zeros(4,device=dev); acc=torch.zeros(256,dtype=torch.float64,device=dev)
tri_uv=_cached_tri_uv(uv,uv_idx)
n_total = B*H*W*C
p=_lib.Objective(pos=_ptr(pos), tri=_ptr(tri), adj=_ptr(adj), B=B, V=V, T=T, H=H, W=W, scratch=_ptr(scratch), uv=_ptr(uv),
   uv_tri=_ptr(uv_idx), Vt=uv.shape[0], tri_uv=_ptr(tri_uv), tex=_ptr(tex), Ht=Ht, Wt=Wt, C=C, boundary_mode=0, ref=_ptr(ref), bg=fit.BACKGROUND,
   color_scale=255.0, grad_scale=1.0/n_total, sil=_ptr(sil), idp=_ptr(idp), occ=_ptr(occ), cmask=_ptr(cmask), rec=_ptr(rec), color=_ptr(color),
   grad_aa=_ptr(g_aa), empty_color=_ptr(ecol), loss_sum=_ptr(acc), grad_pos=None, grad_tex=None)
_lib.call("fpcdr_objective_fwd", ctypes.byref(p), _stream())
torch.cuda.synchronize()
OX,OY=(W+31)//32,(H+31)//32
ids = idp.view(torch.int32).reshape(B,OY,OX,32,32).permute(0,1,3,2,4).reshape(B,OY*32,OX*32)[:, :H, :W]
idv = ids & 0xffffff
# chain
pc = pos.clone().requires_grad_(True)
rast,_ = dr.rasterize(ctx, pc, tri, res)
texc,_ = dr.interpolate(uv[None], rast, uv_idx)
col = dr.texture(tex[None], texc, filter_mode='linear')
aa = dr.antialias(col, rast, pc, tri)
print("id mismatches", (idv != rast[...,3].int()).sum().item(), "covered", (idv>0).sum().item())
defer = (color[...,0] != -1.0)
print("deferred", defer.sum().item(), "blended (chain)", ((aa != col)[...,0] & (rast[...,3]>0)).sum().item())
bl = ((aa != col)[...,0] & (rast[...,3]>0))
print("blended but not deferred", (bl & ~defer).sum().item())
# colour agreement on deferred
print("colour diff on deferred", (color[defer] - col[defer]).abs().max().item())
print("z diff on deferred", (rec[...,2][defer] - rast[...,2][defer]).abs().max().item(), "uv", (rec[...,:2][defer]-rast[...,:2][defer]).abs().max().item())
# final g_aa vs chain gradient of loss wrt aa
img = torch.where(rast[..., 3:] > 0, aa, torch.tensor(fit.BACKGROUND, device=dev))
aa.retain_grad()
loss = torch.mean((ref[..., None].float() - img * 255) ** 2); loss.backward()
gd = (g_aa[defer] - aa.grad[defer]).abs()
print("g_aa diff on deferred max", gd.max().item(), "ref scale", aa.grad.abs().max().item())
bad = (g_aa[...,0] - aa.grad[...,0]).abs() * defer > 1e-3*aa.grad.abs().max()
print("bad pixels", bad.sum().item())
idx = bad.nonzero()[:12]
for i in idx.tolist():
    _,y,x = i
    print((x,y), "id", idv[0,y,x].item(), "nbrs R,U,L,D", [idv[0,yy,xx].item() if 0<=yy<H and 0<=xx<W else None for xx,yy in ((x+1,y),(x,y+1),(x-1,y),(x,y-1))],
          "g_aa", g_aa[0,y,x,0].item(), "chain", aa.grad[0,y,x,0].item(), "col", col[0,y,x,0].item(), "aa", aa[0,y,x,0].item(), "sil", (ids[0,y,x].item()>>24)&7)
print("loss one-pass", (acc.sum().item()), "chain", loss.item())
# expected deferred set from the id planes
import torch.nn.functional as F
full = ids[0].long()
def maybe(a, b): return (((a ^ b) & 0xffffff) != 0) & (((a | b) >> 24) != 0)
exp = torch.zeros(H, W, dtype=torch.bool, device=dev)
exp[:, :-1] |= maybe(full[:, :-1], full[:, 1:]); exp[:, 1:] |= maybe(full[:, 1:], full[:, :-1])
exp[:-1, :] |= maybe(full[:-1, :], full[1:, :]); exp[1:, :] |= maybe(full[1:, :], full[:-1, :])
exp &= (full & 0xffffff) > 0
d = defer[0]
print("expected deferred", exp.sum().item(), "got", d.sum().item(), "missing", (exp & ~d).sum().item(), "extra", (d & ~exp).sum().item())
m = (exp & ~d).nonzero(); e = (d & ~exp).nonzero()
print("missing (y,x):", m[:20].tolist())
print("extra   (y,x):", e[:20].tolist())
