"""
oracle/ops.py -- TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see below).

Differentiable PyTorch-CPU restatement of the four raster operators the reference
fit loop calls (reference src/torch/fit.py:151-160):

    dr.rasterize   fit.py:151     -> rasterize()
    dr.interpolate fit.py:154,157 -> interpolate()
    dr.texture     fit.py:155,158 -> texture()
    dr.antialias   fit.py:160     -> antialias()

The reference obtains these from the third-party package `nvdiffrast`, which is
neither vendored nor pinned nor installed, and the reference
ships no test, golden vector or fixture for them.  The restatement therefore
follows nvdiffrast's *published* behaviour (Laine et al. 2020 + API docs, SURVEY.md
Appendix A) with the open choices (fill rule, snapping, tie-breaks) fixed by this
build's own spec in DESIGN.md.  "Parity" for these four ops means HIP kernel ==
this oracle; that substitution is stated in every report.

Forward values are computed with plain torch ops on CPU; every backward pass is
torch.autograd through those ops (so it is derived independently of the hand-written
HIP backward kernels) and is cross-checked by finite differences in
tests/test_oracle_ops.py.  Integer visibility comes from oracle/raster_ref.c.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product package never does.
"""
import ctypes
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    """Compile oracle/raster_ref.c with gcc (recipe: oracle/Makefile)."""
    out = os.path.join(_HERE, "_build", "liboracle_raster.so")
    src = os.path.join(_HERE, "raster_ref.c")
    if force or not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE] + (["-B"] if force else []))
    return out


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "_build", "liboracle_raster.so")
        if not os.path.exists(path):
            build()
        lib = ctypes.CDLL(path)
        lib.fpcdr_oracle_rasterize_ids.restype = ctypes.c_int
        lib.fpcdr_oracle_rasterize_ids.argtypes = [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 5 + [
            ctypes.c_void_p, ctypes.c_void_p]
        lib.fpcdr_oracle_edge_table.restype = ctypes.c_int
        lib.fpcdr_oracle_edge_table.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        _LIB = lib
    return _LIB


# ------------------------------------------------------------------------------------------------
# rasterize  (reference call site fit.py:151; ctx fit.py:484)
# ------------------------------------------------------------------------------------------------

def rasterize_ids(pos, tri, resolution, return_depth=False):
    """Integer visibility: [B,H,W] int32, triangle index + 1, 0 = empty (oracle/raster_ref.c)."""
    H, W = int(resolution[0]), int(resolution[1])
    p = np.ascontiguousarray(pos.detach().cpu().numpy(), dtype=np.float32)
    t = np.ascontiguousarray(tri.detach().cpu().numpy(), dtype=np.int32)
    B, V, _ = p.shape
    ids = np.zeros((B, H, W), dtype=np.int32)
    depth = np.zeros((B, H, W), dtype=np.float32) if return_depth else None
    rc = _lib().fpcdr_oracle_rasterize_ids(p.ctypes.data, t.ctypes.data, B, V, t.shape[0], H, W, ids.ctypes.data,
                                           depth.ctypes.data if return_depth else None)
    assert rc == 0
    if return_depth:
        return torch.from_numpy(ids), torch.from_numpy(depth)
    return torch.from_numpy(ids)


def _bary(pos, tri, ids, H, W):
    """Perspective-correct barycentrics + analytic pixel derivatives for covered pixels.

    Returns (bidx, yy, xx, u, v, zw, db[N,4]) with float32 tensors differentiable w.r.t. pos.
    """
    bidx, yy, xx = torch.nonzero(ids > 0, as_tuple=True)
    t = (ids[bidx, yy, xx] - 1).long()
    vi = tri.long()[t]  # [N,3]
    p0 = pos[bidx, vi[:, 0]]
    p1 = pos[bidx, vi[:, 1]]
    p2 = pos[bidx, vi[:, 2]]
    fx = (2.0 * xx.to(pos.dtype) + 1.0) / W - 1.0
    fy = (2.0 * yy.to(pos.dtype) + 1.0) / H - 1.0
    w0, w1, w2 = p0[:, 3], p1[:, 3], p2[:, 3]
    p0x, p0y = p0[:, 0] - fx * w0, p0[:, 1] - fy * w0
    p1x, p1y = p1[:, 0] - fx * w1, p1[:, 1] - fy * w1
    p2x, p2y = p2[:, 0] - fx * w2, p2[:, 1] - fy * w2
    a0 = p1x * p2y - p1y * p2x
    a1 = p2x * p0y - p2y * p0x
    a2 = p0x * p1y - p0y * p1x
    at = a0 + a1 + a2
    iw = 1.0 / at
    b0 = a0 * iw
    b1 = a1 * iw
    # depth output carries no gradient (matches the build's HIP backward: only u,v,(db) are chained)
    with torch.no_grad():
        zw = (a0 * p0[:, 2] + a1 * p1[:, 2] + a2 * p2[:, 2]) / (a0 * w0 + a1 * w1 + a2 * w2)
        zw = torch.clamp(zw, -1.0, 1.0)
    # analytic d(b)/d(pixel)
    da0x = w2 * p1y - w1 * p2y
    da0y = w1 * p2x - w2 * p1x
    da1x = w0 * p2y - w2 * p0y
    da1y = w2 * p0x - w0 * p2x
    da2x = w1 * p0y - w0 * p1y
    da2y = w0 * p1x - w1 * p0x
    datx = da0x + da1x + da2x
    daty = da0y + da1y + da2y
    sx, sy = 2.0 / W, 2.0 / H
    dudx = (da0x - b0 * datx) * iw * sx
    dudy = (da0y - b0 * daty) * iw * sy
    dvdx = (da1x - b1 * datx) * iw * sx
    dvdy = (da1y - b1 * daty) * iw * sy
    # clamp to the triangle (float barycentrics of a snapped-coverage pixel may poke out by ~1/256 px)
    uc = torch.clamp(b0, 0.0, 1.0)
    vc = torch.clamp(b1, 0.0, 1.0)
    s = 1.0 / torch.clamp(uc + vc, min=1.0)
    u = uc * s
    v = vc * s
    db = torch.stack([dudx, dudy, dvdx, dvdy], dim=1)
    return bidx, yy, xx, t, u, v, zw, db


def rasterize_ids_ranges(pos2d, tri, resolution, ranges):
    """Range mode: image b shows triangles [first, first + count) of `tri` over the shared vertex array pos2d [V,4]; the ids
    stay indices into `tri` (+ 1)."""
    out = []
    for first, count in ranges.tolist():
        ids = rasterize_ids(pos2d[None], tri[first:first + count], resolution)[0] if count > 0 else \
            torch.zeros(int(resolution[0]), int(resolution[1]), dtype=torch.int32)
        out.append(torch.where(ids > 0, ids + first, ids))
    return torch.stack(out)


def rasterize(pos, tri, resolution, grad_db=True, ids=None, ranges=None):
    """dr.rasterize(glctx, pos[B,V,4], tri[T,3], resolution=(H,W)) -> (rast[B,H,W,4], rast_db[B,H,W,4]).
    Range mode: pos [V,4] + ranges [B,2] (first triangle, count) per image.

    Float outputs follow pos.dtype (float32 like the product; float64 gives a high-precision gradient
    reference).  `ids` overrides the visibility buffer (used to evaluate a float64 run on the float32 run's
    visibility, which is always decided on float32 positions)."""
    H, W = int(resolution[0]), int(resolution[1])
    if pos.dim() == 2:
        assert ranges is not None, "range mode needs ranges"
        if ids is None:
            ids = rasterize_ids_ranges(pos.to(torch.float32), tri, (H, W), ranges)
        pos = pos[None].expand(ranges.shape[0], -1, -1)
    assert pos.dim() == 3 and pos.shape[2] == 4
    B = pos.shape[0]
    if ids is None:
        ids = rasterize_ids(pos.to(torch.float32), tri, (H, W))
    bidx, yy, xx, t, u, v, zw, db = _bary(pos, tri, ids, H, W)
    vals = torch.stack([u, v, zw, (t + 1).to(pos.dtype)], dim=1)
    rast = torch.zeros(B, H, W, 4, dtype=pos.dtype).index_put((bidx, yy, xx), vals)
    if not grad_db:
        db = db.detach()
    rast_db = torch.zeros(B, H, W, 4, dtype=pos.dtype).index_put((bidx, yy, xx), db)
    return rast, rast_db


# ------------------------------------------------------------------------------------------------
# interpolate  (reference call sites fit.py:154, fit.py:157)
# ------------------------------------------------------------------------------------------------

def interpolate(attr, rast, tri, rast_db=None, diff_attrs=None):
    """dr.interpolate(attr[1|B,Vt,A], rast, tri[T,3], rast_db=None, diff_attrs=None) -> (out, out_da)."""
    if attr.dim() == 2:
        attr = attr[None]
    B, H, W, _ = rast.shape
    A = attr.shape[2]
    T = tri.shape[0]
    tid = rast[..., 3].detach().to(torch.int64) - 1
    valid = (tid >= 0) & (tid < T)
    bidx, yy, xx = torch.nonzero(valid, as_tuple=True)
    t = tid[bidx, yy, xx]
    vi = tri.long()[t]
    ab = bidx if attr.shape[0] > 1 else torch.zeros_like(bidx)
    a0 = attr[ab, vi[:, 0]]
    a1 = attr[ab, vi[:, 1]]
    a2 = attr[ab, vi[:, 2]]
    u = rast[bidx, yy, xx, 0:1]
    v = rast[bidx, yy, xx, 1:2]
    val = u * a0 + v * a1 + (1.0 - u - v) * a2
    out = torch.zeros(B, H, W, A, dtype=val.dtype).index_put((bidx, yy, xx), val)
    if diff_attrs is None or rast_db is None:
        return out, torch.zeros(B, H, W, 0, dtype=val.dtype)
    sel = list(range(A)) if (isinstance(diff_attrs, str) and diff_attrs == 'all') else [int(i) for i in diff_attrs]
    d = rast_db[bidx, yy, xx]  # du/dx du/dy dv/dx dv/dy
    e0 = (a0 - a2)[:, sel]
    e1 = (a1 - a2)[:, sel]
    dadx = d[:, 0:1] * e0 + d[:, 2:3] * e1
    dady = d[:, 1:2] * e0 + d[:, 3:4] * e1
    da = torch.stack([dadx, dady], dim=2).reshape(-1, 2 * len(sel))
    out_da = torch.zeros(B, H, W, 2 * len(sel), dtype=da.dtype).index_put((bidx, yy, xx), da)
    return out, out_da


# ------------------------------------------------------------------------------------------------
# texture  (reference call sites fit.py:155, fit.py:158)
# ------------------------------------------------------------------------------------------------

def _wrap_idx(i, n, boundary_mode):
    if boundary_mode == 'wrap':
        return torch.remainder(i, n)
    if boundary_mode in ('clamp', 'zero'):      # 'zero': a safe index; _inside() says whether the texel exists
        return torch.clamp(i, 0, n - 1)
    raise NotImplementedError(boundary_mode)


def _inside(ix, iy, Wt, Ht, boundary_mode, dtype):
    """Boundary mode 'zero' (texture padded with zeros): 1 where texel (ix, iy) lies inside the texture, else 0; [N,1]."""
    if boundary_mode != 'zero':
        return 1.0
    return ((ix >= 0) & (ix < Wt) & (iy >= 0) & (iy < Ht)).to(dtype).unsqueeze(-1)


def _tex_coords(uv, Ht, Wt, boundary_mode):
    u, v = uv[..., 0], uv[..., 1]
    if boundary_mode == 'wrap':
        u = u - torch.floor(u)
        v = v - torch.floor(v)
    elif boundary_mode == 'clamp':
        u = torch.clamp(u, 0.0, 1.0)
        v = torch.clamp(v, 0.0, 1.0)
    x = u * Wt - 0.5
    y = v * Ht - 0.5
    return x, y


def _bilinear(tex, tb, x, y, boundary_mode):
    """tex [N,Ht,Wt,C]; tb batch index per sample; x,y continuous texel coords (centre of texel i at i)."""
    Ht, Wt = tex.shape[1], tex.shape[2]
    x0f = torch.floor(x)
    y0f = torch.floor(y)
    fx = (x - x0f).unsqueeze(-1)
    fy = (y - y0f).unsqueeze(-1)
    x0 = x0f.long()
    y0 = y0f.long()
    ix0 = _wrap_idx(x0, Wt, boundary_mode)
    ix1 = _wrap_idx(x0 + 1, Wt, boundary_mode)
    iy0 = _wrap_idx(y0, Ht, boundary_mode)
    iy1 = _wrap_idx(y0 + 1, Ht, boundary_mode)
    t00 = tex[tb, iy0, ix0] * _inside(x0, y0, Wt, Ht, boundary_mode, tex.dtype)
    t10 = tex[tb, iy0, ix1] * _inside(x0 + 1, y0, Wt, Ht, boundary_mode, tex.dtype)
    t01 = tex[tb, iy1, ix0] * _inside(x0, y0 + 1, Wt, Ht, boundary_mode, tex.dtype)
    t11 = tex[tb, iy1, ix1] * _inside(x0 + 1, y0 + 1, Wt, Ht, boundary_mode, tex.dtype)
    top = t00 + (t10 - t00) * fx
    bot = t01 + (t11 - t01) * fx
    return top + (bot - top) * fy


def build_mip_chain(tex, max_mip_level=None):
    """2x2 box-filter chain; stops when a dim would become odd/zero or max_mip_level is reached."""
    chain = [tex]
    lvl = 0
    while True:
        h, w = chain[-1].shape[1], chain[-1].shape[2]
        if max_mip_level is not None and lvl >= max_mip_level:
            break
        if h % 2 or w % 2 or h < 2 or w < 2:
            break
        t = chain[-1]
        t = (t[:, 0::2, 0::2] + t[:, 0::2, 1::2] + t[:, 1::2, 0::2] + t[:, 1::2, 1::2]) * 0.25
        chain.append(t)
        lvl += 1
    return chain


def texture(tex, uv, uv_da=None, mip_level_bias=None, mip=None, filter_mode='auto', boundary_mode='wrap',
            max_mip_level=None):
    """dr.texture(tex[1|B,Ht,Wt,C], uv[B,H,W,2], uv_da=None, ..., filter_mode, boundary_mode, max_mip_level)."""
    if filter_mode == 'auto':
        filter_mode = 'linear-mipmap-linear' if (uv_da is not None or mip_level_bias is not None) else 'linear'
    B, H, W, _ = uv.shape
    Ht, Wt, C = tex.shape[1], tex.shape[2], tex.shape[3]
    flat_uv = uv.reshape(-1, 2)
    tb = torch.arange(B).repeat_interleave(H * W) if tex.shape[0] > 1 else torch.zeros(B * H * W, dtype=torch.long)
    if filter_mode == 'nearest':
        x, y = _tex_coords(flat_uv, Ht, Wt, boundary_mode)
        rx, ry = torch.floor(x + 0.5).long(), torch.floor(y + 0.5).long()
        ix = _wrap_idx(rx, Wt, boundary_mode)
        iy = _wrap_idx(ry, Ht, boundary_mode)
        return (tex[tb, iy, ix] * _inside(rx, ry, Wt, Ht, boundary_mode, tex.dtype)).reshape(B, H, W, C)
    if filter_mode == 'linear':
        x, y = _tex_coords(flat_uv, Ht, Wt, boundary_mode)
        return _bilinear(tex, tb, x, y, boundary_mode).reshape(B, H, W, C)
    if filter_mode in ('linear-mipmap-linear', 'linear-mipmap-nearest'):
        if mip is not None:       # a caller's own stack (levels 1..n); autograd gives every level its own gradient
            mips = list(mip) if max_mip_level is None else list(mip)[:int(max_mip_level)]
            chain = [tex] + mips
        else:
            chain = build_mip_chain(tex, max_mip_level)
        nlev = len(chain) - 1
        # level of detail from the uv footprint (uv_da = du/dx du/dy dv/dx dv/dy), in texels of level 0
        if uv_da is not None:
            d = uv_da.reshape(-1, 4)
            dudx, dudy, dvdx, dvdy = d[:, 0] * Wt, d[:, 1] * Wt, d[:, 2] * Ht, d[:, 3] * Ht
            # major axis of the footprint ellipse: largest eigenvalue of J J^T
            A_ = dudx * dudx + dudy * dudy
            B_ = dudx * dvdx + dudy * dvdy
            C_ = dvdx * dvdx + dvdy * dvdy
            tr = 0.5 * (A_ + C_)
            df = 0.5 * (A_ - C_)
            l2 = tr + torch.sqrt(df * df + B_ * B_ + 1e-30)  # squared major-axis length
            level = 0.5 * torch.log2(torch.clamp(l2, min=1e-30))
        else:
            level = torch.zeros(B * H * W, dtype=uv.dtype)
        if mip_level_bias is not None:
            level = level + mip_level_bias.reshape(-1)
        level = torch.clamp(level, 0.0, float(nlev))
        if filter_mode == 'linear-mipmap-nearest':
            l0 = torch.floor(level + 0.5).long().clamp(max=nlev)
            out = torch.zeros(B * H * W, C, dtype=tex.dtype)
            for l in range(nlev + 1):
                m = torch.nonzero(l0 == l, as_tuple=True)[0]
                if m.numel():
                    x, y = _tex_coords(flat_uv[m], chain[l].shape[1], chain[l].shape[2], boundary_mode)
                    out = out.index_put((m,), _bilinear(chain[l], tb[m], x, y, boundary_mode))
            return out.reshape(B, H, W, C)
        l0 = torch.floor(level).long().clamp(max=nlev)
        l1 = (l0 + 1).clamp(max=nlev)
        fl = (level - l0.to(level.dtype)).unsqueeze(-1)
        out = torch.zeros(B * H * W, C, dtype=tex.dtype)
        for l in range(nlev + 1):
            m = torch.nonzero(l0 == l, as_tuple=True)[0]
            if not m.numel():
                continue
            x, y = _tex_coords(flat_uv[m], chain[l].shape[1], chain[l].shape[2], boundary_mode)
            c0 = _bilinear(chain[l], tb[m], x, y, boundary_mode)
            lu = min(l + 1, nlev)
            x, y = _tex_coords(flat_uv[m], chain[lu].shape[1], chain[lu].shape[2], boundary_mode)
            c1 = _bilinear(chain[lu], tb[m], x, y, boundary_mode)
            out = out.index_put((m,), c0 + (c1 - c0) * fl[m])
        return out.reshape(B, H, W, C)
    raise NotImplementedError(filter_mode)


# ------------------------------------------------------------------------------------------------
# antialias  (reference call site fit.py:160)
# ------------------------------------------------------------------------------------------------

def edge_table(tri):
    """Per triangle edge: number of incident triangles and the other triangle's opposite vertex."""
    t = np.ascontiguousarray(tri.detach().cpu().numpy(), dtype=np.int32)
    cnt = np.zeros((t.shape[0], 3), dtype=np.int32)
    oth = np.full((t.shape[0], 3), -1, dtype=np.int32)
    rc = _lib().fpcdr_oracle_edge_table(t.ctypes.data, t.shape[0], cnt.ctypes.data, oth.ctypes.data)
    assert rc == 0
    return torch.from_numpy(cnt), torch.from_numpy(oth)


def silhouette_table(pos, tri, cnt, oth, H, W):
    """sil[b,t,e]: edge e of triangle t is a silhouette edge in image b (boundary edge, or both
    opposite vertices project to the same side).  Evaluated once per (image, triangle) in uncentred
    pixel-scaled homogeneous coordinates -- the classification is a determinant sign, so it does not
    depend on the pixel the edge is later tested against."""
    with torch.no_grad():
        B, V = pos.shape[0], pos.shape[1]
        hw, hh = 0.5 * W, 0.5 * H
        vi = tri.long()
        okv = ((vi >= 0) & (vi < V)).all(dim=1)
        vic = vi.clamp(0, V - 1)
        qx = pos[:, :, 0] * hw
        qy = pos[:, :, 1] * hh
        qw = pos[:, :, 3]
        sil = torch.zeros(B, tri.shape[0], 3, dtype=torch.bool)
        for e in range(3):
            a, b_ = vic[:, (e + 1) % 3], vic[:, (e + 2) % 3]
            o = vic[:, e]
            Lx = qy[:, a] * qw[:, b_] - qw[:, a] * qy[:, b_]
            Ly = qw[:, a] * qx[:, b_] - qx[:, a] * qw[:, b_]
            Lz = qx[:, a] * qy[:, b_] - qy[:, a] * qx[:, b_]
            so = Lx * qx[:, o] + Ly * qy[:, o] + Lz * qw[:, o]
            op = oth[:, e].long()
            opv = (op >= 0) & (op < V)
            opc = op.clamp(0, V - 1)
            sp = Lx * qx[:, opc] + Ly * qy[:, opc] + Lz * qw[:, opc]
            same = ((so > 0) & (sp > 0)) | ((so < 0) & (sp < 0))
            sil[:, :, e] = ((cnt[:, e] == 1)[None] | ((cnt[:, e] == 2) & opv)[None] & same) & okv[None]
        return sil


def _aa_pairs(color, rast, pos, tri, sil, d, out, flags):
    """Process all pixel pairs (p, p + e_d); accumulates into `out`, returns it. d=0: x pairs, d=1: y pairs."""
    B, H, W, C = color.shape
    T = tri.shape[0]
    ids = rast[..., 3].detach().to(torch.int64)
    zw = rast[..., 2].detach()
    if d == 0:
        id0, id1, z0, z1 = ids[:, :, :-1], ids[:, :, 1:], zw[:, :, :-1], zw[:, :, 1:]
    else:
        id0, id1, z0, z1 = ids[:, :-1, :], ids[:, 1:, :], zw[:, :-1, :], zw[:, 1:, :]
    bidx, yy, xx = torch.nonzero(id0 != id1, as_tuple=True)
    if bidx.numel() == 0:
        return out
    i0, i1 = id0[bidx, yy, xx], id1[bidx, yy, xx]
    zz0, zz1 = z0[bidx, yy, xx], z1[bidx, yy, xx]
    use1 = torch.where(i0 == 0, torch.ones_like(i0, dtype=torch.bool),
                       torch.where(i1 == 0, torch.zeros_like(i0, dtype=torch.bool), zz1 < zz0))
    tau = torch.where(use1, i1, i0) - 1
    ok = (tau >= 0) & (tau < T)
    ok = ok & sil[bidx, tau.clamp(0, T - 1)].any(dim=1)
    bidx, yy, xx, use1, tau = bidx[ok], yy[ok], xx[ok], use1[ok], tau[ok]
    if bidx.numel() == 0:
        return out
    dx, dy = (1, 0) if d == 0 else (0, 1)
    x0, y0, x1, y1 = xx, yy, xx + dx, yy + dy
    Px = torch.where(use1, x1, x0)
    Py = torch.where(use1, y1, y0)
    Qx = torch.where(use1, x0, x1)
    Qy = torch.where(use1, y0, y1)
    s = torch.where(use1, -torch.ones(1, dtype=pos.dtype), torch.ones(1, dtype=pos.dtype))
    vi = tri.long()[tau]
    hw, hh = 0.5 * W, 0.5 * H
    fxp = Px.to(pos.dtype) + 0.5 - hw
    fyp = Py.to(pos.dtype) + 0.5 - hh

    def proj(vidx):
        c = pos[bidx, vidx]
        return c[:, 0] * hw - fxp * c[:, 3], c[:, 1] * hh - fyp * c[:, 3], c[:, 3]

    q = [proj(vi[:, k]) for k in range(3)]
    cP = color[bidx, Py, Px]
    cQ = color[bidx, Qy, Qx]
    for e in range(3):
        a, b_ = (e + 1) % 3, (e + 2) % 3
        qax, qay, wa = q[a]
        qbx, qby, wb = q[b_]
        Lx = qay * wb - wa * qby
        Ly = wa * qbx - qax * wb
        Lz = qax * qby - qay * qbx
        if d == 0:
            Ld, Lo, ya, yb = Lx, Ly, qay, qby
            orient = Ld.abs() >= Lo.abs()
        else:
            Ld, Lo, ya, yb = Ly, Lx, qax, qbx
            orient = Ld.abs() > Lo.abs()
        extent = (ya < 0) != (yb < 0)
        nz = Ld != 0
        den = torch.where(nz, s * Ld, torch.ones_like(Ld))
        t = -Lz / den
        act = (sil[bidx, tau, e] & orient & extent & nz & (t >= 0) & (t <= 1)).detach()
        m = torch.nonzero(act, as_tuple=True)[0]
        if m.numel() == 0:
            continue
        tm = t[m]
        far = (tm >= 0.5).unsqueeze(-1)
        amt = torch.where(far, tm.unsqueeze(-1) - 0.5, 0.5 - tm.unsqueeze(-1))
        delta = torch.where(far, amt * (cP[m] - cQ[m]), amt * (cQ[m] - cP[m]))
        ry = torch.where(far[:, 0], Qy[m], Py[m])
        rx = torch.where(far[:, 0], Qx[m], Px[m])
        out = out.index_put((bidx[m], ry, rx), delta, accumulate=True)
        if flags is not None:
            flags[bidx[m], yy[m], xx[m]] |= (1 << d)
    return out


def antialias(color, rast, pos, tri, topology_hash=None, pos_gradient_boost=1.0, return_flags=False):
    """dr.antialias(color[B,H,W,C], rast[B,H,W,4], pos[B,V,4], tri[T,3]) -> [B,H,W,C]."""
    if pos.dim() == 2:      # range mode: one shared vertex array
        pos = pos[None].expand(color.shape[0], -1, -1)
    cnt, oth = edge_table(tri) if topology_hash is None else topology_hash
    if pos_gradient_boost != 1.0:
        # value-neutral gradient scaling
        pos = pos * pos_gradient_boost + (pos * (1.0 - pos_gradient_boost)).detach()
    B, H, W, C = color.shape
    flags = torch.zeros(B, H, W, dtype=torch.uint8) if return_flags else None
    sil = silhouette_table(pos, tri, cnt, oth, H, W)
    out = color.clone()
    out = _aa_pairs(color, rast, pos, tri, sil, 0, out, flags)
    out = _aa_pairs(color, rast, pos, tri, sil, 1, out, flags)
    if return_flags:
        return out, flags
    return out
