"""
oracle/fit.py -- TEST INFRASTRUCTURE ONLY (checker and CPU baseline; never on the product path).

PyTorch-CPU restatement of one optimisation step of the reference's fit loop
(reference src/torch/fit.py:524-618) on top of the oracle's software raster ops (oracle/ops.py):

    MVP chain        fit.py:541-553   mvp = P . Rt(q_f,t_f) . Rt(q_c,t_c) . MV . T(0,170,0)
    blend (prior)    fit.py:115-122   V = v_base + B (M2 (M1 e_f))
    render           fit.py:134-162   transform_clip -> rasterize -> interpolate -> texture -> antialias -> where
    loss             fit.py:579       mean((ref - 255 colour)^2)   (+ Laplacian term fit.py:581 when weighted)
    Adam / renorm    fit.py:493-505, 610-618

The reference's own loop cannot run (stray return at fit.py:426-427, absent nvdiffrast / roma /
pytorch3d), so this restatement IS the "PyTorch-CPU software-raster run of the
reference" that BASELINE.json asks to time beside the GPU (cpu_baseline.kind = "port").
Parity of the four raster ops is unpinned (see oracle/ops.py); camera matrices are pinned by
tests/golden/camera_golden.json.
"""
import time

import numpy as np
import torch

from fpc_diffrend_amd import camera  # host camera math, pinned against the reference by golden fixtures
from . import ops as O

BACKGROUND = 45.0 / 255.0


def quat_to_rotmat(q):
    """XYZW -> 3x3, roma.unitquat_to_rotmat convention (reference fit.py:548); restated independently."""
    x, y, z, w = q.unbind(-1)
    return torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
        2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
        2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], dim=-1).reshape(q.shape[:-1] + (3, 3))


def rigid(t, R):
    top = torch.cat([R, t.reshape(t.shape[:-1] + (3, 1))], dim=-1)
    bottom = torch.tensor([0.0, 0.0, 0.0, 1.0], dtype=t.dtype).expand(t.shape[:-1] + (1, 4))
    return torch.cat([top, bottom], dim=-2)


class State:
    """Parameters of the fit (reference fit.py:433-461), CPU tensors."""

    def __init__(self, sc, cams, texture=None, dtype=torch.float32):
        F, K = sc.weights_gt.shape
        self.dtype = dtype
        self.sc, self.cams = sc, list(cams)
        self.v_base = torch.tensor(sc.v_base)
        self.Bmat = torch.tensor(sc.blendshapes)
        self.pos_idx = torch.tensor(sc.pos_idx)
        self.uv = torch.tensor(sc.uv)
        self.uv_idx = torch.tensor(sc.uv_idx)
        self.M1 = torch.zeros(F, F, requires_grad=True)
        self.M2 = torch.eye(K, F).requires_grad_(True)
        self.t_opt = torch.zeros(9, 3, requires_grad=True)
        q = torch.zeros(9, 4); q[:, 3] = 1
        self.q_opt = q.requires_grad_(True)
        self.per_frame_t = torch.zeros(F, 3, requires_grad=True)
        q = torch.zeros(F, 4); q[:, 3] = 1
        self.per_frame_q = q.requires_grad_(True)
        self.tex = torch.tensor(sc.texture if texture is None else texture).clone().requires_grad_(True)
        if dtype != torch.float32:
            for name in ('v_base', 'Bmat', 'uv', 'M1', 'M2', 't_opt', 'q_opt', 'per_frame_t', 'per_frame_q', 'tex'):
                t = getattr(self, name)
                setattr(self, name, t.detach().to(dtype).requires_grad_(t.requires_grad))
        trans = camera.translate(0.0, 170.0, 0.0)
        self.P = torch.tensor(np.stack([camera.intrinsic_to_projection(sc.cams[c]['intr']) for c in self.cams])).to(dtype)
        self.TMV = torch.tensor(np.stack([camera.extrinsic_to_modelview(sc.cams[c]['rot'], sc.cams[c]['trans_calib']) @ trans
                                          for c in self.cams])).to(dtype)

    def params(self):
        return [self.M1, self.M2, self.t_opt, self.q_opt, self.per_frame_t, self.per_frame_q, self.tex]


def clip_positions(st, frame_ids):
    """Blend + MVP chain + transform_clip (reference fit.py:541-564, camera.py:19-23): pos_clip [Fb*Nc,V,4]."""
    Fb, Nc = len(frame_ids), len(st.cams)
    # fit.py:115-122 with a one-hot e_f: column f of M1, then M2, then B
    w = torch.matmul(st.M2, st.M1[:, frame_ids])                     # [K,Fb]
    verts = (st.v_base[None] + torch.matmul(st.Bmat, w).t()).reshape(Fb, -1, 3)
    cs = torch.tensor(st.cams)
    rc = rigid(st.t_opt[cs], quat_to_rotmat(st.q_opt[cs]))             # fit.py:547-548
    rf = rigid(st.per_frame_t[frame_ids], quat_to_rotmat(st.per_frame_q[frame_ids]))  # fit.py:549-550
    tr = torch.matmul(rc, st.TMV)
    mvp = torch.matmul(st.P[None], torch.matmul(rf[:, None], tr[None])).reshape(Fb * Nc, 4, 4)  # fit.py:551-553
    posw = torch.cat([verts, torch.ones(Fb, verts.shape[1], 1, dtype=verts.dtype)], dim=-1).repeat_interleave(Nc, dim=0)
    return torch.matmul(posw, mvp.transpose(1, 2)), verts          # camera.py:19-23


def forward_from_clip(st, pos_clip, targets, enable_mip=False, max_mip_level=6, ids=None):
    """reference fit.py:151-161 + pixel loss fit.py:579 on given clip positions.  Returns (loss, image, rast)."""
    H, W = st.sc.resolution
    B = pos_clip.shape[0]
    rast, rast_db = O.rasterize(pos_clip, st.pos_idx, (H, W), ids=ids)
    if enable_mip:
        texc, texd = O.interpolate(st.uv[None], rast, st.uv_idx, rast_db=rast_db, diff_attrs='all')
        colour = O.texture(st.tex[None], texc, texd, filter_mode='linear-mipmap-linear', max_mip_level=max_mip_level)
    else:
        texc, _ = O.interpolate(st.uv[None], rast, st.uv_idx)
        colour = O.texture(st.tex[None], texc, filter_mode='linear')
    colour = O.antialias(colour, rast, pos_clip, st.pos_idx)
    image = torch.where(rast[..., 3:] > 0, colour, torch.tensor(BACKGROUND, dtype=colour.dtype))   # fit.py:161
    ref = targets.reshape(B, H, W, 1).to(colour.dtype)
    loss = torch.mean((ref - image * 255) ** 2)                                        # fit.py:579
    return loss, image, rast


def forward(st, frame_ids, targets, enable_mip=False, max_mip_level=6, weight_laplacian=0.0, ids=None):
    """Loss of a batch of frames x the state's cameras; targets uint8 [Fb,Nc,H,W].  Returns (loss, image, rast).
    `ids`: visibility override (see oracle.ops.rasterize)."""
    pos_clip, verts = clip_positions(st, frame_ids)
    loss, image, rast = forward_from_clip(st, pos_clip, targets, enable_mip, max_mip_level, ids)
    if weight_laplacian:
        loss = loss + weight_laplacian * uniform_laplacian(verts, st.pos_idx) ** 2      # fit.py:581
    return loss, image, rast


def uniform_laplacian(verts, faces):
    """pytorch3d mesh_laplacian_smoothing(method='uniform') restated: mean_v |mean_{N(v)} x - x_v|."""
    f = faces.long()
    e = torch.cat([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]], dim=0)
    e = torch.unique(torch.sort(e, dim=1)[0], dim=0)
    V = verts.shape[1]
    A = torch.zeros(V, V)
    A[e[:, 0], e[:, 1]] = 1
    A[e[:, 1], e[:, 0]] = 1
    deg = A.sum(dim=1, keepdim=True).clamp(min=1)
    L = A / deg - torch.eye(V)
    return torch.matmul(L[None], verts).norm(dim=2).mean()


def _smoke_state(sc, cams, dtype):
    st = State(sc, cams, dtype=dtype)
    F = sc.weights_gt.shape[0]
    with torch.no_grad():
        st.M1.copy_(torch.eye(F))
        st.M2.copy_(0.5 * torch.tensor(sc.weights_gt).t())
        st.per_frame_t.copy_(0.5 * torch.tensor(sc.t_gt))
    return st, F


def _grads(st):
    return {'grad_w': st.M2.grad.clone(), 'grad_tex': st.tex.grad.clone() if st.tex.grad is not None else None,
            'grad_pose': torch.cat([st.per_frame_t.grad.reshape(-1), st.per_frame_q.grad.reshape(-1),
                                    st.t_opt.grad.reshape(-1), st.q_opt.grad.reshape(-1)])}


def smoke_step(sc, cams=(0, 4), dtype=torch.float32, ids=None):
    """Same small forward + backward as fpc_diffrend_amd.fit.smoke_step, end to end on the oracle."""
    from fpc_diffrend_amd.fit import smoke_targets
    st, F = _smoke_state(sc, cams, dtype)
    loss, image, rast = forward(st, torch.arange(F), smoke_targets(sc, cams), ids=ids)
    loss.backward()
    out = {'loss': loss.detach(), 'ids': rast[..., 3].to(torch.int32), 'image': image.detach()}
    out.update(_grads(st))
    return out


def smoke_from_clip(sc, pos_clip, cams=(0, 4), dtype=torch.float32, ids=None):
    """Raster chain + loss on GIVEN clip positions (a leaf): the four ops are compared on bit-identical input.
    Returns ids, image, loss and the gradients w.r.t. pos_clip and the texture."""
    from fpc_diffrend_amd.fit import smoke_targets
    st, _ = _smoke_state(sc, cams, dtype)
    p = pos_clip.detach().to(dtype).clone().requires_grad_(True)
    loss, image, rast = forward_from_clip(st, p, smoke_targets(sc, cams), ids=ids)
    loss.backward()
    return {'loss': loss.detach(), 'ids': rast[..., 3].to(torch.int32), 'image': image.detach(),
            'grad_pos_clip': p.grad.clone(), 'grad_tex': st.tex.grad.clone()}


def smoke_upstream(sc, grad_pos_clip, cams=(0, 4), dtype=torch.float64):
    """Chain a given d loss / d pos_clip back through transform_clip, the MVP chain and the blend
    (reference fit.py:541-564) on the CPU: the reference for the parameter gradients of the GPU chain."""
    st, F = _smoke_state(sc, cams, dtype)
    pos_clip, _ = clip_positions(st, torch.arange(F))
    pos_clip.backward(grad_pos_clip.detach().to(dtype))
    return _grads(st)


def timed_steps(sc, cams, frame_ids, steps=1, threads=None, keep_first=False):
    """CPU baseline: full optimisation steps (forward, backward, Adam) on the host cores.
    Returns (seconds per step, images per step, threads used)."""
    if threads:
        torch.set_num_threads(threads)
    st = State(sc, cams)
    H, W = sc.resolution
    targets = torch.full((len(frame_ids), len(cams), H, W), 90, dtype=torch.uint8)
    opt = torch.optim.Adam(st.params(), lr=1e-3)
    fid = torch.tensor(list(frame_ids))
    first = None
    t0 = time.perf_counter()
    for i in range(steps):
        opt.zero_grad()
        pos_clip, _ = clip_positions(st, fid)
        loss, image, rast = forward_from_clip(st, pos_clip, targets)
        if i == 0 and keep_first:    # what the first forward pass saw and produced, for a parity check of the same image
            first = {"pos_clip": pos_clip.detach().clone(), "image": image.detach().clone(), "ids": rast[..., 3].detach().clone(),
                     "tex": st.tex.detach().clone(), "state": st}
        loss.backward()
        opt.step()
    dt = (time.perf_counter() - t0) / steps
    if keep_first:
        return dt, len(frame_ids) * len(cams), torch.get_num_threads(), first
    return dt, len(frame_ids) * len(cams), torch.get_num_threads()
