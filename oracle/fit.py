"""
oracle/fit.py -- TEST INFRASTRUCTURE ONLY (checker and CPU baseline; never on the product path).

PyTorch-CPU restatement of the reference's fit loop body (reference src/torch/fit.py:524-618) on top of the
oracle's software raster ops (oracle/ops.py).  Nothing here imports the product package: the camera matrices are
restated from the reference's source text and pinned by tests/golden/camera_golden.json
(tests/test_oracle_fit.py::test_oracle_camera_matches_reference_golden).

    camera matrices  camera.py:27-41, 46-66, 108-112, 128-132
    MVP chain        fit.py:541-553   mvp = P . Rt(q_f,t_f) . Rt(q_c,t_c) . MV . T(0,170,0)
    blend            fit.py:115-122 (prior), 58-62 (free), 88-99 (combined, coefficient 0.5 at fit.py:562)
    render           fit.py:134-162   transform_clip -> rasterize -> interpolate -> texture -> antialias -> where
    loss             fit.py:578-595   pixel L2 + mesh-edge / Laplacian^2 / normal-consistency (+ optional L2 terms)
    optimiser        fit.py:493-505   Adam, ten parameter groups, LambdaLR lr * ramp^(i / max_iter)
    update           fit.py:603-618   combined-mode switch, step, whole-tensor quaternion division (quirk Q3)

The reference optimises ONE (camera, frame) image per step; a batch here is the mean of that per-image loss over
frames x cameras (each frame's regularisers counted once per frame), which is what the product computes.

The reference's own loop cannot run (stray return at fit.py:426-427, absent nvdiffrast / roma / pytorch3d;
SURVEY.md section 8c), so this restatement IS the "PyTorch-CPU software-raster run of the reference" that
BASELINE.json asks to time beside the GPU (cpu_baseline.kind = "port").  Parity of the four raster ops is
unpinned (see oracle/ops.py).
"""
import time

import numpy as np
import torch

from . import ops as O

BACKGROUND = 45.0 / 255.0   # fit.py:161


# ------------------------------------------------------------------------------------------------
# camera matrices, restated from the reference's text
# ------------------------------------------------------------------------------------------------

def intrinsic_to_projection(intr, zn=0.01, zf=200):
    """reference camera.py:27-41: GL projection from pixel intrinsics (centred principal point assumed)."""
    fx, fy, cx, cy = intr[0, 0], intr[1, 1], intr[0, 2], intr[1, 2]
    return np.array([[fx / cx, 0, 0, 0],
                     [0, fy / cy, 0, 0],
                     [0, 0, -(zf + zn) / (zf - zn), -(2 * zf * zn) / (zf - zn)],
                     [0, 0, -1, 0]]).astype(np.float32)


def extrinsic_to_modelview(rmat, tvec):
    """reference camera.py:46-66: [R|t] with the camera's y and z rows negated (OpenCV -> OpenGL)."""
    rt = np.concatenate([np.asarray(rmat), np.asarray(tvec).reshape(3, 1)], axis=1)
    flip = np.diag([1.0, -1.0, -1.0]).astype(rt.dtype)
    return np.concatenate([flip @ rt, np.array([[0, 0, 0, 1]], dtype=rt.dtype)], axis=0)


def translate(x, y, z):
    """reference camera.py:108-112"""
    return np.array([[1, 0, 0, x], [0, 1, 0, y], [0, 0, 1, z], [0, 0, 0, 1]], dtype=np.float32)


def quat_to_rotmat(q):
    """XYZW -> 3x3, roma.unitquat_to_rotmat convention (reference fit.py:548); no normalisation (quirk Q3)."""
    x, y, z, w = q.unbind(-1)
    return torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
        2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
        2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], dim=-1).reshape(q.shape[:-1] + (3, 3))


def rigid(t, R):
    """reference camera.py:128-132"""
    top = torch.cat([R, t.reshape(t.shape[:-1] + (3, 1))], dim=-1)
    bottom = torch.tensor([0.0, 0.0, 0.0, 1.0], dtype=t.dtype).expand(t.shape[:-1] + (1, 4))
    return torch.cat([top, bottom], dim=-2)


# ------------------------------------------------------------------------------------------------
# state
# ------------------------------------------------------------------------------------------------

class State:
    """Parameters of the fit (reference fit.py:433-480), CPU tensors.  `sc` is a plain data container with
    v_base, blendshapes, pos_idx, uv, uv_idx, texture, cams, resolution and the number of frames."""

    NAMES = ('m1', 'm2', 'm3', 'M1', 'M2', 't_opt', 'q_opt', 'per_frame_t', 'per_frame_q', 'tex')   # fit.py:493-502 order

    def __init__(self, sc, cams, texture=None, dtype=torch.float32, mode='prior', n_frames=None):
        assert mode in ('prior', 'free', 'combined')
        F = n_frames if n_frames is not None else sc.weights_gt.shape[0]
        K = sc.blendshapes.shape[1]
        self.dtype, self.mode = dtype, mode
        self.sc, self.cams, self.n_frames = sc, list(cams), F
        self.v_base = torch.tensor(sc.v_base)
        self.Bmat = torch.tensor(sc.blendshapes)
        self.pos_idx = torch.tensor(sc.pos_idx)
        self.uv = torch.tensor(sc.uv)
        self.uv_idx = torch.tensor(sc.uv_idx)
        prior = mode in ('prior', 'combined')
        self.M1 = torch.zeros(F, F, requires_grad=prior)                     # maps['local'], fit.py:223
        self.M2 = torch.eye(K, F).requires_grad_(prior)                      # maps_intermediate['local'], fit.py:227
        self.m1 = torch.eye(F).requires_grad_(mode == 'free')                # fit.py:174-176
        self.m2 = torch.eye(F).requires_grad_(mode == 'free')
        self.m3 = torch.zeros(self.v_base.shape[0], F).requires_grad_(mode == 'free')
        self.t_opt = torch.zeros(9, 3, requires_grad=True)
        q = torch.zeros(9, 4); q[:, 3] = 1
        self.q_opt = q.requires_grad_(True)
        self.per_frame_t = torch.zeros(F, 3, requires_grad=True)
        q = torch.zeros(F, 4); q[:, 3] = 1
        self.per_frame_q = q.requires_grad_(True)
        self.tex = torch.tensor(np.asarray(sc.texture if texture is None else texture, dtype=np.float32)).clone().requires_grad_(True)
        if dtype != torch.float32:
            for name in ('v_base', 'Bmat', 'uv') + self.NAMES:
                t = getattr(self, name)
                setattr(self, name, t.detach().to(dtype).requires_grad_(t.requires_grad))
        trans = translate(0.0, 170.0, 0.0)                                   # fit.py:545
        self.P = torch.tensor(np.stack([intrinsic_to_projection(sc.cams[c]['intr']) for c in self.cams])).to(dtype)
        self.TMV = torch.tensor(np.stack([extrinsic_to_modelview(sc.cams[c]['rot'], sc.cams[c]['trans_calib']) @ trans
                                          for c in self.cams])).to(dtype)

    def params(self):
        """The ten tensors in the order of the reference's Adam groups (fit.py:493-502)."""
        return [getattr(self, n) for n in self.NAMES]


def vertices(st, frame_ids):
    """Blend (reference fit.py:555-562) for a batch of frames: [Fb, 3V].  M e_f = column f of M."""
    out = None
    if st.mode in ('prior', 'combined'):
        w = torch.matmul(st.M2, st.M1[:, frame_ids])                     # fit.py:115-119  [K,Fb]
        out = st.v_base[None] + torch.matmul(st.Bmat, w).t()
        if st.mode == 'prior':
            return out
    basis = torch.matmul(st.m2, st.m1[:, frame_ids])                     # fit.py:58-60 / 92-94
    learned = torch.matmul(st.m3, basis).t()
    if st.mode == 'free':
        return st.v_base[None] + learned                                # fit.py:62
    return out + 0.5 * learned                                          # fit.py:99 with learned_coefficient=0.5 (fit.py:562)


def clip_positions(st, frame_ids):
    """Blend + MVP chain + transform_clip (reference fit.py:541-564, camera.py:19-23): (pos_clip [Fb*Nc,V,4], verts [Fb,V,3])."""
    frame_ids = torch.as_tensor(frame_ids)
    Fb, Nc = len(frame_ids), len(st.cams)
    verts = vertices(st, frame_ids).reshape(Fb, -1, 3)
    cs = torch.tensor(st.cams)
    rc = rigid(st.t_opt[cs], quat_to_rotmat(st.q_opt[cs]))             # fit.py:547-548
    rf = rigid(st.per_frame_t[frame_ids], quat_to_rotmat(st.per_frame_q[frame_ids]))  # fit.py:549-550
    tr = torch.matmul(rc, st.TMV)
    mvp = torch.matmul(st.P[None], torch.matmul(rf[:, None], tr[None])).reshape(Fb * Nc, 4, 4)  # fit.py:551-553
    posw = torch.cat([verts, torch.ones(Fb, verts.shape[1], 1, dtype=verts.dtype)], dim=-1).repeat_interleave(Nc, dim=0)
    return torch.matmul(posw, mvp.transpose(1, 2)), verts          # camera.py:19-23


def forward_from_clip(st, pos_clip, targets, enable_mip=False, max_mip_level=6, ids=None, return_flags=False):
    """reference fit.py:151-161 + pixel loss fit.py:579 on given clip positions.  Returns (loss, image, rast) and, with
    return_flags, the antialias pair flags [B,H,W] uint8 (bit 0: pair (p, p+x) blended, bit 1: pair (p, p+y))."""
    H, W = st.sc.resolution
    B = pos_clip.shape[0]
    rast, rast_db = O.rasterize(pos_clip, st.pos_idx, (H, W), ids=ids)
    if enable_mip:
        texc, texd = O.interpolate(st.uv[None], rast, st.uv_idx, rast_db=rast_db, diff_attrs='all')
        colour = O.texture(st.tex[None], texc, texd, filter_mode='linear-mipmap-linear', max_mip_level=max_mip_level)
    else:
        texc, _ = O.interpolate(st.uv[None], rast, st.uv_idx)
        colour = O.texture(st.tex[None], texc, filter_mode='linear')
    colour, flags = O.antialias(colour, rast, pos_clip, st.pos_idx, return_flags=True)
    image = torch.where(rast[..., 3:] > 0, colour, torch.tensor(BACKGROUND, dtype=colour.dtype))   # fit.py:161
    ref = targets.reshape(B, H, W, 1).to(colour.dtype)
    loss = torch.mean((ref - image * 255) ** 2)                                        # fit.py:579
    if return_flags:
        return loss, image, rast, flags
    return loss, image, rast


# ------------------------------------------------------------------------------------------------
# mesh regularisers (pytorch3d in the reference, fit.py:16-19, 578-582), dense restatements
# ------------------------------------------------------------------------------------------------

def _edges(faces):
    f = faces.long()
    e = torch.cat([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]], dim=0)
    return torch.unique(torch.sort(e, dim=1)[0], dim=0)


def uniform_laplacian(verts, faces):
    """pytorch3d mesh_laplacian_smoothing(method='uniform') per mesh: mean_v |mean_{N(v)} x - x_v|.  verts [Fb,V,3] -> [Fb]."""
    e = _edges(faces)
    V = verts.shape[1]
    A = torch.zeros(V, V, dtype=verts.dtype)
    A[e[:, 0], e[:, 1]] = 1
    A[e[:, 1], e[:, 0]] = 1
    deg = A.sum(dim=1, keepdim=True).clamp(min=1)
    L = A / deg - torch.eye(V, dtype=verts.dtype)
    return torch.matmul(L[None], verts).norm(dim=2).mean(dim=1)


def mesh_edge_loss(verts, faces, target=0.1):
    """pytorch3d mesh_edge_loss: mean over edges of (|e| - target)^2, per mesh [Fb]."""
    e = _edges(faces)
    d = verts[:, e[:, 0]] - verts[:, e[:, 1]]
    return ((d.norm(dim=2) - target) ** 2).mean(dim=1)


def mesh_normal_consistency(verts, faces):
    """pytorch3d mesh_normal_consistency: for every edge shared by exactly two faces, 1 - cos(n_a, n_b); per mesh [Fb]."""
    f = faces.long().numpy()
    owners = {}
    for fi, (a, b, c) in enumerate(f):
        for u, v in ((a, b), (b, c), (c, a)):
            owners.setdefault((min(u, v), max(u, v)), []).append(fi)
    pairs = [v for v in owners.values() if len(v) == 2]
    if not pairs:
        return verts.sum(dim=(1, 2)) * 0.0
    fa = torch.tensor([p[0] for p in pairs]); fb = torch.tensor([p[1] for p in pairs])
    ft = faces.long()

    def normals(idx):
        a, b, c = verts[:, ft[idx, 0]], verts[:, ft[idx, 1]], verts[:, ft[idx, 2]]
        return torch.cross(b - a, c - a, dim=-1)

    cos = torch.nn.functional.cosine_similarity(normals(fa), normals(fb), dim=-1, eps=1e-8)
    return (1.0 - cos).mean(dim=1)


def forward(st, frame_ids, targets, enable_mip=False, max_mip_level=6, weight_laplacian=0.0, weight_meshedge=0.0,
            weight_normalconsistency=0.0, ids=None):
    """Loss of a batch of frames x the state's cameras (fit.py:578-582); targets uint8 [Fb,Nc,H,W].
    Returns (loss, image, rast).  `ids`: visibility override (see oracle.ops.rasterize)."""
    pos_clip, verts = clip_positions(st, frame_ids)
    loss, image, rast = forward_from_clip(st, pos_clip, targets, enable_mip, max_mip_level, ids)
    if weight_meshedge:
        loss = loss + weight_meshedge * mesh_edge_loss(verts, st.pos_idx, 0.1).mean()            # fit.py:580 (Q2: literal 0.1)
    if weight_laplacian:
        loss = loss + weight_laplacian * (uniform_laplacian(verts, st.pos_idx) ** 2).mean()      # fit.py:581 (Q7: squared, per mesh)
    if weight_normalconsistency:
        loss = loss + weight_normalconsistency * mesh_normal_consistency(verts, st.pos_idx).mean()   # fit.py:582
    return loss, image, rast


# ------------------------------------------------------------------------------------------------
# optimiser (reference fit.py:493-505, 603-618)
# ------------------------------------------------------------------------------------------------

class Trainer:
    """Adam over the reference's ten groups + LambdaLR + the per-iteration update rules, on a State."""

    def __init__(self, st, max_iter=80000, lr_base=10e-4, lr_tex_coef=0.5, lr_ramp=0.005, lr_t=10e-6, lr_q=10e-6,
                 weight_laplacian=5000.0, weight_meshedge=0.0, weight_normalconsistency=0.0, regularize_correctives=False,
                 regularize_prior=False, enable_mip=False, max_mip_level=6):
        self.st, self.max_iter = st, max_iter
        self.w = dict(weight_laplacian=weight_laplacian, weight_meshedge=weight_meshedge,
                      weight_normalconsistency=weight_normalconsistency)
        self.regularize_correctives, self.regularize_prior = regularize_correctives, regularize_prior
        self.enable_mip, self.max_mip_level = enable_mip, max_mip_level
        corrective_lr = lr_base * 0.1 if st.mode == 'combined' else lr_base               # fit.py:465-480
        lrs = [corrective_lr] * 3 + [lr_base, lr_base, lr_t, lr_q, lr_t, lr_q, lr_base * lr_tex_coef]
        self.optimizer = torch.optim.Adam([{"params": p, "lr": lr} for p, lr in zip(st.params(), lrs)], lr=lr_base)
        self.scheduler = torch.optim.lr_scheduler.LambdaLR(self.optimizer, lr_lambda=lambda x: lr_ramp ** (float(x) / float(max_iter)))
        self.iteration = 0

    def loss(self, frame_ids, targets):
        """The forward pass of iteration self.iteration (fit.py:555-595) for a batch of frames: the loss tensor."""
        st, i = self.st, self.iteration
        frame_ids = torch.as_tensor(frame_ids)
        loss, _, _ = forward(st, frame_ids, targets, self.enable_mip, self.max_mip_level, **self.w)
        if self.regularize_correctives and st.mode == 'combined' and i > self.max_iter / 2:       # fit.py:583-588
            loss = loss + torch.mean(torch.matmul(st.m3, torch.matmul(st.m2, st.m1[:, frame_ids])) ** 2)
        if self.regularize_prior and st.mode == 'prior':                                          # fit.py:591-594
            loss = loss + torch.mean(torch.matmul(st.M2, st.M1[:, frame_ids]) ** 2)
        return loss

    def switch(self):
        """fit.py:603-608: in combined mode the learned basis becomes trainable once i > max_iter / 2.  The reference does
        this AFTER the iteration's forward pass, so the basis first receives a gradient in the following iteration."""
        if self.st.mode == 'combined' and self.iteration > self.max_iter / 2:
            for m in (self.st.m1, self.st.m2, self.st.m3):
                m.requires_grad = True

    def update(self):
        """fit.py:612-618 on the gradients now held by the parameters: Adam, schedule, whole-tensor quaternion division (Q3)."""
        st = self.st
        self.optimizer.step()
        self.scheduler.step()
        with torch.no_grad():
            st.q_opt /= torch.sum(st.q_opt ** 2) ** 0.5
            st.per_frame_q /= torch.sum(st.per_frame_q ** 2) ** 0.5
        self.iteration += 1

    def step(self, frame_ids, targets):
        """One iteration of the loop body fit.py:555-618 for a batch of frames.  Returns the loss (float)."""
        loss = self.loss(frame_ids, targets)
        self.switch()
        self.optimizer.zero_grad()
        loss.backward()
        self.update()
        return float(loss.detach())

    def step_with_gradients(self, grads):
        """The same iteration with the gradients GIVEN (ten entries in the order of State.NAMES, None = no gradient): the
        update rules alone.  The set of tensors that come with a gradient must be the set that was trainable during this
        iteration's forward pass, i.e. before switch()."""
        params = self.st.params()
        have = [g is not None for g in grads]
        want = [bool(p.requires_grad) for p in params]
        assert have == want, f"iteration {self.iteration}: gradients present for {have}, trainable during the forward pass {want}"
        self.switch()
        self.optimizer.zero_grad()
        for p, g in zip(params, grads):
            if g is not None:
                p.grad = g.detach().to(p.dtype).clone()
        self.update()


# ------------------------------------------------------------------------------------------------
# the smoke configuration shared with fpc_diffrend_amd.fit.smoke_step (callers hand over the targets)
# ------------------------------------------------------------------------------------------------

def perturbed_state(sc, cams, dtype=torch.float32, mode='prior'):
    """Half the synthetic ground truth: every gradient is non-trivial.  Free-form basis: a smooth non-zero pattern."""
    st = State(sc, cams, dtype=dtype, mode=mode)
    F = st.n_frames
    with torch.no_grad():
        st.M1.copy_(torch.eye(F))
        st.M2.copy_(0.5 * torch.tensor(sc.weights_gt).t())
        st.per_frame_t.copy_(0.5 * torch.tensor(sc.t_gt))
        if mode != 'prior':
            st.m3.copy_(free_form_pattern(st.v_base.shape[0], F).to(dtype))
    return st, F


def free_form_pattern(M, F):
    """Deterministic smooth per-vertex offsets [M,F] (about 1 mm) so that m1/m2/m3 all carry gradient."""
    i = torch.arange(M, dtype=torch.float64)[:, None]
    f = torch.arange(F, dtype=torch.float64)[None, :]
    return (0.1 * torch.sin(0.013 * i + 0.7 * f)).to(torch.float32)


def _grads(st):
    def g(t):
        return t.grad.clone() if t.grad is not None else None
    out = {'grad_w': g(st.M2), 'grad_tex': g(st.tex),
           'grad_pose': torch.cat([st.per_frame_t.grad.reshape(-1), st.per_frame_q.grad.reshape(-1),
                                   st.t_opt.grad.reshape(-1), st.q_opt.grad.reshape(-1)])}
    for n in ('M1', 'm1', 'm2', 'm3'):
        out['grad_' + n] = g(getattr(st, n))
    return out


def smoke_step(sc, targets, cams=(0, 4), dtype=torch.float32, ids=None, mode='prior'):
    """Same small forward + backward as fpc_diffrend_amd.fit.smoke_step, end to end on the oracle."""
    st, F = perturbed_state(sc, cams, dtype, mode)
    loss, image, rast = forward(st, torch.arange(F), targets, ids=ids)
    loss.backward()
    out = {'loss': loss.detach(), 'ids': rast[..., 3].to(torch.int32), 'image': image.detach()}
    out.update(_grads(st))
    return out


def smoke_from_clip(sc, pos_clip, targets, cams=(0, 4), dtype=torch.float32, ids=None, texture=None):
    """Raster chain + loss on GIVEN clip positions (a leaf): the four ops are compared on bit-identical input.
    Returns ids, image, loss and the gradients w.r.t. pos_clip and the texture."""
    st = State(sc, cams, dtype=dtype, texture=texture)
    p = pos_clip.detach().to(dtype).clone().requires_grad_(True)
    loss, image, rast, flags = forward_from_clip(st, p, targets, ids=ids, return_flags=True)
    loss.backward()
    return {'loss': loss.detach(), 'ids': rast[..., 3].to(torch.int32), 'image': image.detach(), 'rast': rast.detach(),
            'aa_flags': flags, 'grad_pos_clip': p.grad.clone(), 'grad_tex': st.tex.grad.clone()}


def smoke_upstream(sc, grad_pos_clip, cams=(0, 4), dtype=torch.float64, mode='prior'):
    """Chain a given d loss / d pos_clip back through transform_clip, the MVP chain and the blend
    (reference fit.py:541-564) on the CPU: the reference for the parameter gradients of the GPU chain."""
    st, F = perturbed_state(sc, cams, dtype, mode)
    if mode == 'combined':          # as after the switch of fit.py:603-608
        for m in (st.m1, st.m2, st.m3):
            m.requires_grad = True
    pos_clip, _ = clip_positions(st, torch.arange(F))
    pos_clip.backward(grad_pos_clip.detach().to(dtype))
    return _grads(st)


def timed_steps(sc, cams, frame_ids, steps=1, threads=None, keep_first=False):
    """CPU baseline: full optimisation steps (forward, backward, Adam over the ten groups) on the host cores.
    Returns (seconds per step, images per step, threads used)."""
    if threads:
        torch.set_num_threads(threads)
    st = State(sc, cams)
    H, W = sc.resolution
    targets = torch.full((len(frame_ids), len(cams), H, W), 90, dtype=torch.uint8)
    tr = Trainer(st, weight_laplacian=0.0)
    fid = torch.tensor(list(frame_ids))
    first = None
    t0 = time.perf_counter()
    for i in range(steps):
        if i == 0 and keep_first:    # what the first forward pass saw and produced, for a parity check of the same image
            with torch.no_grad():
                pos_clip, _ = clip_positions(st, fid)
                _, image, rast = forward_from_clip(st, pos_clip, targets)
            first = {"pos_clip": pos_clip.detach().clone(), "image": image.detach().clone(), "ids": rast[..., 3].detach().clone(),
                     "tex": st.tex.detach().clone(), "state": st}
            t0 = time.perf_counter()
        tr.step(fid, targets)
    dt = (time.perf_counter() - t0) / steps
    if keep_first:
        return dt, len(frame_ids) * len(cams), torch.get_num_threads(), first
    return dt, len(frame_ids) * len(cams), torch.get_num_threads()
