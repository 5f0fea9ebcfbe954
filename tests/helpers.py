"""Shared builders for the parity tests (seeded synthetic inputs; nothing reads /root/reference)."""
import numpy as np
import torch

from fpc_diffrend_amd import camera, scene


def rel_l2(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).norm() / max(float(b.norm()), 1e-30))


def scene_mvps(sc, cam_ids, frames=None):
    """mvp [len(frames)*len(cam_ids),4,4] float32 for the ground-truth pose of `frames` (frame-major)."""
    out = []
    T = camera.translate(0.0, 170.0, 0.0)
    frames = [None] if frames is None else frames
    for f in frames:
        for c in cam_ids:
            cam = sc.cams[c]
            P = camera.intrinsic_to_projection(cam['intr'])
            MV = camera.extrinsic_to_modelview(cam['rot'], cam['trans_calib'])
            m = MV @ T
            if f is not None:
                R = camera.unitquat_to_rotmat(torch.tensor(sc.q_gt[f])).numpy()
                Rt = np.eye(4, dtype=np.float32)
                Rt[:3, :3] = R
                Rt[:3, 3] = sc.t_gt[f]
                m = Rt @ m
            out.append((P @ m).astype(np.float32))
    return torch.tensor(np.stack(out))


def clip_positions(sc, cam_ids, frames=None, jitter=0.0, seed=0):
    """pos_clip [B,V,4] for the base mesh (+ blendshape ground truth per frame)."""
    mvps = scene_mvps(sc, cam_ids, frames)
    V = sc.n_vertices
    if frames is None:
        verts = torch.tensor(sc.v_base.reshape(1, V, 3))
    else:
        vb = torch.tensor(sc.v_base)
        Bm = torch.tensor(sc.blendshapes)
        w = torch.tensor(sc.weights_gt[frames])
        verts = (vb[None] + w @ Bm.t()).reshape(len(frames), V, 3)
    if jitter:
        g = torch.Generator().manual_seed(seed)
        verts = verts + jitter * torch.randn(verts.shape, generator=g)
    return camera.transform_clip(mvps, verts), mvps


def random_soup(B, T, seed, spread=1.2, size=0.5, wmin=0.5, wmax=3.0):
    """Random triangle soup in clip space, many overlaps, mixed facing, some off-screen; V = 3T."""
    g = torch.Generator().manual_seed(seed)
    c = (torch.rand(B, T, 1, 2, generator=g) * 2 - 1) * spread
    xy = c + (torch.rand(B, T, 3, 2, generator=g) * 2 - 1) * size
    z = (torch.rand(B, T, 3, 1, generator=g) * 2 - 1) * 0.9
    w = torch.rand(B, T, 3, 1, generator=g) * (wmax - wmin) + wmin
    pos = torch.cat([xy * w, z * w, w], dim=-1).reshape(B, T * 3, 4).contiguous()
    tri = torch.arange(T * 3, dtype=torch.int32).reshape(T, 3)
    return pos, tri


def decode_aa_flags(flags, B, H, W):
    """The antialias flag planes of include/fpcdr.h (uint64 words [2][B][H][ceil(W/64)], bit x % 64 of word x // 64; plane 0:
    pair (p, p+x) blended, plane 1: pair (p, p+y)) -> [B,H,W] uint8 with bit 0 / bit 1, the oracle's form."""
    Wq = (W + 63) // 64
    words = flags.detach().cpu().reshape(2, B, H, Wq)
    shifts = torch.arange(64, dtype=torch.int64)
    bits = ((words[..., None] >> shifts) & 1).reshape(2, B, H, Wq * 64)[..., :W].to(torch.uint8)
    return bits[0] | (bits[1] << 1)


def near_crossing_soup(B, T, seed, frac=0.4):
    """Triangles in front of a perspective-like camera, a share of them with one or two vertices pushed through the near plane
    (z + w < 0, mostly w < 0 as well): clip-space z = a * w_eye + b with the near plane at w = 1."""
    g = torch.Generator().manual_seed(seed)
    c = (torch.rand(B, T, 1, 3, generator=g) * 2 - 1) * torch.tensor([1.5, 1.5, 0.0]) + torch.tensor([0.0, 0.0, 4.0])
    v = c + (torch.rand(B, T, 3, 3, generator=g) * 2 - 1) * torch.tensor([0.9, 0.9, 1.2])       # eye space, depth along +z
    push = torch.rand(B, T, 3, generator=g) < frac * 0.6
    push[:, ::3] = False                                                                  # a third of the triangles untouched
    v[..., 2] = torch.where(push, -torch.rand(B, T, 3, generator=g) * 3.0 + 0.5, v[..., 2])     # depth in (-2.5, 0.5): behind / near
    n, f = 1.0, 20.0
    w = v[..., 2:3]
    z = (f + n) / (f - n) * w - 2 * f * n / (f - n)
    pos = torch.cat([v[..., :2] * 1.2, z, w], dim=-1).reshape(B, T * 3, 4).contiguous()
    tri = torch.arange(T * 3, dtype=torch.int32).reshape(T, 3)
    return pos, tri


def decode_id_planes(idp, B, H, W):
    """The id planes of fpcdr_objective_fwd (include/fpcdr.h: 1024 uint32 per 32 x 32 bin, bins in (image, bin row, bin column) order,
    pixel (y & 31) * 32 + (x & 31) inside, value (triangle + 1) | silhouette bits << 24) -> triangle + 1 as [B,H,W] int32 (0 = empty)."""
    OY, OX = (H + 31) // 32, (W + 31) // 32
    v = idp.detach().cpu().contiguous().view(torch.int32).reshape(B, OY, OX, 32, 32) & 0xffffff
    return v.permute(0, 1, 3, 2, 4).reshape(B, OY * 32, OX * 32)[:, :H, :W].contiguous()


def comparison_pair(i, height=1600, width=1200):
    """Image pair i of the re-render comparison fixture (tests/golden/make_golden.py::rerender_golden and
    tests/test_next_rows.py regenerate the same 120 pairs from this formula): two smooth 8-bit images of the reference's
    1600 x 1200 shape (comparisons.py:10) that differ by a slow pattern depending on i -- low entropy, so the PNG / TIFF
    files the generator writes are small."""
    y = np.arange(height, dtype=np.int64)[:, None]
    x = np.arange(width, dtype=np.int64)[None, :]
    ref = ((y // 8) * 3 + (x // 16) * 5 + 7 * i) % 200
    img = (ref + ((y // 64 + x // 32 + i) % 9) * ((y // 32 + i) % 3) - ((x // 128 + 2 * i) % 5)).clip(0, 255)
    return img.astype(np.uint8), ref.astype(np.uint8)
