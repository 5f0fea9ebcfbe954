"""
Camera matrices and the clip-space transform -- host-side mirror of the reference's
src/torch/camera.py, same function names and argument meaning so the reference's
render()/fitTake read unchanged (reference fit.py:150, 541-553).

Differences from the reference, all additive:
  * tensors follow the device of their inputs instead of hard-coding 'cuda'
    (reference camera.py:21,118,130);
  * `transform_clip` also accepts a batch of matrices [B,4,4] and/or a batch of
    vertex buffers [F,V,3] (the reference is batch-1, camera.py:22);
  * `unitquat_to_rotmat` restates roma.unitquat_to_rotmat (XYZW order), which the
    reference imports from the absent third-party package roma (fit.py:548,550).
"""
import json
import os

import numpy as np
import torch

_DATA = os.path.dirname(os.path.abspath(__file__))


def transform_clip(mvp, pos):
    """pos [V,3] (or [F,V,3]) -> clip space [1,V,4] (or [B,V,4]);  reference camera.py:11-23.

    mvp: np.ndarray or tensor, [4,4] or [B,4,4].  With pos [F,V,3] and mvp [F*C,4,4] the views of a
    frame are consecutive: image b uses vertex buffer b // (B // F).
    """
    if isinstance(mvp, np.ndarray):
        mvp = torch.from_numpy(mvp).to(pos.device)
    ones = torch.ones(pos.shape[:-1] + (1,), dtype=pos.dtype, device=pos.device)
    posw = torch.cat([pos, ones], dim=-1)
    if mvp.dim() == 2 and pos.dim() == 2:
        return torch.matmul(posw, mvp.t())[None, ...]
    if mvp.dim() == 2:
        mvp = mvp[None]
    if posw.dim() == 2:
        posw = posw[None]
    B, F = mvp.shape[0], posw.shape[0]
    if B != F:
        assert B % F == 0, "number of matrices must be a multiple of the number of vertex buffers"
        posw = posw.repeat_interleave(B // F, dim=0)
    return torch.matmul(posw, mvp.transpose(1, 2)).contiguous()


def intrinsic_to_projection(intr=None, zn=0.01, zf=200):
    """OpenGL projection from a 3x3 pixel intrinsic matrix; reference camera.py:27-41.

    Assumes a centred principal point (fx/cx, fy/cy), as the reference does.
    """
    m = np.zeros((4, 4), dtype=np.float64)
    m[0, 0] = intr[0, 0] / intr[0, 2]
    m[1, 1] = intr[1, 1] / intr[1, 2]
    m[2, 2] = -(zf + zn) / (zf - zn)
    m[2, 3] = -(2 * zf * zn) / (zf - zn)
    m[3, 2] = -1.0
    return m.astype(np.float32)


def extrinsic_to_modelview(rmat=None, tvec=None):
    """OpenGL modelview from OpenCV extrinsics: [R|t] with camera y and z flipped; reference camera.py:46-66."""
    rmat = np.asarray(rmat)
    tvec = np.asarray(tvec).reshape(3, 1)
    out = np.eye(4, dtype=np.result_type(rmat.dtype, tvec.dtype, np.float32))
    out[:3, :3] = rmat
    out[:3, 3:] = tvec
    out[1:3, :] *= -1
    return out


def default_projection(xn=1.0, xf=50.0, x=0.1):
    """reference camera.py:70-74"""
    m = np.zeros((4, 4), dtype=np.float64)
    m[0, 0] = xn / x
    m[1, 1] = xn / -x
    m[2, 2] = -(xf + xn) / (xf - xn)
    m[2, 3] = -(2 * xf * xn) / (xf - xn)
    m[3, 2] = -1.0
    return m.astype(np.float32)


def default_modelview(zoffset=-30):
    """reference camera.py:79-83"""
    return translate(0, 0, zoffset)


def rotate_y(a):
    """reference camera.py:88-93"""
    s, c = np.sin(a), np.cos(a)
    m = np.eye(4)
    m[0, 0], m[0, 2], m[2, 0], m[2, 2] = c, s, -s, c
    return m.astype(np.float32)


def rotate_x(a):
    """reference camera.py:98-103"""
    s, c = np.sin(a), np.cos(a)
    m = np.eye(4)
    m[1, 1], m[1, 2], m[2, 1], m[2, 2] = c, s, -s, c
    return m.astype(np.float32)


def translate(x, y, z):
    """reference camera.py:108-112"""
    m = np.eye(4, dtype=np.float32)
    m[:3, 3] = (x, y, z)
    return m


def translate_tensor(vec):
    """4x4 translation keeping autograd on vec; reference camera.py:117-123"""
    eye = torch.eye(4, dtype=torch.float32, device=vec.device)
    col = torch.cat([torch.flatten(vec), torch.zeros(1, dtype=vec.dtype, device=vec.device)])
    return eye + torch.outer(col, eye[3])


def rigid_grad(tvec, rotmat):
    """[R|t; 0 0 0 1] keeping autograd; reference camera.py:128-132.  Batched: tvec [...,3], rotmat [...,3,3]."""
    top = torch.cat([rotmat, tvec.reshape(rotmat.shape[:-2] + (3, 1))], dim=-1)
    bottom = torch.zeros(rotmat.shape[:-2] + (1, 4), dtype=rotmat.dtype, device=rotmat.device)
    bottom[..., 0, 3] = 1.0
    return torch.cat([top, bottom], dim=-2)


def unitquat_to_rotmat(quat):
    """XYZW quaternion(s) [...,4] -> rotation matrices [...,3,3] (roma convention; reference fit.py:548,550).

    Like roma, no normalisation is applied: the reference feeds it non-unit quaternions after its
    whole-tensor "renormalisation" (fit.py:616-618, quirk Q3 in SURVEY.md section 8a-9).
    """
    x, y, z, w = quat[..., 0], quat[..., 1], quat[..., 2], quat[..., 3]
    tx, ty, tz = 2.0 * x, 2.0 * y, 2.0 * z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    one = torch.ones_like(x)
    m = torch.stack([one - (tyy + tzz), txy - twz, txz + twy,
                     txy + twz, one - (txx + tzz), tyz - twx,
                     txz - twy, tyz + twx, one - (txx + tyy)], dim=-1)
    return m.reshape(quat.shape[:-1] + (3, 3))


def load_rig(path=None):
    """The 9-camera rig of the reference's calibration/calibration.json (the only real data it ships).

    Returns a list of dicts {'cam', 'intr', 'dist', 'rot', 'trans_calib'} like the reference's
    calib_lookup (fit.py:514-521).
    """
    with open(path or os.path.join(_DATA, "rig9.json")) as f:
        rig = json.load(f)["cameras"]
    out = []
    for name, c in rig.items():
        out.append({'cam': name,
                    'intr': np.asarray(c['intrinsic'], dtype=np.float32),
                    'dist': np.zeros((5, 1), dtype=np.float32),
                    'rot': np.asarray(c['rotation'], dtype=np.float32),
                    'trans_calib': np.asarray(c['translation'], dtype=np.float32)})
    return out
