"""
Offline multi-camera re-render and numerical comparison of a finished fit -- forward-only consumers of the same
operators.  Host-side mirror of the reference's src/torch/render_multicam.py:95-169
(read result/{i}.obj + texture + pose.json, render every camera, tile 3 x 3) and comparisons.py:54-81
(mean absolute difference over a crop, one CSV line per image).  The reference's mp4 / GLFW output is out of scope.
"""
import json
import os

import numpy as np
import torch

from . import camera
from . import ops as dr
from .fit import render


def read_result_obj(path):
    """Vertices [V,3] of a result/{i}.obj: the 'v' lines up to the first 'vt' (reference render_multicam.py:121-127)."""
    vertices = []
    with open(path, "r") as f:
        for line in f:
            if line.startswith("v "):
                vertices.extend(float(x) for x in line.strip().split(" ")[1:])
            elif line.startswith("vt "):
                break
    return np.asarray(vertices, dtype=np.float32).reshape(-1, 3)


def read_pose(directory):
    """pose.json of fit.save() (reference fit.py:276-286, read back at render_multicam.py:95-98)."""
    with open(os.path.join(directory, "pose.json"), "r", encoding="utf-8") as f:
        d = json.load(f)
    return np.asarray(d["translation"], dtype=np.float32), np.asarray(d["rotation"], dtype=np.float32)


def read_texture(path):
    """texture.png of fit.save() back to [Ht,Wt,C] float in [0,1] with row 0 at the bottom (it was flipped on save)."""
    from PIL import Image
    img = np.asarray(Image.open(path), dtype=np.float32) / 255.0
    if img.ndim == 2:
        img = img[..., None]
    return np.flip(img, 0).copy()


def make_img(arr, ncols=3):
    """Tile n images [n,H,W,C] into a grid with ncols columns (reference utils.make_img, utils.py:179-190)."""
    n, height, width, nc = arr.shape
    nrows = n // ncols
    assert n == nrows * ncols, "number of images must be a multiple of ncols"
    return arr.reshape(nrows, ncols, height, width, nc).swapaxes(1, 2).reshape(height * nrows, width * ncols, nc)


@torch.no_grad()
def render_multicam(glctx, vertices, pos_idx, uv, uv_idx, tex, cams, resolution, pose=None, modelview_offset=(0.0, 0.0, 0.0)):
    """All cameras of the rig for one mesh: [Nc,H,W,C] in 0..255, top row first (reference render_multicam.py:131-158).

    vertices [V,3] tensor; cams: calib_lookup entries; pose: optional (t [3], q [4]) rigid head pose applied like
    `reproduce_pose` (render_multicam.py:146-150).  One batched launch per operator instead of one render per camera."""
    dev = vertices.device
    mvps = []
    for c in cams:
        proj = camera.intrinsic_to_projection(c['intr'])
        mv = camera.extrinsic_to_modelview(c['rot'], c['trans_calib']) @ camera.translate(*modelview_offset)
        mvps.append((proj, mv))
    proj = torch.tensor(np.stack([p for p, _ in mvps]), dtype=torch.float32, device=dev)
    t_mv = torch.tensor(np.stack([m for _, m in mvps]), dtype=torch.float32, device=dev)
    if pose is not None:
        t, q = (torch.as_tensor(a, dtype=torch.float32, device=dev) for a in pose)
        t_mv = torch.matmul(camera.rigid_grad(t, camera.unitquat_to_rotmat(q))[None], t_mv)
    mvp = torch.matmul(proj, t_mv)
    colour = render(glctx, mvp, vertices[None], pos_idx, uv, uv_idx, tex, resolution, False, 0) * 255.0
    return torch.flip(colour, dims=[1])      # row 0 = bottom in the raster -> top row first on disk


def mean_abs_diff(img, ref, rows=(200, 1401), cols=(100, 1100)):
    """(image mean, per-row means) of |img - ref| over a crop (reference comparisons.py:66-75: rows 200..1400 inclusive,
    columns 100..1099, integer arithmetic).  The crop is clipped to the image."""
    a = np.asarray(img).astype(np.int32)
    b = np.asarray(ref).astype(np.int32)
    r0, r1 = max(rows[0], 0), min(rows[1], a.shape[0])
    c0, c1 = max(cols[0], 0), min(cols[1], a.shape[1])
    row_means = np.abs(a[r0:r1, c0:c1] - b[r0:r1, c0:c1]).reshape(r1 - r0, -1).mean(axis=1)
    return float(row_means.mean()), row_means


def compare_sequence_numerical(inferred, references, csv_path, **crop):
    """One CSV line per image 'mean, row means...' and the mean of means last (reference comparisons.py:56-80).
    inferred / references: sequences of arrays.  Returns the list of image means."""
    os.makedirs(os.path.dirname(os.path.abspath(csv_path)), exist_ok=True)
    means = []
    with open(csv_path, "w") as f:
        for img, ref in zip(inferred, references):
            m, rows = mean_abs_diff(img, ref, **crop)
            means.append(m)
            f.write(f"{m}, {', '.join(str(x) for x in rows)}\n")
        f.write(str(float(np.mean(means))))
    return means


@torch.no_grad()
def rerender_result(result_dir, sc, device='cuda', frames=None, reproduce_pose=True, ncols=3):
    """Re-render a directory written by Fitter.save(): yields (frame index, grid image uint8 [3H,3W,C]).
    sc supplies what the reference re-reads from the take (index buffers, uv, cameras, resolution)."""
    dev = torch.device(device)
    glctx = dr.RasterizeGLContext(device=dev)
    pos_idx = torch.tensor(sc.pos_idx, dtype=torch.int32, device=dev)
    uv = torch.tensor(sc.uv, dtype=torch.float32, device=dev)
    uv_idx = torch.tensor(sc.uv_idx, dtype=torch.int32, device=dev)
    tex = torch.tensor(read_texture(os.path.join(result_dir, "texture.png")), dtype=torch.float32, device=dev)
    t_all, q_all = read_pose(result_dir) if reproduce_pose else (None, None)
    n = len([f for f in os.listdir(result_dir) if f.endswith(".obj") and f[:-4].isdigit()])
    for i in (frames if frames is not None else range(n)):
        verts = torch.tensor(read_result_obj(os.path.join(result_dir, f"{i}.obj")), device=dev)
        pose = (t_all[i], q_all[i]) if reproduce_pose else None
        imgs = render_multicam(glctx, verts, pos_idx, uv, uv_idx, tex, sc.cams, sc.resolution, pose=pose,
                               modelview_offset=(0.0, 170.0, 0.0))
        grid = make_img(imgs.cpu().numpy(), ncols=ncols)
        yield i, np.clip(np.rint(grid), 0, 255).astype(np.uint8)
