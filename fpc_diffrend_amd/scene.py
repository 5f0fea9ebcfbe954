"""
Seeded synthetic take: the reference ships no mesh, blendshape, texture or image data
(its .gitignore:10-11 drops *.obj and data/), only the 9-camera
calibration.  Everything the fit loop consumes (reference fit.py:424-461, 514-533) is
generated here with numpy.random.default_rng(seed):

  mesh        UV-sphere deformed to an 8 x 11 x 9 cm ellipsoid "head", separate uv / uv_idx
              with a seam (Vt > V) like an OBJ with v/vt indices (reference data.py:7-39)
  blendshapes K smooth bumps  A exp(-|x - c|^2 / 2 sigma^2) n   -> B[3V,K]  (fit.py:199-220)
  texture     [Ht,Wt,C] value noise + grid
  cameras     the 9 extrinsics of the reference rig; intrinsics rebuilt for the target raster
              with a centred principal point (reference camera.py:38-39 assumes one)
  weights     per-frame ground-truth activations, sparse and temporally smooth; small per-frame pose
"""
import os
from dataclasses import dataclass
from typing import Optional

import numpy as np

from . import camera, data

# BASELINE.json configs -> (n_lon, n_rings): T = 2 n_lon + (n_rings - 1) 2 n_lon, V = n_rings n_lon + 2
MESH_1K = (32, 16)      # T = 1024,  V = 514
MESH_30K = (125, 120)   # T = 30000, V = 15002


@dataclass
class Scene:
    v_base: np.ndarray      # [3V] float32, (x,y,z,x,...) like reference data.MeshData.vertices
    pos_idx: np.ndarray     # [T,3] int32
    uv: np.ndarray          # [Vt,2] float32
    uv_idx: np.ndarray      # [T,3] int32
    blendshapes: np.ndarray  # [3V,K] float32  (reference datasets['local'], fit.py:216-217)
    texture: np.ndarray     # [Ht,Wt,C] float32 in [0,1]
    cams: list              # calib_lookup entries (fit.py:514-521) with rebuilt 'intr'
    resolution: tuple       # (H, W)
    weights_gt: Optional[np.ndarray] = None  # [F,K]    hidden ground truth of a synthetic take (None for a take
    t_gt: Optional[np.ndarray] = None        # [F,3]    read from disk: its reference images are in `images`)
    q_gt: Optional[np.ndarray] = None        # [F,4] XYZW
    images: Optional[np.ndarray] = None      # [F,Ncam,H,W] uint8: reference images as the reference's loader leaves them
                                             # (fit.py:529-533: clipped to [0,140], row 0 = bottom)
    frames: int = 0                          # number of frames when there is no ground truth to count

    @property
    def n_vertices(self):
        return self.v_base.shape[0] // 3

    @property
    def n_frames(self):
        if self.weights_gt is not None:
            return self.weights_gt.shape[0]
        return self.images.shape[0] if self.images is not None else self.frames


def make_mesh(n_lon, n_rings, radii=(8.0, 11.0, 9.0)):
    """Closed UV-sphere: poles + n_rings rings of n_lon vertices; y is up."""
    rx, ry, rz = radii
    verts = [(0.0, ry, 0.0)]
    for r in range(n_rings):
        th = np.pi * (r + 1) / (n_rings + 1)
        for l in range(n_lon):
            ph = 2 * np.pi * l / n_lon
            verts.append((rx * np.sin(th) * np.sin(ph), ry * np.cos(th), rz * np.sin(th) * np.cos(ph)))
    verts.append((0.0, -ry, 0.0))
    verts = np.asarray(verts, dtype=np.float32)
    south = verts.shape[0] - 1

    def vid(r, l):
        return 1 + r * n_lon + (l % n_lon)

    # uv grid with a duplicated seam column and one pole vertex per column
    def uvid_pole_n(l):
        return l

    def uvid(r, l):
        return n_lon + r * (n_lon + 1) + l

    def uvid_pole_s(l):
        return n_lon + n_rings * (n_lon + 1) + l

    uv = []
    for l in range(n_lon):
        uv.append(((l + 0.5) / n_lon, 1.0))
    for r in range(n_rings):
        for l in range(n_lon + 1):
            uv.append((l / n_lon, 1.0 - (r + 1) / (n_rings + 1)))
    for l in range(n_lon):
        uv.append(((l + 0.5) / n_lon, 0.0))
    uv = np.asarray(uv, dtype=np.float32)

    tris, tuvs = [], []
    for l in range(n_lon):
        tris.append((0, vid(0, l), vid(0, l + 1)))
        tuvs.append((uvid_pole_n(l), uvid(0, l), uvid(0, l + 1)))
    for r in range(n_rings - 1):
        for l in range(n_lon):
            a, b, c, d = vid(r, l), vid(r + 1, l), vid(r + 1, l + 1), vid(r, l + 1)
            ua, ub, uc, ud = uvid(r, l), uvid(r + 1, l), uvid(r + 1, l + 1), uvid(r, l + 1)
            tris.append((a, b, c)); tuvs.append((ua, ub, uc))
            tris.append((a, c, d)); tuvs.append((ua, uc, ud))
    for l in range(n_lon):
        tris.append((south, vid(n_rings - 1, l + 1), vid(n_rings - 1, l)))
        tuvs.append((uvid_pole_s(l), uvid(n_rings - 1, l + 1), uvid(n_rings - 1, l)))
    return verts, np.asarray(tris, dtype=np.int32), uv, np.asarray(tuvs, dtype=np.int32)


def make_blendshapes(verts, K, rng):
    V = verts.shape[0]
    radii = np.abs(verts).max(axis=0)
    normals = verts / (radii ** 2)
    normals /= np.linalg.norm(normals, axis=1, keepdims=True)
    B = np.zeros((V, 3, K), dtype=np.float32)
    for k in range(K):
        c = verts[rng.integers(0, V)]
        amp = rng.uniform(0.3, 1.0) * rng.choice([-1.0, 1.0])
        sigma = rng.uniform(1.5, 4.0)
        d2 = ((verts - c) ** 2).sum(axis=1)
        g = amp * np.exp(-d2 / (2 * sigma * sigma))
        direction = normals + 0.3 * rng.normal(size=3)
        B[:, :, k] = (g[:, None] * direction).astype(np.float32)
    return B.reshape(V * 3, K)


def make_texture(Ht, Wt, C, rng):
    """Value noise over three octaves plus a faint grid; [Ht,Wt,C] in [0.05, 0.5]: the reference clips its
    reference images to [0,140] of 255 (fit.py:531), so a renderable target must stay below 140/255."""
    tex = np.zeros((Ht, Wt), dtype=np.float64)
    yy, xx = np.meshgrid(np.arange(Ht) / Ht, np.arange(Wt) / Wt, indexing='ij')
    for octave, amp in ((4, 0.5), (16, 0.3), (64, 0.2)):
        g = rng.uniform(0, 1, size=(octave + 1, octave + 1))
        g[-1, :] = g[0, :]
        g[:, -1] = g[:, 0]
        fy, fx = yy * octave, xx * octave
        iy, ix = np.floor(fy).astype(int), np.floor(fx).astype(int)
        ty, tx = fy - iy, fx - ix
        ty, tx = ty * ty * (3 - 2 * ty), tx * tx * (3 - 2 * tx)
        v = (g[iy, ix] * (1 - tx) + g[iy, ix + 1] * tx) * (1 - ty) + (g[iy + 1, ix] * (1 - tx) + g[iy + 1, ix + 1] * tx) * ty
        tex += amp * v
    grid = ((np.floor(yy * 32) + np.floor(xx * 32)) % 2) * 0.12
    tex = np.clip(0.15 + 0.7 * tex + grid - 0.06, 0.0, 1.0)
    out = np.stack([np.clip(tex * (1.0 - 0.08 * c) + 0.03 * c, 0, 1) for c in range(C)], axis=-1)
    return (0.05 + 0.45 * out).astype(np.float32)


def make_cameras(resolution, fill=0.6, head_height=22.0, target=(0.0, 170.0, 0.0)):
    """Reference rig extrinsics; focal length per camera so the head spans `fill` of the image height."""
    H, W = resolution
    cams = camera.load_rig()
    tgt = np.asarray(target, dtype=np.float64)
    for c in cams:
        pc = c['rot'].astype(np.float64) @ tgt + c['trans_calib'].astype(np.float64).ravel()
        dist = float(np.linalg.norm(pc))
        f = fill * H * dist / head_height
        c['intr'] = np.array([[f, 0, W / 2.0], [0, f, H / 2.0], [0, 0, 1]], dtype=np.float32)
    return cams


def make_motion(F, K, rng):
    """Sparse (about 10 % active), temporally smooth activations in [0,1]; small per-frame pose."""
    active = rng.uniform(size=K) < 0.10
    if not active.any():
        active[rng.integers(0, K)] = True
    w = np.zeros((F, K), dtype=np.float64)
    for k in np.nonzero(active)[0]:
        walk = np.cumsum(rng.normal(scale=0.15, size=F + 8))
        walk = np.convolve(walk, np.ones(8) / 8.0, mode='valid')[:F]
        walk = (walk - walk.min()) / max(walk.max() - walk.min(), 1e-6)
        w[:, k] = walk * rng.uniform(0.4, 1.0)
    t = rng.normal(scale=0.3, size=(F, 3))
    ang = np.deg2rad(rng.uniform(0, 3.0, size=F))
    axis = rng.normal(size=(F, 3))
    axis /= np.linalg.norm(axis, axis=1, keepdims=True)
    q = np.concatenate([axis * np.sin(ang / 2)[:, None], np.cos(ang / 2)[:, None]], axis=1)
    return w.astype(np.float32), t.astype(np.float32), q.astype(np.float32)


def make_scene(mesh=MESH_1K, K=10, n_frames=4, resolution=(256, 256), texshape=(256, 256, 1), seed=0):
    rng = np.random.default_rng(seed)
    verts, tris, uv, tuvs = make_mesh(*mesh)
    B = make_blendshapes(verts, K, rng)
    tex = make_texture(texshape[0], texshape[1], texshape[2], rng)
    cams = make_cameras(resolution)
    w, t, q = make_motion(n_frames, K, rng)
    return Scene(v_base=verts.reshape(-1).copy(), pos_idx=tris, uv=uv, uv_idx=tuvs, blendshapes=B, texture=tex,
                 cams=cams, resolution=tuple(resolution), weights_gt=w, t_gt=t, q_gt=q)


def from_take(basemeshpath, localblpath, calibpath, imdir, texpath="", texshape=(1024, 1024, 1), seed=0):
    """A take on disk in the reference's layout -> Scene (reference fit.py:415-439, 461, 514-533):

      basemeshpath  base mesh .obj (v / vt / f v/vt triangles)                       fit.py:424-432, data.py:7-39
      localblpath   directory of blendshape .obj files (same vertex order)           fit.py:199-220
      calibpath     calibration.json keyed by camera name                            fit.py:419-420, 514-521
      imdir         one directory per camera, named <x>_<camera name>..., holding
                    <dir>_<frame:0{digits}d>.tif, the same number in every directory fit.py:415-416, 29-43, 528-530
      texpath       optional start texture (8-bit image); else uniform noise         fit.py:433-438

    The reference decodes ONE image from disk per iteration; here every frame of every camera is read once (clipped to
    [0,140] and flipped like fit.py:531-532) so that the fit loop finds them resident in HBM as 8-bit.  Camera
    directories are taken in sorted order (the reference uses os.listdir order, which is file-system dependent)."""
    cams = sorted(os.listdir(imdir))
    n_frames, digits = data.assert_num_frames(cams, imdir)
    base = data.MeshData(basemeshpath)
    lookup = data.load_calibration(calibpath, cams)
    first = data.load_reference_image(os.path.join(imdir, cams[0], f"{cams[0]}_{0:0{digits}d}.tif"))
    H, W = first.shape[:2]
    images = np.empty((n_frames, len(cams), H, W), dtype=np.uint8)
    for c, cam in enumerate(cams):
        for f in range(n_frames):
            img = data.load_reference_image(os.path.join(imdir, cam, f"{cam}_{f:0{digits}d}.tif"))
            assert img.shape[:2] == (H, W), f"{cam} frame {f}: {img.shape[:2]} differs from {(H, W)}"
            images[f, c] = img if img.ndim == 2 else img[..., 0]
    if texpath:
        from PIL import Image
        tex = np.array(Image.open(texpath)) / 255.0                      # fit.py:434-436
        tex = np.flip(tex[..., np.newaxis] if tex.ndim == 2 else tex, 0)
    else:
        tex = np.random.default_rng(seed).uniform(low=0.0, high=1.0, size=texshape)   # fit.py:438 (seeded here)
    blend = data.load_blendshape_deltas(localblpath, base.vertices)
    return Scene(v_base=base.vertices, pos_idx=base.faces, uv=base.uv, uv_idx=base.fuv, blendshapes=blend,
                 texture=np.ascontiguousarray(tex, dtype=np.float32), cams=lookup, resolution=(H, W), images=images)


def write_take(sc, directory, images, cam_idxs=None, blendshape_scale=1.0):
    """Write a Scene + reference images [F,Nc,H,W] uint8 (row 0 = bottom) to `directory` in the layout `from_take`
    reads (the reference's, fit.py:415-432, 514-533): basemesh.obj, blendshapes/*.obj, calibration.json,
    images/cam_<name>/cam_<name>_<frame>.tif.  Returns the four paths.  (Test / example helper: the reference ships
    no data.)"""
    import json
    from PIL import Image
    cam_idxs = list(range(len(sc.cams))) if cam_idxs is None else list(cam_idxs)
    os.makedirs(directory, exist_ok=True)
    V = sc.n_vertices

    def write_obj(path, verts):
        with open(path, "w") as f:
            for v in np.asarray(verts, dtype=np.float32).reshape(-1, 3):
                f.write(f"v {float(v[0])!r} {float(v[1])!r} {float(v[2])!r}\n")
            for u in sc.uv:
                f.write(f"vt {float(u[0])!r} {float(u[1])!r}\n")
            for fv, ft in zip(sc.pos_idx, sc.uv_idx):
                f.write("f " + " ".join(f"{int(a) + 1}/{int(b) + 1}" for a, b in zip(fv, ft)) + "\n")

    base = os.path.join(directory, "basemesh.obj")
    write_obj(base, sc.v_base)
    bldir = os.path.join(directory, "blendshapes")
    os.makedirs(bldir, exist_ok=True)
    for k in range(sc.blendshapes.shape[1]):
        write_obj(os.path.join(bldir, f"shape_{k:04d}.obj"), sc.v_base + blendshape_scale * sc.blendshapes[:, k])
    calib = {}
    for c in cam_idxs:
        cam = sc.cams[c]
        calib[cam['cam']] = {'intrinsic': np.asarray(cam['intr'], dtype=np.float64).tolist(), 'distortion': np.zeros((5, 1)).tolist(),
                             'rotation': np.asarray(cam['rot'], dtype=np.float64).tolist(),
                             'translation': np.asarray(cam['trans_calib'], dtype=np.float64).reshape(3, 1).tolist()}
    calibpath = os.path.join(directory, "calibration.json")
    with open(calibpath, "w") as f:
        json.dump(calib, f)
    imdir = os.path.join(directory, "images")
    F = images.shape[0]
    digits = 2 if F < 100 else 3
    for j, c in enumerate(cam_idxs):
        name = f"cam_{sc.cams[c]['cam']}"
        os.makedirs(os.path.join(imdir, name), exist_ok=True)
        for fr in range(F):
            Image.fromarray(np.flip(np.asarray(images[fr, j]), 0)).save(os.path.join(imdir, name, f"{name}_{fr:0{digits}d}.tif"))
    return base, bldir, calibpath, imdir


def cfg(name, n_frames=None, seed=0):
    """Scenes of BASELINE.json's configs."""
    if name == 'cfg1':
        return make_scene(MESH_1K, 10, n_frames or 4, (256, 256), (256, 256, 1), seed)
    if name in ('cfg2', 'cfg3', 'cfg4'):
        nf = n_frames or (1 if name == 'cfg2' else 32)
        return make_scene(MESH_30K, 150, nf, (1080, 1920), (1024, 1024, 1), seed)
    if name == 'ref':      # the reference's own run shape (main.py:28-30): ONE 1600 x 1200 image per step, 1024^2 x 1 texture
        return make_scene(MESH_30K, 150, n_frames or 4, (1600, 1200), (1024, 1024, 1), seed)
    if name == 'cfg5':
        return make_scene(MESH_30K, 150, n_frames or 4, (2160, 3840), (1024, 1024, 1), seed)
    raise ValueError(name)
