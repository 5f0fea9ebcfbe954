"""
Dataset ingest -- host-side mirror of the reference's src/torch/data.py and of the loaders inside fitTake
(reference fit.py:199-220 blendshape OBJs, :514-521 calibration lookup, :529-533 reference images).  SURVEY.md section
8(f-1).  The OBJ semantics are pinned against the reference by tests/golden/meshdata_golden.json.
"""
import json
import os

import numpy as np


class MeshData:
    """Wavefront .obj reader with the reference's semantics (data.py:7-39): `v x y z` -> flat vertices[3V];
    `vt u v` -> uv[Vt,2]; `f a/ta b/tb c/tc` (triangles only, 1-based) -> faces[T,3], fuv[T,3] (0-based)."""

    def __init__(self, obj):
        vertices, uv, faces, fuv = [], [], [], []
        with open(obj, "r") as f:
            for line in f:
                if line.startswith("v "):
                    vertices.extend(float(x) for x in line.strip().split(" ")[1:])
                elif line.startswith("vt "):
                    uv.append([float(x) for x in line.strip().split(" ")[1:]])
                elif line.startswith("f "):
                    corners = [c.split("/") for c in line.strip().split(" ")[1:]]
                    assert len(corners) == 3, "only triangles are supported (as in the reference)"
                    faces.append([int(c[0]) - 1 for c in corners])
                    fuv.append([int(c[1]) - 1 for c in corners])
        self.vertices = np.asarray(vertices, dtype=np.float32)
        self.uv = np.asarray(uv, dtype=np.float32)
        self.faces = np.asarray(faces, dtype=np.int32)
        self.fuv = np.asarray(fuv, dtype=np.int32)


def load_blendshape_deltas(directory, v_basemesh):
    """Reference fit.py:199-220: every OBJ of `directory` (os.listdir order, as the reference) minus the base mesh,
    returned as B[3V,K] float32 (the reference's datasets['local'])."""
    objs = os.listdir(directory)
    out = np.empty((len(objs), v_basemesh.shape[0]), dtype=np.float32)
    for i, name in enumerate(objs):
        verts = []
        with open(os.path.join(directory, name), "r") as f:
            for line in f:
                if line.startswith("v "):
                    verts.extend(float(x) for x in line.strip().split(" ")[1:])
        out[i] = np.subtract(np.asarray(verts, dtype=np.float32), v_basemesh)
    return np.ascontiguousarray(out.transpose())


def load_reference_image(path):
    """Reference fit.py:529-533: 8-bit image clipped to [0,140], rows flipped to the OpenGL convention; uint8 [H,W]."""
    from PIL import Image
    img = np.array(Image.open(path))
    img = np.clip(img, 0, 140)
    return np.flip(img, 0).astype(np.uint8).copy()


def load_calibration(calibpath, camera_dirs):
    """Reference fit.py:514-521: calibration.json looked up by the second '_' field of each camera directory name."""
    with open(calibpath) as f:
        calibs = json.load(f)
    lookup = []
    for cam in camera_dirs:
        c = calibs[cam.split("_")[1]]
        lookup.append({'cam': cam, 'intr': np.asarray(c['intrinsic'], dtype=np.float32),
                       'dist': np.asarray(c['distortion'], dtype=np.float32), 'rot': np.asarray(c['rotation'], dtype=np.float32),
                       'trans_calib': np.asarray(c['translation'], dtype=np.float32)})
    return lookup


def assert_num_frames(cams, imdir):
    """Reference fit.py:29-43: every camera directory holds the same number of frames; zero-pad width 2 below 100 else 3."""
    n = [len(os.listdir(os.path.join(imdir, c))) for c in cams]
    assert not any(x != n[0] for x in n), "All cameras do not have the same number of frames!"
    return (n[0], 2) if n[0] < 100 else (n[0], 3)
