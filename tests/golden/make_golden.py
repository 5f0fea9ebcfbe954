"""
Generates the committed golden fixtures in this directory.  Runs ONLY in the build container,
where the reference checkout is mounted at /root/reference (it does not exist on the GPU box and
nothing in tests/, smoke() or bench.py reads it at run time).

What can be pinned against the reference by import: the numpy camera
functions of src/torch/camera.py (intrinsic_to_projection :27-41, extrinsic_to_modelview :46-66,
translate :108-112) and the OBJ reader src/torch/data.py MeshData :7-39.  The four raster ops
themselves live in the absent third-party package nvdiffrast -> no vectors exist for them.

Outputs (data only -- inputs and expected outputs, no reference source text):
    camera_golden.json   per camera of calibration/calibration.json: inputs (intrinsic, rotation,
                         translation) and the reference's P and MV 4x4 float32 matrices; plus
                         translate(0,170,0)
    meshdata_golden.json a tiny hand-written OBJ (text) and the arrays MeshData parses from it

    python tests/golden/make_golden.py
"""
import json
import os
import sys
import tempfile

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))

TINY_OBJ = """# tiny fixture: a quad split into two triangles with a uv seam
v 0.0 0.0 0.0
v 1.0 0.0 0.5
v 1.0 1.0 0.0
v 0.0 1.0 -0.25
vt 0.0 0.0
vt 1.0 0.0
vt 1.0 1.0
vt 0.0 1.0
vt 0.5 0.5
f 1/1 2/2 3/3
f 1/5 3/3 4/4
"""


def main():
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    import src.torch.camera as rcam  # noqa: E402  (reference, numpy-only functions)
    import src.torch.data as rdata  # noqa: E402

    with open(os.path.join(REF, "calibration", "calibration.json")) as f:
        calibs = json.load(f)
    cams = {}
    for name, c in calibs.items():
        intr = np.asarray(c["intrinsic"], dtype=np.float32)
        rot = np.asarray(c["rotation"], dtype=np.float32)
        trans = np.asarray(c["translation"], dtype=np.float32)
        P = rcam.intrinsic_to_projection(intr)
        MV = rcam.extrinsic_to_modelview(rot, trans)
        cams[name] = {
            "intrinsic": np.asarray(c["intrinsic"], dtype=np.float64).tolist(),
            "rotation": np.asarray(c["rotation"], dtype=np.float64).tolist(),
            "translation": np.asarray(c["translation"], dtype=np.float64).tolist(),
            "P": np.asarray(P, dtype=np.float32).astype(np.float64).tolist(),
            "MV": np.asarray(MV, dtype=np.float32).astype(np.float64).tolist(),
            "P_dtype": str(P.dtype), "MV_dtype": str(MV.dtype),
        }
    out = {"cameras": cams,
           "translate_0_170_0": rcam.translate(0.0, 170.0, 0.0).astype(np.float64).tolist(),
           "default_projection": rcam.default_projection().astype(np.float64).tolist(),
           "default_modelview": rcam.default_modelview().astype(np.float64).tolist(),
           "rotate_x_0p3": rcam.rotate_x(0.3).astype(np.float64).tolist(),
           "rotate_y_0p3": rcam.rotate_y(0.3).astype(np.float64).tolist()}
    with open(os.path.join(HERE, "camera_golden.json"), "w") as f:
        json.dump(out, f, indent=1)

    with tempfile.NamedTemporaryFile("w", suffix=".obj", delete=False) as tf:
        tf.write(TINY_OBJ)
        path = tf.name
    md = rdata.MeshData(path)
    os.unlink(path)
    with open(os.path.join(HERE, "meshdata_golden.json"), "w") as f:
        json.dump({"obj_text": TINY_OBJ,
                   "vertices": md.vertices.astype(np.float64).tolist(), "vertices_dtype": str(md.vertices.dtype),
                   "uv": md.uv.astype(np.float64).tolist(), "uv_dtype": str(md.uv.dtype),
                   "faces": md.faces.tolist(), "faces_dtype": str(md.faces.dtype),
                   "fuv": md.fuv.tolist(), "fuv_dtype": str(md.fuv.dtype)}, f, indent=1)
    print("wrote camera_golden.json, meshdata_golden.json")


if __name__ == "__main__":
    main()
