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
    blend_golden.json    seeded inputs and the outputs + autograd gradients of the reference's blend / blend_free /
                         blend_combined (src/torch/fit.py:47-129).  `import src.torch.fit` fails on the absent roma /
                         nvdiffrast / pytorch3d (ordinary ModuleNotFoundErrors), but the three functions are
                         self-contained torch code: their FunctionDef nodes are taken out of the parsed file (ast),
                         compiled as they are and run on the CPU with `torch` as their only global.  No stand-in for
                         any missing library is involved and no source text is stored.
    numframes_golden.json  assertNumFrames (fit.py:29-43, os only; taken out the same way) on two throw-away
                         directory trees: the (count, zero-pad digits) pairs and the assertion on unequal counts
    save_golden.json     save (fit.py:235-286: os / json / codecs / numpy; taken out of the file's syntax tree like the blend functions) on two
                         small meshes: the text of the OBJ files and of pose.json as the reference's own function wrote them.  Its
                         texture.png goes through imageio, which is absent: the function's own try / except reports that and carries
                         on, so the texture file is NOT pinned (recorded as such)
    rerender_golden.json make_img (src/torch/utils.py:179-190, numpy only; `import src.torch.utils` fails on the absent cv2) on
                         seeded image stacks, and compareSequenceNumerical (src/torch/comparisons.py:54-81, numpy + PIL; the module
                         runs a comparison on the author's W: drive at import) on 120 synthetic 1600 x 1200 image pairs
                         (tests/helpers.py::comparison_pair) written to a temporary directory under the file names the function
                         reads: the 120 image means, the final mean, the first row means of three lines and a hash of the whole CSV

    python tests/golden/make_golden.py
"""
import ast
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
    blend_golden()
    numframes_golden()
    rerender_golden()
    save_golden()
    print("wrote camera_golden.json, meshdata_golden.json, blend_golden.json, numframes_golden.json, rerender_golden.json, save_golden.json")


def reference_functions(names, extra_globals, module="fit.py"):
    """The named top-level functions of the reference's src/torch/<module>, compiled from the file's own syntax tree (the modules
    themselves cannot be imported: roma, nvdiffrast, pytorch3d, torchvision, imageio and cv2 are absent)."""
    path = os.path.join(REF, "src", "torch", module)
    with open(path) as f:
        tree = ast.parse(f.read(), filename=path)
    nodes = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert sorted(n.name for n in nodes) == sorted(names), [n.name for n in nodes]
    ns = dict(extra_globals)
    exec(compile(ast.Module(body=nodes, type_ignores=[]), path, "exec"), ns)
    return {n: ns[n] for n in names}, {n.name: (n.lineno, n.end_lineno) for n in nodes}


def blend_golden():
    import torch
    fns, lines = reference_functions(("blend", "blend_free", "blend_combined"), {"torch": torch})
    rng = np.random.default_rng(7)
    M, K, F = 3 * 43, 7, 5                 # 3V (not a multiple of the MFMA tile), blendshapes, frames
    f32 = lambda a: np.asarray(a, dtype=np.float32)
    inp = {"v_base": f32(rng.normal(size=M) * 5), "Bmat": f32(rng.normal(size=(M, K))), "M1": f32(rng.normal(size=(F, F)) * 0.5 + np.eye(F)),
           "M2": f32(rng.normal(size=(K, F)) * 0.4), "m1": f32(rng.normal(size=(F, F)) * 0.3 + np.eye(F)),
           "m2": f32(rng.normal(size=(F, F)) * 0.3 + np.eye(F)), "m3": f32(rng.normal(size=(M, F)) * 0.2),
           "gy": f32(rng.normal(size=(F, M))), "learned_coefficient": 0.5}        # fit.py:562 passes the literal 0.5
    out = {"inputs": {k: (v.astype(np.float64).tolist() if isinstance(v, np.ndarray) else v) for k, v in inp.items()},
           "reference_lines": lines, "cases": {}}

    def run(name, call, leaves):
        """One frame at a time, as the reference does (frames = one-hot vector, fit.py:536): values [F,M] and, from
        sum_f <out_f, gy_f>, the autograd gradients of the reference's own code."""
        t = {k: torch.tensor(inp[k]).requires_grad_(k in leaves) for k in ("v_base", "Bmat", "M1", "M2", "m1", "m2", "m3")}
        vals, total = [], 0.0
        for f in range(F):
            e = torch.zeros(F); e[f] = 1.0
            o = call(t, e)
            assert o.shape == (M,)
            vals.append(o.detach().numpy().astype(np.float64).tolist())
            total = total + (o * torch.tensor(inp["gy"][f])).sum()
        total.backward()
        out["cases"][name] = {"out": vals, "grads": {k: t[k].grad.numpy().astype(np.float64).tolist() for k in leaves}}

    run("blend", lambda t, e: fns["blend"](t["v_base"], {"local": t["M1"]}, {"local": t["M2"]}, {"local": t["Bmat"]}, e),
        ("v_base", "Bmat", "M1", "M2"))
    # the 'global' in dataset branch (fit.py:122-128): no intermediate map, maps['local'] is [K,F]
    run("blend_global_key", lambda t, e: fns["blend"](t["v_base"], {"local": t["M2"]}, {}, {"local": t["Bmat"], "global": None}, e),
        ("v_base", "Bmat", "M2"))
    run("blend_free", lambda t, e: fns["blend_free"](t["v_base"], t["m1"], t["m2"], t["m3"], e), ("v_base", "m1", "m2", "m3"))
    run("blend_combined", lambda t, e: fns["blend_combined"](t["v_base"], t["m1"], t["m2"], t["m3"], {"local": t["M1"]}, {"local": t["M2"]},
                                                          {"local": t["Bmat"]}, e, learned_coefficient=inp["learned_coefficient"]),
        ("v_base", "Bmat", "M1", "M2", "m1", "m2", "m3"))
    run("blend_combined_default_coefficient", lambda t, e: fns["blend_combined"](t["v_base"], t["m1"], t["m2"], t["m3"], {"local": t["M1"]},
                                                                              {"local": t["M2"]}, {"local": t["Bmat"]}, e),
        ("M1", "M2", "m1", "m2", "m3"))
    with open(os.path.join(HERE, "blend_golden.json"), "w") as f:
        json.dump(out, f)


def numframes_golden():
    fns, lines = reference_functions(("assertNumFrames",), {"os": os})
    cases = []
    for counts in ([3, 3, 3], [120, 120], [99], [100], [4, 5]):
        with tempfile.TemporaryDirectory() as d:
            cams = []
            for i, n in enumerate(counts):
                cams.append(f"cam{i}")
                os.makedirs(os.path.join(d, cams[-1]))
                for j in range(n):
                    open(os.path.join(d, cams[-1], f"{j:04d}.tif"), "w").close()
            try:
                res = list(fns["assertNumFrames"](cams, d))
            except AssertionError as e:
                res = {"AssertionError": str(e)}
        cases.append({"counts": counts, "result": res})
    with open(os.path.join(HERE, "numframes_golden.json"), "w") as f:
        json.dump({"reference_lines": lines, "cases": cases}, f, indent=1)


def rerender_golden():
    """f-4: the reference's grid tiler and its numerical sequence comparison, run as they are."""
    import contextlib
    import hashlib
    import io
    from pathlib import Path
    from PIL import Image
    sys.path[:0] = [os.path.dirname(HERE), os.path.dirname(os.path.dirname(HERE))]
    from helpers import comparison_pair
    out = {}
    fns, lines = reference_functions(("make_img",), {"np": np}, module="utils.py")
    out["make_img_lines"] = lines
    rng = np.random.default_rng(11)
    cases = []
    for n, h, w, c, ncols in ((6, 2, 3, 1, 3), (9, 4, 5, 1, 3), (4, 3, 2, 3, 2), (8, 2, 2, 1, 4)):
        arr = rng.integers(0, 256, size=(n, h, w, c)).astype(np.float32)
        kw = {} if ncols == 2 else {"ncols": ncols}        # (the reference's default is two columns)
        cases.append({"shape": [n, h, w, c], "ncols": ncols, "input": arr.astype(np.float64).tolist(),
                      "grid": fns["make_img"](arr, **kw).astype(np.float64).tolist()})
    try:
        fns["make_img"](np.zeros((5, 2, 2, 1)), 3)
        cases.append({"uneven": "no error"})
    except AssertionError:
        cases.append({"uneven": "AssertionError"})
    out["make_img"] = cases

    fns, lines = reference_functions(("compareSequenceNumerical",), {"np": np, "os": os, "Path": Path, "Image": Image}, module="comparisons.py")
    out["compare_lines"] = lines
    with tempfile.TemporaryDirectory() as d:
        inf, ref, save = (os.path.join(d, k) for k in ("inferred", "reference", "save"))
        os.makedirs(inf)
        os.makedirs(ref)
        for i in range(120):
            a, b = comparison_pair(i)
            Image.fromarray(a).save(os.path.join(inf, f"frame{i}_pose.png"))
            Image.fromarray(b).save(os.path.join(ref, f"pod2colour_pod2primary_{i:03d}.tif"))
        with contextlib.redirect_stdout(io.StringIO()):
            fns["compareSequenceNumerical"](inf, ref, save, colour=False)
        text = open(os.path.join(save, "numerical_clip.csv")).read()
    rows = text.split("\n")
    assert len(rows) == 121
    out["compare"] = {"images": 120, "shape": [1600, 1200], "file": "numerical_clip.csv",
                      "csv_sha256": hashlib.sha256(text.encode()).hexdigest(), "csv_bytes": len(text),
                      "image_means": [r.split(", ")[0] for r in rows[:120]],            # as the text the reference wrote
                      "values_per_line": len(rows[0].split(", ")),
                      "first_values": {str(k): rows[k].split(", ")[:12] for k in (0, 57, 119)},
                      "last_line": rows[120]}
    with open(os.path.join(HERE, "rerender_golden.json"), "w") as f:
        json.dump(out, f)


def save_golden():
    """f-3: the reference's result writer, run as it is on CPU tensors."""
    import codecs
    import contextlib
    import io
    import torch
    fns, lines = reference_functions(("save",), {"os": os, "json": json, "codecs": codecs, "np": np})
    rng = np.random.default_rng(5)
    meshes = (rng.normal(size=(2, 12)) * 7.3).astype(np.float32)
    meshes[0, :4] = [0.1, -2.5, 1e-7, 123456.789]               # values whose float32 / double text forms differ
    uv = rng.uniform(size=(5, 2)).astype(np.float32)
    uv[0] = [0.0, 1.0]
    tex = rng.uniform(size=(4, 4, 1)).astype(np.float32)
    t = (rng.normal(size=(2, 3)) * 0.3).astype(np.float32)
    q = np.array([[0, 0, 0, 1], [0.01, -0.02, 0.03, 0.999]], dtype=np.float32)
    faces = ["f 1/1 2/2 3/3\n", "f 1/5 3/3 4/4\n"]
    with tempfile.TemporaryDirectory() as d:
        os.mkdir(os.path.join(d, "result"))
        with open(os.path.join(d, "result", "faces.txt"), "w") as f:          # (the reference copies the face lines from this file)
            f.writelines(faces)
        log = io.StringIO()
        with contextlib.redirect_stdout(log):
            fns["save"](torch.tensor(meshes), torch.tensor(uv), None, tex, torch.tensor(t), torch.tensor(q), d)
        files = sorted(os.listdir(os.path.join(d, "result")))
        text = {n: open(os.path.join(d, "result", n)).read() for n in files if n.endswith((".obj", ".json"))}
    assert "imageio failed" in log.getvalue() and "texture.png" not in files      # imageio is absent here: the texture is not pinned
    with open(os.path.join(HERE, "save_golden.json"), "w") as f:
        json.dump({"reference_lines": lines, "inputs": {"meshes": meshes.astype(np.float64).tolist(), "uv": uv.astype(np.float64).tolist(),
                                                        "texture": tex.astype(np.float64).tolist(), "translation": t.astype(np.float64).tolist(),
                                                        "rotation": q.astype(np.float64).tolist(), "faces": faces},
                   "files": files, "text": text,
                   "texture_png": "not written: imageio is absent in the build container, the function's own except branch reported it"}, f, indent=1)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "rerender":
        rerender_golden()
    elif len(sys.argv) > 1 and sys.argv[1] == "save":
        save_golden()
    else:
        main()
