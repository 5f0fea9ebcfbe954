"""SURVEY.md section 8 'next' rows on the host side: f-2 normal consistency, f-4 re-render helpers (no GPU needed)."""
import os

import numpy as np
import torch

from fpc_diffrend_amd import fit, rerender


def test_make_img_tiles_row_major():
    imgs = np.stack([np.full((2, 3, 1), i, dtype=np.float32) for i in range(6)])
    grid = rerender.make_img(imgs, ncols=3)
    assert grid.shape == (4, 9, 1)
    assert (grid[:2, :3] == 0).all() and (grid[:2, 3:6] == 1).all() and (grid[2:, 6:] == 5).all()


def test_mean_abs_diff_follows_the_reference_crop(tmp_path):
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, size=(1600, 1200), dtype=np.uint8)
    ref = rng.integers(0, 256, size=(1600, 1200), dtype=np.uint8)
    # the reference's loop, restated (comparisons.py:64-75): rows 200..1400 inclusive, columns 100..1099, int32 differences
    a, b = img.astype(np.int32), ref.astype(np.int32)
    row_means = [np.mean(abs(a[y][100:1100] - b[y][100:1100])) for y in range(1600) if 200 <= y <= 1400]
    want = float(np.mean(np.array(row_means)))
    got, rows = rerender.mean_abs_diff(img, ref)
    assert abs(got - want) < 1e-12 and len(rows) == 1201 and np.allclose(rows, row_means)
    means = rerender.compare_sequence_numerical([img, ref], [ref, ref], str(tmp_path / "c" / "numerical_clip.csv"))
    lines = open(tmp_path / "c" / "numerical_clip.csv").read().split("\n")
    assert len(lines) == 3 and abs(float(lines[0].split(",")[0]) - want) < 1e-9 and means[1] == 0.0
    assert abs(float(lines[2]) - want / 2) < 1e-9


def test_result_obj_and_pose_round_trip(tmp_path):
    v = np.array([[0.5, 1.25, -2.0], [3.0, 4.0, 5.0]], dtype=np.float32)
    with open(tmp_path / "0.obj", "w") as f:
        for p in v:
            f.write(f"v {p[0]} {p[1]} {p[2]}\n")
        f.write("vt 0.1 0.2\nf 1/1 2/1 1/1\n")
    assert np.array_equal(rerender.read_result_obj(str(tmp_path / "0.obj")), v)
    import json
    json.dump({"translation": [[1, 2, 3]], "rotation": [[0, 0, 0, 1]]}, open(tmp_path / "pose.json", "w"))
    t, q = rerender.read_pose(str(tmp_path))
    assert t.shape == (1, 3) and q.shape == (1, 4)


def test_normal_consistency_flat_and_folded():
    faces = np.array([[0, 1, 2], [0, 2, 3]])                       # two triangles sharing the edge (0, 2)
    topo = fit.MeshTopology(faces, 4, 'cpu')
    flat = torch.tensor([[[0., 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0]]])
    assert float(fit.mesh_normal_consistency(flat, topo)) < 1e-6
    folded = flat.clone()
    folded[0, 3] = torch.tensor([0., 0.5, 0.5])                     # lift one wing
    folded.requires_grad_(True)
    val = fit.mesh_normal_consistency(folded, topo)
    n0 = np.cross([1, 0, 0], [1, 1, 0])
    n1 = np.cross([1, 1, 0], [0, 0.5, 0.5])
    want = 1 - n0 @ n1 / np.linalg.norm(n0) / np.linalg.norm(n1)
    assert abs(float(val) - want) < 1e-6
    val.backward()
    assert torch.isfinite(folded.grad).all() and float(folded.grad.abs().sum()) > 0
