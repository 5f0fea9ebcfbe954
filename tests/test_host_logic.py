"""CPU tests (no GPU): host-side mirrors of the reference against the committed golden fixtures, the
synthetic scene generator, and that the C-ABI library loads and exports every symbol include/fpcdr.h declares."""
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch

from fpc_diffrend_amd import camera, scene

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def cam_gold():
    with open(os.path.join(GOLD, "camera_golden.json")) as f:
        return json.load(f)


def test_camera_matrices_match_reference_golden(cam_gold):
    # golden: outputs of the reference's camera.intrinsic_to_projection / extrinsic_to_modelview
    # (reference camera.py:27-66) on its own calibration.json, captured by tests/golden/make_golden.py
    assert len(cam_gold["cameras"]) == 9
    for name, c in cam_gold["cameras"].items():
        intr = np.asarray(c["intrinsic"], dtype=np.float32)
        rot = np.asarray(c["rotation"], dtype=np.float32)
        trans = np.asarray(c["translation"], dtype=np.float32)
        P = camera.intrinsic_to_projection(intr)
        MV = camera.extrinsic_to_modelview(rot, trans)
        assert P.dtype == np.float32 and str(MV.dtype) == c["MV_dtype"]
        np.testing.assert_array_equal(P, np.asarray(c["P"], dtype=np.float32))
        np.testing.assert_array_equal(MV, np.asarray(c["MV"], dtype=np.float32))
    # known answers recorded in SURVEY.md section 8c for pod2primary
    c = cam_gold["cameras"]["pod2primary"]
    P = camera.intrinsic_to_projection(np.asarray(c["intrinsic"], dtype=np.float32))
    assert abs(P[0, 0] - 12.08315) < 1e-4 and abs(P[1, 1] - 8.954283) < 1e-5
    assert abs(P[2, 2] + 1.0001) < 1e-6 and abs(P[2, 3] + 0.020001) < 1e-7


def test_small_camera_helpers_match_golden(cam_gold):
    np.testing.assert_array_equal(camera.translate(0.0, 170.0, 0.0), np.asarray(cam_gold["translate_0_170_0"], dtype=np.float32))
    np.testing.assert_array_equal(camera.default_projection(), np.asarray(cam_gold["default_projection"], dtype=np.float32))
    np.testing.assert_array_equal(camera.default_modelview(), np.asarray(cam_gold["default_modelview"], dtype=np.float32))
    np.testing.assert_allclose(camera.rotate_x(0.3), np.asarray(cam_gold["rotate_x_0p3"], dtype=np.float32), atol=1e-7)
    np.testing.assert_allclose(camera.rotate_y(0.3), np.asarray(cam_gold["rotate_y_0p3"], dtype=np.float32), atol=1e-7)


def test_rig_file_equals_golden_inputs(cam_gold):
    rig = camera.load_rig()
    assert [r['cam'] for r in rig] == list(cam_gold["cameras"].keys())
    for r in rig:
        c = cam_gold["cameras"][r['cam']]
        np.testing.assert_array_equal(r['rot'], np.asarray(c["rotation"], dtype=np.float32))
        np.testing.assert_array_equal(r['trans_calib'], np.asarray(c["translation"], dtype=np.float32))


def test_transform_clip_and_rigid():
    g = torch.Generator().manual_seed(0)
    pos = torch.randn(7, 3, generator=g)
    mvp = torch.randn(4, 4, generator=g)
    out = camera.transform_clip(mvp, pos)
    assert out.shape == (1, 7, 4)
    ref = torch.cat([pos, torch.ones(7, 1)], 1) @ mvp.t()       # reference camera.py:21-22
    assert torch.allclose(out[0], ref)
    assert torch.allclose(camera.transform_clip(mvp.numpy(), pos), out)
    # batched: 2 vertex buffers x 3 views each
    posb = torch.randn(2, 7, 3, generator=g)
    mv = torch.randn(6, 4, 4, generator=g)
    ob = camera.transform_clip(mv, posb)
    assert ob.shape == (6, 7, 4)
    for b in range(6):
        assert torch.allclose(ob[b], torch.cat([posb[b // 3], torch.ones(7, 1)], 1) @ mv[b].t(), atol=1e-6)
    # rigid_grad: reference camera.py:128-132
    R = camera.unitquat_to_rotmat(torch.tensor([0.0, 0.0, 0.0, 1.0]))
    assert torch.equal(R, torch.eye(3))
    Rt = camera.rigid_grad(torch.tensor([1.0, 2.0, 3.0]), R)
    assert torch.equal(Rt, torch.tensor(camera.translate(1, 2, 3)))
    # quaternion: 90 degrees about z (XYZW)
    s = float(np.sqrt(0.5))
    Rz = camera.unitquat_to_rotmat(torch.tensor([0.0, 0.0, s, s]))
    assert torch.allclose(Rz, torch.tensor([[0.0, -1, 0], [1, 0, 0], [0, 0, 1]]), atol=1e-6)
    # gradients flow
    t = torch.zeros(3, requires_grad=True)
    q = torch.tensor([0.0, 0.0, 0.0, 1.0], requires_grad=True)
    camera.rigid_grad(t, camera.unitquat_to_rotmat(q)).sum().backward()
    assert t.grad is not None and q.grad is not None


def test_scene_generator_matches_config_sizes():
    sc = scene.cfg('cfg1')
    assert sc.pos_idx.shape == (1024, 3) and sc.n_vertices == 514 and sc.blendshapes.shape == (1542, 10)
    assert sc.uv.shape[0] > sc.n_vertices and sc.uv_idx.shape == sc.pos_idx.shape
    assert sc.uv_idx.max() < sc.uv.shape[0] and sc.pos_idx.max() < sc.n_vertices
    assert sc.texture.shape == (256, 256, 1) and 0 <= sc.texture.min() and sc.texture.max() <= 1
    assert len(sc.cams) == 9
    # deterministic
    sc2 = scene.cfg('cfg1')
    np.testing.assert_array_equal(sc.blendshapes, sc2.blendshapes)
    np.testing.assert_array_equal(sc.texture, sc2.texture)
    # closed manifold: every edge shared by exactly two triangles
    from oracle import ops as O
    O.build()
    cnt, _ = O.edge_table(torch.tensor(sc.pos_idx))
    assert (cnt == 2).all()
    v, t, uv, tuv = scene.make_mesh(*scene.MESH_30K)
    assert t.shape[0] == 30000 and v.shape[0] == 15002
    # the head projects inside every camera's frame
    from helpers import clip_positions
    pos, _ = clip_positions(sc, list(range(9)))
    ndc = pos[..., :2] / pos[..., 3:]
    assert (pos[..., 3] > 0).all() and ndc.abs().max() < 1.0


def test_abi_library_exports_every_declared_symbol():
    from fpc_diffrend_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    header = open(os.path.join(ROOT, "include", "fpcdr.h")).read()
    header_tc = open(os.path.join(ROOT, "include", "fpcdr_twocall.h")).read()

    def declared_in(text):
        # (declarations: a return type at the start of a line; comments mention other entry points in call form)
        return set(re.findall(r"^(?:int|size_t|const char \*)\s*(fpcdr_[a-z0-9_]+)\s*\(", text, flags=re.M))

    declared, declared_tc = declared_in(header), declared_in(header_tc)
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    assert declared_tc == set(_lib.SYMBOLS_TWOCALL), declared_tc ^ set(_lib.SYMBOLS_TWOCALL)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"{name} not exported"
    # the superseded two-call form is NOT part of the product library: it lives in a second, complete library
    for name in declared_tc:
        assert not hasattr(lib, name), f"{name} (include/fpcdr_twocall.h) must not be exported by libfpcdr.so"
    lib_tc = ctypes.CDLL(_lib.TWOCALL_LIB_PATH)
    for name in declared | declared_tc:
        assert hasattr(lib_tc, name), f"{name} not exported by libfpcdr_twocall.so"
    assert _lib.load().fpcdr_abi_version() == _lib.ABI_VERSION
    header = header + header_tc
    # every parameter struct of the header has the size (and so the trailing-field layout) of its ctypes mirror: ask the C
    # compiler
    import subprocess
    import tempfile
    pairs = {"fpcdr_rasterize_fwd_params": _lib.RasterizeFwd, "fpcdr_rasterize_bwd_params": _lib.RasterizeBwd,
             "fpcdr_render_fwd_params": _lib.RenderFwd, "fpcdr_render_bwd_params": _lib.RenderBwd,
             "fpcdr_aa_loss_fwd_params": _lib.AaLossFwd, "fpcdr_render_aa_bwd_params": _lib.RenderAaBwd,
             "fpcdr_interpolate_fwd_params": _lib.InterpolateFwd, "fpcdr_interpolate_bwd_params": _lib.InterpolateBwd,
             "fpcdr_texture_fwd_params": _lib.TextureFwd, "fpcdr_texture_bwd_params": _lib.TextureBwd,
             "fpcdr_antialias_fwd_params": _lib.AntialiasFwd, "fpcdr_antialias_bwd_params": _lib.AntialiasBwd,
             "fpcdr_pixel_loss_params": _lib.PixelLoss, "fpcdr_adam_params": _lib.AdamParams,
             "fpcdr_objective_params": _lib.Objective}
    structs = set(re.findall(r"\}\s*(fpcdr_[a-z0-9_]+_params)\s*;", header))
    assert structs == set(pairs), structs ^ set(pairs)
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "sz.c")
        with open(src, "w") as f:
            f.write('#include <stdio.h>\n#include <stddef.h>\n#include "fpcdr.h"\n#include "fpcdr_twocall.h"\nint main(void) {\n')
            for name in sorted(pairs):
                f.write(f'    printf("{name} %zu\\n", sizeof({name}));\n')
                for field, _ in pairs[name]._fields_:      # every field of the binding exists in the header, at the same offset
                    f.write(f'    printf("{name}.{field} %zu\\n", offsetof({name}, {field}));\n')
            f.write("    return 0;\n}\n")
        exe = os.path.join(td, "sz")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), src, "-o", exe])
        sizes = dict(l.split() for l in subprocess.check_output([exe]).decode().splitlines())
    for name, cls in pairs.items():
        assert ctypes.sizeof(cls) == int(sizes[name]), (name, ctypes.sizeof(cls), sizes[name])
        for field, _ in cls._fields_:
            assert getattr(cls, field).offset == int(sizes[f"{name}.{field}"]), (name, field)


def test_ops_reject_cpu_tensors_and_missing_gpu():
    import fpc_diffrend_amd.ops as dr
    with pytest.raises((ValueError, RuntimeError)):
        dr.interpolate(torch.zeros(1, 3, 2), torch.zeros(1, 4, 4, 4), torch.zeros(1, 3, dtype=torch.int32))
    with pytest.raises(ValueError):
        dr.texture(torch.zeros(1, 4, 4, 1), torch.zeros(1, 4, 4, 2), filter_mode='bogus')


def test_meshdata_matches_reference_golden(tmp_path):
    # golden: what the reference's data.MeshData (data.py:7-39) parses from the fixture's OBJ text
    from fpc_diffrend_amd import data
    with open(os.path.join(GOLD, "meshdata_golden.json")) as f:
        g = json.load(f)
    p = tmp_path / "tiny.obj"
    p.write_text(g["obj_text"])
    md = data.MeshData(str(p))
    for name in ("vertices", "uv", "faces", "fuv"):
        got = getattr(md, name)
        assert str(got.dtype) == g[name + "_dtype"]
        np.testing.assert_array_equal(got, np.asarray(g[name], dtype=got.dtype))
    # blendshape deltas (reference fit.py:199-220) and frame count helper (fit.py:29-43)
    d = tmp_path / "bs"
    d.mkdir()
    (d / "a.obj").write_text("v 1 1 1\nv 2 0 0.5\nv 1 1 0\nv 0 1 -0.25\n")
    B = data.load_blendshape_deltas(str(d), md.vertices)
    assert B.shape == (12, 1) and B.dtype == np.float32
    np.testing.assert_allclose(B[:3, 0], [1, 1, 1])
    cams = tmp_path / "take"
    for c in ("x_pod1primary", "x_pod2primary"):
        (cams / c).mkdir(parents=True)
        for i in range(3):
            (cams / c / f"{c}_{i:02d}.tif").write_text("")
    assert data.assert_num_frames(["x_pod1primary", "x_pod2primary"], str(cams)) == (3, 2)
    rig = os.path.join(ROOT, "fpc_diffrend_amd", "rig9.json")
    with open(rig) as f:
        cal = {k: dict(v, distortion=[[0.0]] * 5) for k, v in json.load(f)["cameras"].items()}
    cp = tmp_path / "calibration.json"
    cp.write_text(json.dumps(cal))
    look = data.load_calibration(str(cp), ["x_pod1primary", "x_pod2primary"])
    assert look[1]['cam'] == "x_pod2primary" and look[1]['intr'].shape == (3, 3) and look[1]['trans_calib'].shape == (3, 1)


def test_assert_num_frames_matches_reference_golden(tmp_path):
    """data.assert_num_frames against the reference's own assertNumFrames (fit.py:29-43) as run by tests/golden/make_golden.py."""
    from fpc_diffrend_amd import data
    with open(os.path.join(GOLD, "numframes_golden.json")) as f:
        g = json.load(f)
    assert len(g["cases"]) >= 5
    for k, case in enumerate(g["cases"]):
        root = tmp_path / f"take{k}"
        cams = []
        for i, n in enumerate(case["counts"]):
            cams.append(f"cam{i}")
            (root / cams[-1]).mkdir(parents=True)
            for j in range(n):
                (root / cams[-1] / f"{j:04d}.tif").write_text("")
        if isinstance(case["result"], dict):
            with pytest.raises(AssertionError) as e:
                data.assert_num_frames(cams, str(root))
            assert str(e.value).startswith(case["result"]["AssertionError"])
        else:
            assert list(data.assert_num_frames(cams, str(root))) == case["result"]


def test_fitter_host_rules_graph_choice_and_frame_selection():
    """Host-side rules of the fit loop that need no GPU: FitConfig.hip_graph='auto' (graphs for few images per step, eager launches
    with launch hints for large batches) and the frame selection helper (a slice over every frame is the tensor itself, a slice of
    some frames a view, an index tensor goes through index_select: same values either way)."""
    from fpc_diffrend_amd import fit
    assert fit.Fitter.auto_graph(1, (1600, 1200))              # the reference's one-image step
    assert fit.Fitter.auto_graph(9, (1080, 1920))              # cfg2: nine views of one frame
    assert not fit.Fitter.auto_graph(288, (1080, 1920))        # cfg3: the headline batch
    assert fit.Fitter.auto_graph(288, (64, 64)) and not fit.Fitter.auto_graph(64, (2160, 3840))
    t = torch.arange(24.0).reshape(6, 4)
    assert fit.Fitter._take(t, 0, slice(0, 6)) is t and fit.Fitter._take(t, 1, slice(0, 4)) is t
    assert torch.equal(fit.Fitter._take(t, 0, slice(2, 5)), t[2:5]) and torch.equal(fit.Fitter._take(t, 1, slice(1, 3)), t[:, 1:3])
    ids = torch.tensor([4, 1])
    cols = torch.tensor([3, 0])
    assert torch.equal(fit.Fitter._take(t, 0, ids), t[ids]) and torch.equal(fit.Fitter._take(t, 1, cols), t[:, cols])
    p = t.clone().requires_grad_(True)
    fit.Fitter._take(p, 0, ids).sum().backward()
    assert torch.equal(p.grad, torch.zeros(6, 4).index_fill_(0, ids, 1.0))


def test_mapped_hints_and_zero_pool_bookkeeping():
    """Host side of two round-4 mechanisms, without a GPU.  ops._MappedHints: the objective's last kernel writes its counters into host
    memory as a sequence lock (the call's number in front of and behind them); poll() adopts counters only when both numbers agree and
    are NEW, adds the launch margin, and keeps the old caps otherwise; the overflow count is cumulative, so none is lost between polls.  fit.ZeroPool: views of one flat buffer, 16-byte aligned, plain zeros when the
    pool has no buffer or is exhausted; two pools are independent objects."""
    import fpc_diffrend_amd.ops as dr
    from fpc_diffrend_amd import fit
    h = dr._MappedHints()
    assert h.poll() == (0, 0, 0)

    def device_writes(seq, counters, overflow=False, slots_valid=1, finish=True):
        # the order of k_objective_finish (include/fpcdr.h counts_out): [6] = seq, the counters, [5] += overflow, [7], then [4] = seq
        h.host[6] = seq
        h.host[:4] = torch.tensor(counters, dtype=torch.int32)
        if overflow:
            h.host[5] += 1
        h.host[7] = slots_valid
        if finish:
            h.host[4] = seq

    s1 = h.next_seq()
    device_writes(s1, [40, 500, 8000, 7000], finish=False)                 # [deferred bins, slot demand, live bins, occupied bins]
    assert h.poll() == (0, 0, 0)                                            # (a writer in progress: [4] != [6])
    h.host[4] = s1
    assert h.poll() == (8000 + 1000, 7000 + 875, 40 + 256)                  # max(256, n / 8) of margin
    assert h.sil_bins == 500 and h.slots == 500 + max(dr.RECORD_SLOT_MARGIN, 250)
    s2 = h.next_seq()
    device_writes(s2, [1, 0, 2, 3], slots_valid=0, finish=False)            # a later call has begun to write: nothing is adopted, nothing torn
    assert h.poll() == (9000, 7875, 296)
    h.host[4] = s2
    assert h.poll() == (2 + 256, 3 + 256, 1 + 256)
    assert h.sil_bins == 500                                                # (a dense call -- [7] = 0 -- says nothing about the slot demand)
    # an overflow is cumulative on the device: call s3 overflows, call s4 lands before the host polls -- still seen
    assert h.overflowed is None
    s3, s4 = h.next_seq(), h.next_seq()
    device_writes(s3, [5, 900, 10, 9], overflow=True)
    device_writes(s4, [5, 600, 10, 9])
    h.poll()
    assert h.overflowed == s4 and h.overflow_count == 1 and h.sil_bins == 600
    h.overflowed = None
    h.poll()
    assert h.overflowed is None                                             # (reported once)
    h.frozen, h.caps = True, (3, 2, 1)
    device_writes(h.next_seq(), [7, 7, 7, 7], overflow=True)
    assert h.poll() == (3, 2, 1) and h.overflowed is not None               # frozen caps, but an overflow is never missed
    assert 0 < h.next_seq() < 0x7ffffff1

    dev = torch.device("cpu")
    z = fit.ZeroPool().zeros((3, 5), dev)
    assert z.shape == (3, 5) and float(z.abs().sum()) == 0.0
    assert fit._pool_zeros(None, (2, 2), dev).shape == (2, 2)
    buf = torch.zeros(64, dtype=torch.float32)
    pool, other = fit.ZeroPool(buf), fit.ZeroPool(torch.zeros(8))
    a = pool.zeros((3, 5), dev)
    b = pool.zeros((7,), dev)
    assert a.data_ptr() == buf.data_ptr() and b.data_ptr() == buf.data_ptr() + 16 * 4      # 15 floats -> next multiple of four
    c = pool.zeros((60,), dev)                                                              # does not fit any more
    assert c.data_ptr() < buf.data_ptr() or c.data_ptr() >= buf.data_ptr() + 64 * 4
    a += 1.0
    assert float(buf[:15].sum()) == 15.0 and float(buf[15:].sum()) == 0.0
    assert other.off == 0 and other.zeros((4,), dev).data_ptr() == other.buf.data_ptr()     # (no shared state between pools)


def test_fit_loop_kernels_have_no_private_segment():
    """a kernel with a private segment (scratch) is dispatched several times slower, taken or not -- k_fix<1> spent 220 us
    launching one-wave workgroups with 16 bytes of it, 47 without.  The kernels the fit loop launches (one-pass objective, its rasteriser,
    the step's small kernels) must report private_segment_fixed_size 0 in the code object built from the sources here."""
    import os, re, shutil, subprocess, tempfile
    llvm = "/opt/rocm/lib/llvm/bin"
    build = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fpc_diffrend_amd", "csrc", "_build")
    if not (os.path.isdir(build) and os.path.exists(os.path.join(llvm, "llvm-readelf"))):
        pytest.skip("no built objects / llvm tools")
    hot = {"objective.o": r"k_shade_list|k_fix_list|k_shade_mip_list|k_fix_mip_list|k_objective_finish|k_count_sil|k_shade_queue|k_shade_mip_queue|k_fix_queue|k_fix_mip_queue",
           "rasterize.o": r"k_setupILb1ELb1|k_bins_listILb0ELb0ELb0ELi0ELin1ELb0ELb1|k_list_|k_init_objective",      # (k_setup_clip keeps the clipper's)
           "clip.o": r"k_clip_|k_mvp_|k_lap_", "blend.o": r"k_blend_fwd_lds|k_blend_bwd_w|k_rig_", "adam.o": r"k_adam"}
    seen = 0
    for obj, pattern in hot.items():
        path = os.path.join(build, obj)
        if not os.path.exists(path):
            pytest.skip("objects not built")
        tmp = tempfile.mkdtemp()
        try:
            subprocess.check_call([f"{llvm}/llvm-objcopy", f"--dump-section=.hip_fatbin={tmp}/fb.bin", path], stderr=subprocess.DEVNULL)
            subprocess.check_call([f"{llvm}/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                                   f"--input={tmp}/fb.bin", f"--output={tmp}/dev.co", "--unbundle"], stderr=subprocess.DEVNULL)
            notes = subprocess.check_output([f"{llvm}/llvm-readelf", "--notes", f"{tmp}/dev.co"], text=True)
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
        for blk in re.split(r"\n\s*- \.agpr_count", notes)[1:]:
            name = re.search(r"\.name:\s*(\S+)", blk).group(1)
            if re.search(pattern, name):
                seen += 1
                scratch = int(re.search(r"\.private_segment_fixed_size:\s*(\d+)", blk).group(1))
                assert scratch == 0, (obj, name, scratch)
    assert seen >= 20


def test_bench_launcher_decides_from_the_command_line_and_fails_with_its_ranks():
    """bench.py: `--gpus N` (N > 1) without a torchrun environment makes the process a LAUNCHER of N fresh ranks (it never imports torch,
    let alone touches a GPU); with WORLD_SIZE / RANK set it is a rank.  A rank that fails takes the launcher down: here -- no GPU -- both
    children stop at "bench.py needs a GPU", the launcher ends the survivors by PID and exits non-zero without a result line."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import importlib.util
    src = open(os.path.join(root, "bench.py")).read()
    head = src[:src.index("import torch  # noqa: E402")]
    assert "import torch" not in head.replace("does not even import torch", "")      # the launcher part stands in front of the torch import
    ns = {"__name__": "bench_head", "__file__": os.path.join(root, "bench.py")}
    exec(compile(head, "bench.py", "exec"), ns)
    assert ns["requested_gpus"](["--steps", "3"]) == 1 and ns["requested_gpus"](["--gpus", "8"]) == 8 and ns["requested_gpus"](["--gpus=4"]) == 4
    assert ns["wants_self_launch"](["--gpus", "2"], env={}) and not ns["wants_self_launch"](["--gpus", "1"], env={})
    assert not ns["wants_self_launch"](["--gpus", "2"], env={"WORLD_SIZE": "2", "RANK": "0"})      # started by torch.distributed.run
    assert ns["wants_self_launch"](["--gpus", "2"], env={"RANK": "0", "WORLD_SIZE": "1"})      # a wrapper's one-process environment: still a launcher
    assert not ns["wants_self_launch"](["--gpus", "2"], env={"RANK": "1", "WORLD_SIZE": "2"})
    import torch
    if torch.cuda.is_available():
        return      # (the GPU box runs the real thing: tests/test_gpu_dist.py::test_bench_launches_its_own_ranks)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["FPCDR_BENCH_LAUNCH_TIMEOUT"] = "240"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--workload", "cfg1"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    err = r.stderr.decode(errors="replace")
    assert r.returncode != 0 and "bench.py launcher (--gpus 2)" in err and "needs a GPU" in err, err[-2000:]
    assert not r.stdout.decode().strip().startswith("{")
