"""GPU: the fit-loop pieces around the four ops -- MFMA blend, fused pixel loss, the whole smoke step vs the
oracle, and that a few Adam steps reduce the loss."""
import os

import numpy as np
import pytest
import torch

from helpers import rel_l2

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.mark.parametrize("M,K,F", [(1542, 10, 4), (4500, 150, 32), (999, 70, 5), (3 * 15002, 150, 32)])
def test_blend_mfma_fwd_bwd(M, K, F):
    from fpc_diffrend_amd import fit
    g = torch.Generator().manual_seed(0)
    vb, Bm, w = torch.randn(M, generator=g), torch.randn(M, K, generator=g), torch.randn(F, K, generator=g)
    go = torch.randn(F, M, generator=g)
    vb_r, Bm_r, w_r = (x.double().requires_grad_(True) for x in (vb, Bm, w))
    out_r = vb_r[None] + w_r @ Bm_r.t()
    (out_r * go.double()).sum().backward()
    vb_g, Bm_g, w_g = (x.cuda().requires_grad_(True) for x in (vb, Bm, w))
    out = fit.blend_batched(vb_g, Bm_g, w_g)
    (out * go.cuda()).sum().backward()
    assert rel_l2(out, out_r) < 1e-6
    assert rel_l2(w_g.grad, w_r.grad) < 1e-5
    assert rel_l2(Bm_g.grad, Bm_r.grad) < 1e-6
    assert rel_l2(vb_g.grad, vb_r.grad) < 1e-6


def test_blend_reference_forms_match_torch():
    from fpc_diffrend_amd import fit
    g = torch.Generator().manual_seed(1)
    M, K, F = 300, 12, 6
    v = torch.randn(M, generator=g).cuda()
    maps = {'local': torch.randn(F, F, generator=g).cuda()}
    mi = {'local': torch.randn(K, F, generator=g).cuda()}
    ds = {'local': torch.randn(M, K, generator=g).cuda()}
    m1, m2, m3 = torch.randn(F, F, generator=g).cuda(), torch.randn(F, F, generator=g).cuda(), torch.randn(M, F, generator=g).cuda()
    e = torch.zeros(F).cuda()
    e[2] = 1.0
    ref = v + ds['local'] @ (mi['local'] @ (maps['local'] @ e))          # reference fit.py:115-119
    assert rel_l2(fit.blend(v, maps, mi, ds, e), ref) < 1e-5
    ref_free = v + m3 @ (m2 @ (m1 @ e))                                   # reference fit.py:58-62
    assert rel_l2(fit.blend_free(v, m1, m2, m3, e), ref_free) < 1e-5
    ref_c = ref + 0.5 * (m3 @ (m2 @ (m1 @ e)))                            # reference fit.py:88-99
    assert rel_l2(fit.blend_combined(v, m1, m2, m3, maps, mi, ds, e, learned_coefficient=0.5), ref_c) < 1e-5


@pytest.mark.parametrize("batched", [False, True])
def test_blend_matches_reference_golden(batched):
    """fit.blend / blend_free / blend_combined (MFMA kernel behind them) and the algebra Fitter.vertices runs (fpcdr_rig_weights +
    fpcdr_blend) against tests/golden/blend_golden.json: outputs and autograd gradients of the reference's OWN three functions
    (src/torch/fit.py:47-129) on seeded inputs, captured by tests/golden/make_golden.py.  batched=False calls them the reference's way,
    one one-hot frame vector at a time (fit.py:536); batched=True hands all frames over as one-hot columns."""
    import json
    from fpc_diffrend_amd import fit
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "blend_golden.json")))
    inp = {k: (torch.tensor(np.asarray(v, dtype=np.float32)) if isinstance(v, list) else v) for k, v in gold["inputs"].items()}
    F, M = inp["gy"].shape
    names = ("v_base", "Bmat", "M1", "M2", "m1", "m2", "m3")
    calls = {
        "blend": lambda t, e: fit.blend(t["v_base"], {"local": t["M1"]}, {"local": t["M2"]}, {"local": t["Bmat"]}, e),
        "blend_global_key": lambda t, e: fit.blend(t["v_base"], {"local": t["M2"]}, {}, {"local": t["Bmat"], "global": None}, e),
        "blend_free": lambda t, e: fit.blend_free(t["v_base"], t["m1"], t["m2"], t["m3"], e),
        "blend_combined": lambda t, e: fit.blend_combined(t["v_base"], t["m1"], t["m2"], t["m3"], {"local": t["M1"]}, {"local": t["M2"]},
                                                          {"local": t["Bmat"]}, e, learned_coefficient=inp["learned_coefficient"]),
        "blend_combined_default_coefficient": lambda t, e: fit.blend_combined(t["v_base"], t["m1"], t["m2"], t["m3"], {"local": t["M1"]},
                                                                              {"local": t["M2"]}, {"local": t["Bmat"]}, e),
    }
    assert set(calls) == set(gold["cases"])
    for case, call in calls.items():
        leaves = gold["cases"][case]["grads"]
        t = {k: inp[k].cuda().requires_grad_(k in leaves) for k in names}
        if batched:
            out = call(t, torch.eye(F, device='cuda'))
        else:
            out = torch.stack([call(t, torch.eye(F, device='cuda')[f]) for f in range(F)])
        want = torch.tensor(np.asarray(gold["cases"][case]["out"], dtype=np.float32))
        assert out.shape == want.shape == (F, M)
        assert float((out.detach().cpu() - want).abs().max()) <= 2e-6 * float(want.abs().max()), case
        (out * inp["gy"].cuda()).sum().backward()
        for k, g in leaves.items():
            assert rel_l2(t[k].grad, torch.tensor(np.asarray(g, dtype=np.float32))) < 1e-5, (case, k)
    # what Fitter.vertices launches for the three modes: rig_weights (index tensor and slice forms) + blend_batched
    for ids in (torch.arange(F, device='cuda'), slice(0, F), torch.tensor([3, 0, 4], device='cuda'), slice(1, 4)):
        rows = torch.arange(F)[ids.cpu() if torch.is_tensor(ids) else ids]
        t = {k: inp[k].cuda().requires_grad_(True) for k in names}
        prior = fit.blend_batched(t["v_base"], t["Bmat"], fit.rig_weights(t["M2"], t["M1"], ids))
        basis_t = fit.rig_weights(t["m2"], t["m1"], ids)
        free = fit.blend_batched(t["v_base"], t["m3"], basis_t)
        comb = prior + 0.5 * fit.blend_batched(None, t["m3"], basis_t)
        for case, out in (("blend", prior), ("blend_free", free), ("blend_combined", comb)):
            want = torch.tensor(np.asarray(gold["cases"][case]["out"], dtype=np.float32))[rows]
            assert float((out.detach().cpu() - want).abs().max()) <= 2e-6 * float(want.abs().max()), (case, ids)
    # ... and its backward over all frames (ids = every frame, the last-but-one loop entry is a permutation: use arange again)
    for case, mode in (("blend", "prior"), ("blend_free", "free"), ("blend_combined", "combined")):
        leaves = gold["cases"][case]["grads"]
        t = {k: inp[k].cuda().requires_grad_(k in leaves) for k in names}
        ids = torch.arange(F, device='cuda')
        if mode != "free":
            out = fit.blend_batched(t["v_base"], t["Bmat"], fit.rig_weights(t["M2"], t["M1"], ids))
        if mode != "prior":
            basis_t = fit.rig_weights(t["m2"], t["m1"], ids)
            out = fit.blend_batched(t["v_base"], t["m3"], basis_t) if mode == "free" else out + 0.5 * fit.blend_batched(None, t["m3"], basis_t)
        (out * inp["gy"].cuda()).sum().backward()
        for k, g in leaves.items():
            assert rel_l2(t[k].grad, torch.tensor(np.asarray(g, dtype=np.float32))) < 1e-5, (case, k)


def test_pixel_loss_fused_equals_reference_chain():
    from fpc_diffrend_amd import fit
    g = torch.Generator().manual_seed(2)
    B, H, W, C = 3, 37, 53, 1
    colour = torch.rand(B, H, W, C, generator=g).cuda().requires_grad_(True)
    rast = torch.zeros(B, H, W, 4)
    rast[..., 3] = (torch.rand(B, H, W, generator=g) > 0.4).float() * 7
    rast = rast.cuda()
    ref = torch.randint(0, 141, (B, H, W), generator=g, dtype=torch.uint8).cuda()
    col = torch.where(rast[..., 3:] > 0, colour, torch.tensor(fit.BACKGROUND).cuda())     # reference fit.py:161
    loss = torch.mean((ref[..., None].float() - col * 255) ** 2)                            # reference fit.py:579
    loss.backward()
    s, grad = fit.pixel_loss_fused(colour, rast, ref)
    assert abs(float(s[0]) / colour.numel() - float(loss)) < 1e-3 * float(loss)
    assert rel_l2(grad, colour.grad) < 1e-6


def test_smoke_step_matches_oracle(oracle_ops):
    from fpc_diffrend_amd import fit, scene
    from oracle import fit as ofit
    cams = (0, 4)
    sc = scene.cfg('cfg1', n_frames=2)
    targets = fit.smoke_targets(sc, cams)
    res = fit.smoke_step(sc, device='cuda:0', cams=cams)
    pos_clip = res['pos_clip'].cpu()
    # (1) the four ops + loss on bit-identical input (the clip positions the GPU chain produced), against the float32
    # oracle: integer buffer bit-exact, image / loss / both gradients within 1e-4 relative L2 -- for the operator chain and
    # for the fused objective (the form Fitter.step runs)
    ref = ofit.smoke_from_clip(sc, pos_clip, targets, cams)
    assert torch.equal(res['ids'].cpu(), ref['ids'])
    assert rel_l2(res['image'], ref['image']) < TOL
    assert abs(float(res['loss']) - float(ref['loss'])) < 1e-4 * float(ref['loss'])
    assert abs(float(res['loss_fused']) - float(ref['loss'])) < 1e-4 * float(ref['loss'])
    ref64 = ofit.smoke_from_clip(sc, pos_clip, targets, cams, dtype=torch.float64, ids=ref['ids'])
    for k in ('grad_pos_clip', 'grad_tex'):
        e_ops, e_fused = rel_l2(res[k], ref[k]), rel_l2(res[k + '_fused'], ref[k])
        print(f"{k}: operators vs f32 oracle {e_ops:.2e}, fused vs f32 oracle {e_fused:.2e}; (information) operators vs f64 "
              f"{rel_l2(res[k], ref64[k]):.2e}, f32 oracle vs f64 {rel_l2(ref[k], ref64[k]):.2e}")
        assert e_ops < TOL, (k, e_ops)
        assert e_fused < TOL, (k, e_fused)
        # the float64 evaluation of the same rules on the same visibility is the yardstick that is NOT this build's float32 arithmetic:
        # the HIP path is no further from it than the float32 oracle itself is (profiles/r04_float_rules.txt: 1.6e-3 for positions --
        # the depth test and the texel cells are step functions of rounded inputs --, 7e-6 for the texture), asserted, not printed
        floor = rel_l2(ref[k], ref64[k])
        assert (1e-4 if k == 'grad_pos_clip' else 1e-7) < floor < (1e-2 if k == 'grad_pos_clip' else 1e-4), (k, floor)
        for name in (k, k + '_fused'):
            d64 = rel_l2(res[name], ref64[k])
            assert abs(d64 - floor) <= 0.02 * floor + 1e-7, (name, d64, floor)
    # (2) upstream of the ops (transform_clip, MVP chain, MFMA blend): chain the GPU's d loss / d pos_clip through
    # the CPU restatement in float64 and compare the parameter gradients
    up = ofit.smoke_upstream(sc, res['grad_pos_clip'].cpu(), cams)
    for k in ('grad_w', 'grad_pose'):
        assert rel_l2(res[k], up[k]) < TOL, (k, rel_l2(res[k], up[k]))
    # (3) the end-to-end oracle (its own CPU matmul for the positions) agrees up to depth near-ties at folds
    e2e = ofit.smoke_step(sc, targets, cams)
    assert int((res['ids'].cpu() != e2e['ids']).sum()) <= 1e-4 * e2e['ids'].numel()
    assert abs(float(res['loss']) - float(e2e['loss'])) < 1e-3 * float(e2e['loss'])


@pytest.mark.parametrize("mode", ["free", "combined"])
def test_free_form_modes_gradients_match_oracle(mode, oracle_ops):
    """BASELINE configs[4]'s "per-vertex free-form offsets" (reference fit.py:47-62, 66-99): the gradients of the learned
    basis m1 / m2 / m3 (and, combined, of the rig prior M1 / M2) through fpcdr_blend_bwd_basis / _bwd_w equal the float64
    restatement chained from the same d loss / d pos_clip; the raster ops are compared on the same positions as above."""
    from fpc_diffrend_amd import fit, scene
    from oracle import fit as ofit
    cams = (0, 4)
    sc = scene.cfg('cfg1', n_frames=2)
    targets = fit.smoke_targets(sc, cams)
    m3 = ofit.free_form_pattern(sc.v_base.shape[0], 2)
    res = fit.smoke_step(sc, device='cuda:0', cams=cams, mode=mode, m3_init=m3)
    ref = ofit.smoke_from_clip(sc, res['pos_clip'].cpu(), targets, cams)
    assert torch.equal(res['ids'].cpu(), ref['ids'])
    for k in ('grad_pos_clip', 'grad_tex'):
        assert rel_l2(res[k], ref[k]) < TOL, (k, rel_l2(res[k], ref[k]))
    up = ofit.smoke_upstream(sc, res['grad_pos_clip'].cpu(), cams, mode=mode)
    keys = ['grad_m1', 'grad_m2', 'grad_m3', 'grad_pose'] + (['grad_w', 'grad_M1'] if mode == 'combined' else [])
    for k in keys:
        assert res[k] is not None and float(up[k].abs().max()) > 0, k
        assert rel_l2(res[k], up[k]) < TOL, (k, rel_l2(res[k], up[k]))
    # the oracle's own end-to-end evaluation of the mode (its CPU blend) sees the same image up to fold near-ties
    e2e = ofit.smoke_step(sc, targets, cams, mode=mode)
    assert int((res['ids'].cpu() != e2e['ids']).sum()) <= 1e-4 * e2e['ids'].numel()
    assert abs(float(res['loss']) - float(e2e['loss'])) < 1e-3 * float(e2e['loss'])


@pytest.mark.parametrize("fused", [True, False])
def test_fit_reduces_loss(fused):
    from fpc_diffrend_amd import fit, scene
    sc = scene.cfg('cfg1', n_frames=4)
    sc.q_gt[:] = (0.0, 0.0, 0.0, 1.0)     # no head rotation in the targets (see Fitter.init_near_truth)
    cfg = fit.FitConfig(max_iter=30, cam_idxs=(0, 3, 6), lr_base=5e-3, lr_t=5e-3, lr_q=1e-5, fused_loss=fused,
                        weight_laplacian=0.0, optimize_texture=False)
    ft = fit.Fitter(sc, cfg, device='cuda')
    ft.init_near_truth(0.8)
    act = sc.weights_gt > 0
    w0 = np.abs(ft.weights().cpu().numpy() - sc.weights_gt)[act].mean()
    losses = [float(ft.step()) for _ in range(30)]
    assert np.isfinite(losses).all()
    assert losses[-1] < 0.5 * losses[0], losses
    # the active blendshapes' weights move towards the hidden ground truth
    assert np.abs(ft.weights().cpu().numpy() - sc.weights_gt)[act].mean() < w0


@pytest.mark.parametrize("C,boundary", [(1, 'wrap'), (3, 'clamp'), (1, 'zero'), (4, 'zero')])
def test_fused_render_equals_separate_ops(C, boundary):
    """ops.render_textured == rasterize -> interpolate -> texture('linear'): bitwise forward, gradients to tolerance."""
    import fpc_diffrend_amd.ops as dr
    from fpc_diffrend_amd import scene
    from helpers import clip_positions
    sc = scene.cfg('cfg1', n_frames=2)
    pos, _ = clip_positions(sc, [0, 3, 7], frames=[0, 1])
    dev = 'cuda'
    tri = torch.tensor(sc.pos_idx, device=dev)
    uv = torch.tensor(sc.uv, device=dev) * 1.3 - 0.1          # exercise the boundary mode
    uv_idx = torch.tensor(sc.uv_idx, device=dev)
    g = torch.Generator().manual_seed(0)
    tex0 = torch.rand(64, 48, C, generator=g)
    gy = torch.randn(pos.shape[0], sc.resolution[0], sc.resolution[1], C, generator=g).to(dev)
    ctx = dr.RasterizeGLContext(device=dev)

    p1 = pos.to(dev).requires_grad_(True)
    t1 = tex0.to(dev).requires_grad_(True)
    rast, _ = dr.rasterize(ctx, p1, tri, sc.resolution)
    texc, _ = dr.interpolate(uv[None], rast, uv_idx)
    col = dr.texture(t1[None], texc, filter_mode='linear', boundary_mode=boundary)
    (col * gy).sum().backward()

    p2 = pos.to(dev).requires_grad_(True)
    t2 = tex0.to(dev).requires_grad_(True)
    col2, rast2 = dr.render_textured(ctx, p2, tri, uv, uv_idx, t2, sc.resolution, boundary_mode=boundary)
    (col2 * gy).sum().backward()
    assert torch.equal(rast2, rast)
    assert torch.equal(col2, col)
    # every texel but the four taps of uv = (0,0): the 1e-4 bar.  Those four collect the random gradient of EVERY empty pixel -- a float32
    # sum of ~3e5 terms in the arbitrary order of the atomics, on both sides -- and are held to the conditioning of such a sum instead:
    # a difference below 1e-5 of the sum of the terms' magnitudes (r3 asserted 1e-3 on the whole tensor because of them)
    corner = torch.zeros(t1.grad.shape[:2], dtype=torch.bool, device=dev)
    corner[0, 0] = corner[0, -1] = corner[-1, 0] = corner[-1, -1] = True
    assert rel_l2(t2.grad[~corner], t1.grad[~corner]) < 1e-4
    empty = (rast[..., 3] == 0)
    terms = float(gy[empty].abs().sum())
    assert float((t2.grad[corner] - t1.grad[corner]).abs().max()) < 1e-5 * terms, (float((t2.grad[corner] - t1.grad[corner]).abs().max()), terms)
    assert rel_l2(p2.grad, p1.grad) < 1e-4


@pytest.mark.parametrize("C,mip", [(1, False), (3, False), (1, True)])
def test_fitter_fused_and_unfused_paths_agree(C, mip):
    """The three execution paths of the pixel term -- one-shot objective (3 kernels), fused render + separate
    antialias / loss, and the four nvdiffrast-style ops + reference loss chain -- give the same loss and gradients; with
    enable_mip (the reference's other render() branch) the one-shot objective runs the mip-mapped lookup itself."""
    from fpc_diffrend_amd import fit, scene
    sc = scene.cfg('cfg1', n_frames=2)
    if C != 1:
        sc.texture = np.repeat(sc.texture, C, axis=2) * np.linspace(1.0, 0.6, C, dtype=np.float32)
    results = []
    for kw in (dict(), dict(sparse_objective=False), dict(fused_objective=False),
               dict(fused_objective=False, fused_render=False, fused_loss=False)):
        cfg = fit.FitConfig(max_iter=10, cam_idxs=(0, 5), weight_laplacian=10.0, enable_mip=mip, max_mip_level=3, **kw)
        ft = fit.Fitter(sc, cfg, device='cuda')
        ft.init_near_truth(0.7)
        loss = ft.loss_and_backward(torch.arange(0, 2, device='cuda'))
        results.append((float(loss), ft.maps_intermediate['local'].grad.clone(), ft.tex_opt.grad.clone(),
                        ft.per_frame_t.grad.clone(), ft.t_opt.grad.clone()))
    ref = results[-1]
    for r in results[:-1]:
        assert abs(r[0] - ref[0]) < 1e-5 * abs(ref[0])
        for a, b in zip(r[1:], ref[1:]):
            assert rel_l2(a, b) < 1e-4


def test_laplacian_gather_form_matches_dense():
    from fpc_diffrend_amd import fit, scene
    sc = scene.cfg('cfg1', n_frames=3)
    topo = fit.MeshTopology(sc.pos_idx, sc.n_vertices, 'cuda')
    g = torch.Generator().manual_seed(0)
    verts = (torch.tensor(sc.v_base).reshape(1, -1, 3) + 0.1 * torch.randn(3, sc.n_vertices, 3, generator=g)).cuda()
    v1 = verts.clone().requires_grad_(True)
    l1 = fit.mesh_laplacian_smoothing(v1, topo)
    l1.backward()
    # dense restatement (pytorch3d mesh_laplacian_smoothing(method='uniform') semantics)
    V = sc.n_vertices
    A = torch.zeros(V, V, device='cuda')
    A[topo.edges[:, 0], topo.edges[:, 1]] = 1
    A[topo.edges[:, 1], topo.edges[:, 0]] = 1
    L = A / A.sum(1, keepdim=True).clamp(min=1) - torch.eye(V, device='cuda')
    v2 = verts.clone().requires_grad_(True)
    l2 = torch.matmul(L[None], v2).norm(dim=2).mean()
    l2.backward()
    assert abs(float(l1) - float(l2)) < 1e-6 * abs(float(l2))
    assert rel_l2(v1.grad, v2.grad) < 1e-5


def test_laplacian_penalty_one_launch_matches_the_torch_chain():
    """fit.laplacian_penalty (fpcdr_laplacian_penalty_fwd / _bwd: value by the last workgroup, transposed gather of the normalised
    Laplacian) == weight * mean_f mesh_laplacian_smoothing(mesh f)^2 through the gather kernel and torch, value and gradient; with
    an upstream factor, with vertices whose Laplacian is exactly zero (a flat regular patch: torch's norm passes 0 there), for one
    mesh and for several, and twice in a row (the call leaves its accumulators zeroed)."""
    from fpc_diffrend_amd import fit, scene
    sc = scene.cfg('cfg1', n_frames=3)
    topo = fit.MeshTopology(sc.pos_idx, sc.n_vertices, 'cuda')
    g = torch.Generator().manual_seed(5)
    for F in (1, 3):
        verts = (torch.tensor(sc.v_base).reshape(1, -1, 3) + 0.1 * torch.randn(F, sc.n_vertices, 3, generator=g)).cuda()
        for rep in range(2):
            v1 = verts.clone().requires_grad_(True)
            l1 = fit.laplacian_penalty(v1, topo, 7.5)
            (l1 * 0.3).backward()
            v2 = verts.clone().requires_grad_(True)
            l2 = 7.5 * (fit.mesh_laplacian_smoothing(v2, topo, per_mesh=True) ** 2).mean()
            (l2 * 0.3).backward()
            assert abs(float(l1) - float(l2)) < 2e-6 * abs(float(l2)), (F, rep, float(l1), float(l2))
            assert rel_l2(v1.grad, v2.grad) < 1e-5, (F, rep)
    # a flat 5 x 5 grid: the interior vertices' uniform Laplacian is exactly zero
    n = 5
    idx = lambda i, j: i * n + j
    faces = [[idx(i, j), idx(i + 1, j), idx(i, j + 1)] for i in range(n - 1) for j in range(n - 1)] + \
            [[idx(i + 1, j), idx(i + 1, j + 1), idx(i, j + 1)] for i in range(n - 1) for j in range(n - 1)]
    topo2 = fit.MeshTopology(np.asarray(faces, dtype=np.int32), n * n, 'cuda')
    # (a symmetric neighbourhood is needed for a zero: shear the grid so that the six-ring of an interior vertex is centred)
    xy = torch.tensor([[float(i) + 0.5 * j, float(j), 0.0] for i in range(n) for j in range(n)])[None].cuda()
    deg = torch.tensor([int((topo2.nbr[v] < n * n).sum()) for v in range(n * n)])
    v1 = xy.clone().requires_grad_(True)
    l1 = fit.laplacian_penalty(v1, topo2, 2.0)
    l1.backward()
    v2 = xy.clone().requires_grad_(True)
    l2 = 2.0 * (fit.mesh_laplacian_smoothing(v2, topo2, per_mesh=True) ** 2).mean()
    l2.backward()
    lap = fit._uniform_laplacian.apply(xy, topo2.nbr, topo2.nbr32, topo2.inv_deg)
    assert int((lap.norm(dim=2) == 0).sum()) >= 1 and int(deg.max()) == 6
    assert torch.isfinite(v1.grad).all()
    assert abs(float(l1) - float(l2)) < 2e-6 * abs(float(l2))
    assert rel_l2(v1.grad, v2.grad) < 1e-5


def test_transform_clip_kernel_matches_torch():
    from fpc_diffrend_amd import camera, fit
    g = torch.Generator().manual_seed(0)
    F, Nc, V = 3, 4, 1000
    mvp = torch.randn(F * Nc, 4, 4, generator=g).cuda().requires_grad_(True)
    verts = torch.randn(F, V, 3, generator=g).cuda().requires_grad_(True)
    go = torch.randn(F * Nc, V, 4, generator=g).cuda()
    ref = camera.transform_clip(mvp, verts)            # reference camera.py:19-23, batched torch.matmul
    (ref * go).sum().backward()
    gm, gv = mvp.grad.clone(), verts.grad.clone()
    mvp.grad = None; verts.grad = None
    out = fit.transform_clip_batched(mvp, verts)
    (out * go).sum().backward()
    assert rel_l2(out, ref) < 1e-6
    assert rel_l2(mvp.grad, gm) < 1e-5 and rel_l2(verts.grad, gv) < 1e-6


@pytest.mark.parametrize("mode,fps,shading", [("prior", 0, "texture"), ("combined", 2, "texture"), ("prior", 0, "vertex"), ("prior", 0, "mip")])
def test_hip_graph_steps_equal_eager_steps(mode, fps, shading):
    """FitConfig.hip_graph replays forward+backward and the Adam update as two HIP graphs: the loss trajectory is the
    eager one (same kernels, same arguments; capturable Adam keeps its step count on the device, so allow rounding)."""
    from fpc_diffrend_amd import fit, scene
    traj = {}
    for graph in (False, True):
        sc = scene.cfg('cfg1', n_frames=4)
        sc.q_gt[:] = (0.0, 0.0, 0.0, 1.0)
        mip = shading == "mip"      # the mip-mapped fused objective (its chain is rebuilt inside the captured forward)
        cfg = fit.FitConfig(max_iter=14, cam_idxs=(0, 3), lr_base=5e-3, lr_t=5e-3, lr_q=1e-5, mode=mode, frames_per_step=fps,
                            shading="texture" if mip else shading, optimize_texture=(shading != "vertex"), hip_graph=graph,
                            weight_laplacian=20.0, enable_mip=mip, max_mip_level=3)
        ft = fit.Fitter(sc, cfg, device='cuda')
        if mode == "prior":
            ft.init_near_truth(0.8)
        traj[graph] = [float(ft.step()) for _ in range(14)]
        if graph:
            assert ft._graphs is not None
    a, b = np.asarray(traj[False]), np.asarray(traj[True])
    assert np.isfinite(b).all()
    assert np.allclose(a, b, rtol=2e-3), (a, b)
    assert np.allclose(a[:5], b[:5], rtol=1e-5), (a, b)


def test_mvp_kernel_matches_torch_chain():
    """fpcdr_mvp_fwd / _bwd against the reference's chain of small torch ops (fit.py:541-553; camera.py:117-132)."""
    from fpc_diffrend_amd import camera, fit
    torch.manual_seed(3)
    dev = 'cuda'
    Fb, Nc = 5, 3
    proj = torch.randn(Nc, 4, 4, device=dev)
    t_mv = torch.randn(Nc, 4, 4, device=dev)
    leaves = [torch.randn(Nc, 4, device=dev), torch.randn(Nc, 3, device=dev), torch.randn(Fb, 4, device=dev),
              torch.randn(Fb, 3, device=dev)]          # quaternions deliberately not normalised (quirk Q3)
    g = torch.randn(Fb * Nc, 4, 4, device=dev)
    res = []
    for fused in (True, False):
        qc, tc, qf, tf = (a.clone().requires_grad_(True) for a in leaves)
        if fused:
            out = fit._mvp_func.apply(qc, tc, qf, tf, proj, t_mv)
        else:
            rigid_cam = camera.rigid_grad(tc, camera.unitquat_to_rotmat(qc))
            rigid_frame = camera.rigid_grad(tf, camera.unitquat_to_rotmat(qf))
            tr = torch.matmul(rigid_cam, t_mv)
            out = torch.matmul(proj[None], torch.matmul(rigid_frame[:, None], tr[None])).reshape(-1, 4, 4)
        out.backward(g)
        res.append([out.detach()] + [a.grad for a in (qc, tc, qf, tf)])
    for a, b in zip(*res):
        assert rel_l2(a, b) < 1e-5


def test_checkpoint_resume_continues_identically(tmp_path):
    """Fitter.save_checkpoint / load_checkpoint (the reference has none): a resumed run repeats the original's losses."""
    from fpc_diffrend_amd import fit, scene

    def make():
        sc = scene.cfg('cfg1', n_frames=4)
        sc.q_gt[:] = (0.0, 0.0, 0.0, 1.0)
        cfg = fit.FitConfig(max_iter=20, cam_idxs=(0, 3), lr_base=5e-3, lr_t=5e-3, lr_q=1e-5, frames_per_step=2,
                            weight_laplacian=20.0, weight_normalconsistency=0.1)
        ft = fit.Fitter(sc, cfg, device='cuda')
        ft.init_near_truth(0.8)
        return ft

    a = make()
    for _ in range(5):
        a.step()
    a.save_checkpoint(str(tmp_path / "ck.pt"))
    want = [float(a.step()) for _ in range(5)]
    b = make()
    b.load_checkpoint(str(tmp_path / "ck.pt"))
    got = [float(b.step()) for _ in range(5)]
    assert np.allclose(got, want, rtol=1e-5), (got, want)
    assert b.iteration == a.iteration


def test_rerender_of_saved_result_matches_the_fit_images(tmp_path):
    """f-3 + f-4: save() / save_config() write the reference's files; rerender_result() reads them back and renders the
    nine cameras into a 3 x 3 grid that equals rendering the fitter's own state."""
    from fpc_diffrend_amd import camera, fit, rerender, scene
    sc = scene.cfg('cfg1', n_frames=2)
    sc.q_gt[:] = (0.0, 0.0, 0.0, 1.0)
    cfg = fit.FitConfig(max_iter=4, lr_base=5e-3, lr_t=5e-3, lr_q=1e-5, init_texture='truth', optimize_texture=False)
    ft = fit.Fitter(sc, cfg, device='cuda')
    ft.init_near_truth(0.9)
    for _ in range(2):
        ft.step()
    ft.save(str(tmp_path))
    ft.save_config(str(tmp_path), extra={"take": "synthetic"})
    rdir = tmp_path / "result"
    assert sorted(os.listdir(rdir)) == ["0.obj", "1.obj", "pose.json", "texture.png"]
    assert "max_iter: '4'" in open(tmp_path / "config.txt").read()
    grids = dict(rerender.rerender_result(str(rdir), sc, device='cuda'))
    H, W = sc.resolution
    assert set(grids) == {0, 1} and grids[0].shape == (3 * H, 3 * W, 1) and grids[0].dtype == np.uint8
    # the same frame rendered from the fitter's tensors (texture quantised to 8 bit like the file)
    tex8 = torch.floor(ft.tex_opt.detach() * 255.0).clamp(0, 255) / 255.0      # save() truncates like the reference (fit.py:268)
    verts = ft.result[0].reshape(-1, 3)
    glctx = rerender.dr.RasterizeGLContext(device='cuda')
    imgs = rerender.render_multicam(glctx, verts, ft.pos_idx, ft.uv, ft.uv_idx, tex8, sc.cams, sc.resolution,
                                    pose=(ft.per_frame_t[0].detach(), ft.per_frame_q[0].detach()), modelview_offset=(0.0, 170.0, 0.0))
    want = np.clip(np.rint(rerender.make_img(imgs.cpu().numpy(), 3)), 0, 255).astype(np.uint8)
    diff, _ = rerender.mean_abs_diff(grids[0][..., 0], want[..., 0], rows=(0, 3 * H), cols=(0, 3 * W))
    assert diff < 0.05, diff       # OBJ text round trip of the vertices moves a few silhouette pixels at most
    assert (grids[0] > 50).mean() > 0.02   # something other than background was drawn


@pytest.mark.parametrize("C,res,boundary,geom", [(1, (150, 200), 'wrap', 'mesh'), (3, (97, 131), 'clamp', 'mesh'), (4, (64, 320), 'wrap', 'mesh'),
                                                 (1, (97, 131), 'wrap', 'few'), (3, (33, 65), 'wrap', 'few'),
                                                 (1, (97, 131), 'zero', 'mesh'), (3, (70, 96), 'zero', 'few')])
def test_objective_sparse_dense_chain_agree_at_odd_sizes(C, res, boundary, geom):
    """pixel_objective (candidate-based sparse forward, dense forward) == the operator chain + pixel loss when neither
    image side is a multiple of the 32-pixel bin or the 64-pixel flag word, for every channel count it supports.
    geom 'few': a handful of large open triangles, so that most bins hold no or one or two antialias candidates and some
    are entirely covered -- the k_aa_fix case whose barrier predicate once raced (commit 675621d)."""
    import fpc_diffrend_amd.ops as dr
    from fpc_diffrend_amd import fit, scene
    from helpers import clip_positions, random_soup
    sc = scene.cfg('cfg1', n_frames=2)
    sc.resolution = res
    dev = 'cuda'
    if geom == 'mesh':
        pos, _ = clip_positions(sc, [0, 3, 7], frames=[0, 1])
        tri = torch.tensor(sc.pos_idx, device=dev)
        uv = torch.tensor(sc.uv, device=dev) * 1.2 - 0.05
        uv_idx = torch.tensor(sc.uv_idx, device=dev)
    else:
        pos, tri = random_soup(4, 5, seed=21, spread=0.7, size=0.9)
        tri = tri.to(dev)
        g0 = torch.Generator().manual_seed(8)
        uv = (torch.rand(15, 2, generator=g0) * 1.2 - 0.1).to(dev)
        uv_idx = tri.clone()
    g = torch.Generator().manual_seed(1)
    tex0 = torch.rand(48, 64, C, generator=g) * 0.5
    ref = torch.randint(0, 141, (pos.shape[0], res[0], res[1]), generator=g, dtype=torch.uint8).to(dev)
    ctx = dr.RasterizeGLContext(device=dev)
    out = {}
    for name in ("sparse", "dense", "chain"):
        p = pos.to(dev).clone().requires_grad_(True)
        t = tex0.to(dev).clone().requires_grad_(True)
        if name == "chain":
            rast, _ = dr.rasterize(ctx, p, tri, res)
            texc, _ = dr.interpolate(uv[None], rast, uv_idx)
            col = dr.antialias(dr.texture(t[None], texc, filter_mode='linear', boundary_mode=boundary), rast, p, tri)
            img = torch.where(rast[..., 3:] > 0, col, torch.tensor(fit.BACKGROUND, device=dev))
            loss = torch.mean((ref[..., None].float() - img * 255) ** 2)
        else:
            loss = dr.pixel_objective(ctx, p, tri, uv, uv_idx, t, ref, res, boundary_mode=boundary, sparse=(name == "sparse"))
        loss.backward()
        out[name] = (float(loss), p.grad.double().cpu(), t.grad.double().cpu())
    for name in ("sparse", "dense"):
        assert abs(out[name][0] - out["chain"][0]) <= 2e-6 * abs(out["chain"][0]), (name, out[name][0], out["chain"][0])
        assert rel_l2(out[name][1], out["chain"][1]) < 1e-4, name
        assert rel_l2(out[name][2], out["chain"][2]) < 1e-4, name


def test_objective_backward_applies_the_upstream_scalar():
    """d(3.5 * objective) = 3.5 * d(objective): the upstream scalar is multiplied inside the backward kernel."""
    import fpc_diffrend_amd.ops as dr
    from fpc_diffrend_amd import scene
    from helpers import clip_positions
    sc = scene.cfg('cfg1', n_frames=1)
    pos, _ = clip_positions(sc, [0, 5], frames=[0])
    dev = 'cuda'
    tri = torch.tensor(sc.pos_idx, device=dev)
    uv, uv_idx = torch.tensor(sc.uv, device=dev), torch.tensor(sc.uv_idx, device=dev)
    ref = torch.full((pos.shape[0],) + tuple(sc.resolution), 90, dtype=torch.uint8, device=dev)
    ctx = dr.RasterizeGLContext(device=dev)
    grads = []
    for scale in (1.0, 3.5):
        p = pos.to(dev).clone().requires_grad_(True)
        t = torch.tensor(sc.texture, device=dev).clone().requires_grad_(True)
        (dr.pixel_objective(ctx, p, tri, uv, uv_idx, t, ref, sc.resolution) * scale).backward()
        grads.append((p.grad.double(), t.grad.double()))
    assert rel_l2(grads[1][0], 3.5 * grads[0][0]) < 1e-5 and rel_l2(grads[1][1], 3.5 * grads[0][1]) < 1e-5


@pytest.mark.parametrize("mode,steps,max_iter", [("prior", 4, 100), ("combined", 6, 2)])
def test_optimiser_steps_match_oracle(mode, steps, max_iter, oracle_ops):
    """ten Adam groups, lr * ramp^(i/max_iter), whole-tensor quaternion division (Q3) and -- combined mode --
    the learned basis switching on half way (reference fit.py:493-505, 603-618).

    (A) update rules: after every Fitter.step() the oracle's Trainer takes the SAME iteration from the gradients the GPU
        step left on its parameters (which must exist exactly for the tensors the reference would have trained in that
        iteration); all ten parameter tensors must agree after every step.
    (B) end to end: a second Trainer runs its own forward / backward (oracle raster ops, per-mesh squared Laplacian, edge and
        normal-consistency terms, fit.py:578-582) from the same start; its losses follow the GPU's.  (Parameters are not
        compared there: a handful of depth near-ties at the rim decide differently for positions that differ in the last
        bit -- GPU vs CPU matmul -- and the grazing-angle pixels they belong to carry ~10 % of this chain's gradient.)"""
    from fpc_diffrend_amd import fit, scene
    from oracle import fit as ofit
    cams = (0, 4)
    sc = scene.cfg('cfg1', n_frames=2)
    targets = fit.smoke_targets(sc, cams)
    hp = dict(max_iter=max_iter, lr_base=2e-3, lr_tex_coef=0.5, lr_ramp=0.005, lr_t=1e-3, lr_q=1e-4, weight_laplacian=300.0,
              weight_meshedge=0.5, weight_normalconsistency=0.2)
    st, F = ofit.perturbed_state(sc, cams, mode=mode)
    st_e2e, _ = ofit.perturbed_state(sc, cams, mode=mode)
    start = [p.detach().clone() for p in st.params()]
    ft = fit.Fitter(sc, fit.FitConfig(cam_idxs=cams, mode=mode, **hp), device='cuda', targets=targets.cuda())
    with torch.no_grad():
        for p, v in zip(ft.params, start):
            p.copy_(v.cuda())
    tr, tr_e2e = ofit.Trainer(st, **hp), ofit.Trainer(st_e2e, **hp)
    lg, lo = [], []
    for i in range(steps):
        for p in ft.params:
            p.grad = None
        lg.append(float(ft.step()))
        tr.step_with_gradients([p.grad.cpu() if p.grad is not None else None for p in ft.params])
        for name, p, q in zip(ofit.State.NAMES, ft.params, st.params()):
            assert rel_l2(p, q) < 2e-6, (i, name, rel_l2(p, q))
        assert abs(float(ft.scheduler.get_last_lr()[3]) - tr.scheduler.get_last_lr()[3]) < 1e-12
        lo.append(tr_e2e.step(torch.arange(F), targets))
    assert np.allclose(lg, lo, rtol=2e-3), (lg, lo)
    moved = [float((q.detach() - p0).abs().max()) > 0 for q, p0 in zip(st.params(), start)]
    assert moved == ([False, False, False] + [True] * 7 if mode == "prior" else [True] * 10), moved


def test_fit_from_a_take_on_disk_equals_the_in_memory_run(tmp_path):
    """base mesh OBJ + blendshape OBJ directory + calibration.json + per-camera TIFF directories (the
    reference's take layout, fit.py:415-432, 461, 514-533) -> scene.from_take -> Fitter, without any synthetic ground
    truth; ten steps from disk equal ten steps of the in-memory Scene the files were written from."""
    from fpc_diffrend_amd import fit, scene
    cams = (0, 3, 6)
    sc = scene.cfg('cfg1', n_frames=4)
    sc.q_gt[:] = (0.0, 0.0, 0.0, 1.0)
    cfg = dict(max_iter=40, lr_base=5e-3, lr_t=5e-3, lr_q=1e-5, weight_laplacian=20.0, init_texture='truth',
               log_interval=2, reg_log_interval=4)
    a = fit.Fitter(sc, fit.FitConfig(cam_idxs=cams, log_path=str(tmp_path / "a.jsonl"), **cfg), device='cuda')
    images = a.targets.cpu().numpy()                                    # [F,3,H,W] uint8, rendered from the ground truth
    from PIL import Image
    Image.fromarray((np.flip(sc.texture[..., 0], 0) * 255 + 0.5).astype(np.uint8)).save(tmp_path / "tex.png")
    base, bldir, calib, imdir = scene.write_take(sc, str(tmp_path / "take"), images, cam_idxs=cams)
    take = scene.from_take(base, bldir, calib, imdir, texpath=str(tmp_path / "tex.png"))
    assert take.weights_gt is None and take.n_frames == 4 and take.images.shape == images.shape
    assert np.array_equal(take.images, images) and np.array_equal(take.pos_idx, sc.pos_idx) and np.array_equal(take.uv_idx, sc.uv_idx)
    assert np.allclose(take.v_base, sc.v_base) and [c['cam'] for c in take.cams] == ['cam_' + sc.cams[c]['cam'] for c in cams]
    # blendshape files come back in os.listdir order (as in the reference): same set of columns
    assert np.allclose(np.sort(np.abs(take.blendshapes).sum(0)), np.sort(np.abs(sc.blendshapes).sum(0)), rtol=1e-4)
    b = fit.Fitter(take, fit.FitConfig(cam_idxs=(0, 1, 2), log_path=str(tmp_path / "b.jsonl"), **cfg), device='cuda')
    with torch.no_grad():       # same texture start (the PNG is 8 bit) and same column order of the basis
        a.tex_opt.copy_(b.tex_opt)
        b.datasets['local'].copy_(a.datasets['local'])
    la = [float(a.step()) for _ in range(10)]
    lb = [float(b.step()) for _ in range(10)]
    assert np.isfinite(lb).all() and lb[-1] < lb[0]
    assert np.allclose(la, lb, rtol=1e-4), (la, lb)
    # the step log (reference fit.py:597-601, 621-623) as JSON lines
    import json
    recs = [json.loads(l) for l in open(tmp_path / "b.jsonl")]
    assert [r["it"] for r in recs] == [0, 2, 4, 6, 8]
    assert all(len(r["lr"]) == 10 and r["frames"] == 4 for r in recs) and abs(recs[0]["loss"] - lb[0]) < 1e-6 * lb[0]
    assert "LAP" in recs[0] and "LAP" in recs[2] and "LAP" not in recs[1] and recs[2]["frames_per_s"] > 0
    b.save(str(tmp_path / "out"))
    assert sorted(os.listdir(tmp_path / "out" / "result"))[:4] == ["0.obj", "1.obj", "2.obj", "3.obj"]


def test_objective_launch_hints_do_not_change_the_result():
    """The sparse objective sizes its list kernels from the bin counts of the previous call (ops._ListHints).  Whatever the
    hint -- none, right, or far too small (the strided sweep kernels then do nearly all the work) -- loss and gradients are the
    same.  (The two-call form; the one-pass form: tests/test_gpu_objective.py.)"""
    import fpc_diffrend_amd.ops as dr
    from fpc_diffrend_amd import scene
    from helpers import clip_positions
    sc = scene.cfg('cfg1', n_frames=2)
    pos, _ = clip_positions(sc, [0, 3, 7], frames=[0, 1])
    dev = 'cuda'
    tri = torch.tensor(sc.pos_idx, device=dev)
    uv, uv_idx = torch.tensor(sc.uv, device=dev), torch.tensor(sc.uv_idx, device=dev)
    g = torch.Generator().manual_seed(3)
    ref = torch.randint(0, 141, (pos.shape[0],) + tuple(sc.resolution), generator=g, dtype=torch.uint8).to(dev)
    ctx = dr.RasterizeGLContext(device=dev)

    def run(**kw):
        p = pos.to(dev).clone().requires_grad_(True)
        t = torch.tensor(sc.texture, device=dev).clone().requires_grad_(True)
        loss = dr.pixel_objective(ctx, p, tri, uv, uv_idx, t, ref, sc.resolution, one_pass=False, **kw)
        loss.backward()
        torch.cuda.synchronize()
        return float(loss), p.grad.double().cpu(), t.grad.double().cpu()

    base = run(launch_hints=False)
    dr.clear_hints()
    first = run(queued_backward=True)   # no hint yet; leaves its counts behind
    key = next(iter(dr._list_hints))
    hints = dr._list_hints[key]
    assert hints.poll() != (0, 0, 0) and all(c >= 256 for c in hints.caps)
    second = run(queued_backward=True)  # hinted launches
    hints.event = None
    hints.caps = (3, 2, 5)              # absurdly small hints: almost every bin goes through the strided sweep
    hints.update = lambda counts: None
    third = run(queued_backward=True)
    fourth = run()                      # default: list forward, grid backward
    for r in (first, second, third, fourth):
        assert abs(r[0] - base[0]) <= 1e-6 * abs(base[0])
        assert rel_l2(r[1], base[1]) < 1e-5 and rel_l2(r[2], base[2]) < 1e-5
    dr.clear_hints()


@pytest.mark.gpu
def test_grouped_adam_equals_torch_adam_and_renormalises():
    """fpcdr_adam_step (all groups in one launch, reference fit.py:493-505, 610-618) against torch.optim.Adam on the same
    tensors: several steps, different learning rates, a tensor that starts receiving gradients later (its own step count, as
    the learned basis of the combined mode), odd sizes (scalar tail of the float4 loop), and the whole-tensor quaternion
    division (quirk Q3) -- also for a quaternion tensor that receives no gradient in a step."""
    from fpc_diffrend_amd import fit
    g = torch.Generator().manual_seed(3)
    shapes = [(150, 7), (1001,), (9, 4), (5, 4), (64, 64, 1)]
    lrs = [1e-2, 3e-3, 1e-3, 1e-4, 5e-3]
    a = [torch.randn(s, generator=g).cuda().requires_grad_(True) for s in shapes]
    b = [t.detach().clone().requires_grad_(True) for t in a]
    oa = fit.GroupedAdam([{"params": p, "lr": lr} for p, lr in zip(a, lrs)], lr=1e-3, renorm=(a[2], a[3]))
    ob = torch.optim.Adam([{"params": p, "lr": lr} for p, lr in zip(b, lrs)], lr=1e-3)
    sched_a = torch.optim.lr_scheduler.LambdaLR(oa, lr_lambda=lambda x: 0.5 ** (x / 3))
    sched_b = torch.optim.lr_scheduler.LambdaLR(ob, lr_lambda=lambda x: 0.5 ** (x / 3))
    for it in range(5):
        for k, (p, q) in enumerate(zip(a, b)):
            late = (k == 1 and it < 2) or (k == 3 and it == 1)      # no gradient in these steps
            gr = None if late else torch.randn(p.shape, generator=g).cuda() * (10.0 ** (k - 2))
            p.grad = gr
            q.grad = None if gr is None else gr.clone()
        oa.step()
        ob.step()
        with torch.no_grad():
            for q in (b[2], b[3]):
                q /= torch.sum(q ** 2) ** 0.5
        sched_a.step()
        sched_b.step()
        for k, (p, q) in enumerate(zip(a, b)):
            assert rel_l2(p, q) < 2e-6, (it, k, rel_l2(p, q))
    assert float(oa.state[a[1]]['step']) == 3 and float(oa.state[a[0]]['step']) == 5
    # the state is torch.optim.Adam's: it round-trips through state_dict into a fresh optimiser
    oc = fit.GroupedAdam([{"params": p, "lr": lr} for p, lr in zip(a, lrs)], lr=1e-3, renorm=(a[2], a[3]))
    oc.load_state_dict(oa.state_dict())
    assert torch.equal(oc.state[a[0]]['exp_avg'], oa.state[a[0]]['exp_avg']) and float(oc.state[a[1]]['step']) == 3


@pytest.mark.gpu
def test_fit_recovers_hidden_weights_and_pose():
    """SURVEY section 8c, validation (iii): the synthetic fit recovers its hidden parameters.  Targets rendered from the ground
    truth (texture scaled so that 255 x colour stays below the reference's clip at 140, fit.py:531), texture known, no
    regularisers, start at 0.9 x the true weights and translations (inside the basin of the checkered texture): the pixel
    loss falls to the 8-bit quantisation floor, the blendshape weights come back to within a few 1e-3, the translations
    move towards the truth."""
    from fpc_diffrend_amd import fit, scene
    sc = scene.cfg('cfg1', n_frames=8)
    sc.q_gt[:] = (0.0, 0.0, 0.0, 1.0)
    sc.texture = (sc.texture * 0.5).astype(np.float32)
    cfg = fit.FitConfig(max_iter=800, frames_per_step=0, init_texture="truth", optimize_texture=False, lr_base=1e-3, lr_t=1e-3, lr_q=1e-7,
                        weight_laplacian=0.0, weight_meshedge=0.0, weight_normalconsistency=0.0)
    ft = fit.Fitter(sc, cfg, device="cuda")
    ft.init_near_truth(0.9)
    act = sc.weights_gt > 0
    w_err = lambda: float(np.abs(ft.weights().cpu().numpy() - sc.weights_gt)[act].mean())
    t_err = lambda: float(np.abs(ft.per_frame_t.detach().cpu().numpy() - sc.t_gt).mean())
    w0, t0 = w_err(), t_err()
    l0 = float(ft.step())
    for _ in range(399):
        l = float(ft.step())
    assert l < 0.15 * l0, (l0, l)
    assert w_err() < 0.25 * w0, (w0, w_err())
    assert t_err() < 0.7 * t0, (t0, t_err())


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [True, False])
def test_reference_run_shape_one_random_view_per_step(fused):
    """frames_per_step = 1 with views_per_step = 1 is the reference's loop shape: ONE random (camera, frame) image per
    iteration (fit.py:525-526).  The step then equals a hand-made step on exactly that image (same camera, same frame, the
    four operators + the reference's torch loss), the camera / frame pairs vary from step to step, eager and HIP-graph
    replay give the same trajectory, and only the drawn camera's and frame's pose rows receive a gradient."""
    from fpc_diffrend_amd import fit, scene
    import fpc_diffrend_amd.ops as dr
    kw = dict(max_iter=12, cam_idxs=(0, 2, 3, 7), lr_base=5e-3, lr_t=5e-3, lr_q=1e-5, frames_per_step=1, views_per_step=1,
              weight_laplacian=0.0, init_texture='truth')
    if not fused:
        kw.update(fused_objective=False, fused_render=False, fused_loss=False)
    traj, picks = {}, []
    for graph in (False, True):
        sc = scene.cfg('cfg1', n_frames=4)
        sc.q_gt[:] = (0.0, 0.0, 0.0, 1.0)
        ft = fit.Fitter(sc, fit.FitConfig(hip_graph=graph, **kw), device='cuda')
        ft.init_near_truth(0.8)
        losses = []
        for it in range(12):
            if not graph and it == 0:
                # the same draw, by hand: peek at the generator's next picks without consuming them
                state = ft.rng.bit_generator.state
                f = int(ft.pick_frames()[0])
                v = int(ft.pick_views()[0])
                ft.rng.bit_generator.state = state
                cam = ft.cam_idxs[v]
                verts = ft.vertices(torch.tensor([f], device='cuda')).reshape(1, -1, 3)
                mvp = ft.mvp(torch.tensor([f], device='cuda'), torch.tensor([v], device='cuda'))
                col = fit.render(ft.glctx, mvp, verts, ft.pos_idx, ft.uv, ft.uv_idx, ft.tex_opt, ft.resolution, False, 0)
                want = torch.mean((ft.targets[f, v].float()[..., None] - col[0] * 255) ** 2)
                picks.append((f, cam))
            losses.append(float(ft.step()))
            if not graph and it == 0:
                assert abs(losses[0] - float(want)) < 1e-4 * float(want), (losses[0], float(want))
                rows_q = (ft.q_opt.grad.abs().sum(dim=1) > 0).nonzero().flatten().tolist()
                rows_f = (ft.per_frame_t.grad.abs().sum(dim=1) > 0).nonzero().flatten().tolist()
                assert rows_q == [cam] and rows_f == [f], (rows_q, rows_f, cam, f)
        traj[graph] = losses
        if graph:
            assert ft._graphs is not None
            # hip_graph='auto' picks graphs for a one-image step and eager launches for a batch of many large images
            assert fit.Fitter(sc, fit.FitConfig(hip_graph='auto', **kw), device='cuda').use_graph
            assert fit.Fitter.auto_graph(1, (1600, 1200)) and fit.Fitter.auto_graph(9, (1080, 1920))
            assert not fit.Fitter.auto_graph(288, (1080, 1920))
    a, b = np.asarray(traj[False]), np.asarray(traj[True])
    assert np.isfinite(b).all() and len(set(np.round(a, 3))) > 6          # different images from step to step
    assert np.allclose(a, b, rtol=2e-3), (a, b)


@pytest.mark.gpu
def test_laplacian_penalty_eager_gradient_heavy_rings_and_second_stream():
    """(i) eager_grad=True (the gradient kernel runs with the value; backward() hands the buffer over, times the upstream scalar unless
    unit_upstream) == the lazy form, on the current stream and on a second one; (ii) a fan whose hub has 200 neighbours -- rings beyond
    eight slots are finished by the whole wave, here in four rounds of 64 slots -- against the torch chain, value and gradient."""
    from fpc_diffrend_amd import fit, scene
    sc = scene.cfg('cfg1', n_frames=3)
    topo = fit.MeshTopology(sc.pos_idx, sc.n_vertices, 'cuda')
    g = torch.Generator().manual_seed(11)
    verts = (torch.tensor(sc.v_base).reshape(1, -1, 3) + 0.1 * torch.randn(3, sc.n_vertices, 3, generator=g)).cuda()
    v0 = verts.clone().requires_grad_(True)
    l0 = fit.laplacian_penalty(v0, topo, 7.5)
    (l0 * 0.3).backward()
    side = torch.cuda.Stream()
    for kw, up in ((dict(eager_grad=True), 0.3), (dict(eager_grad=True, unit_upstream=True), 1.0), (dict(eager_grad=True, stream=side), 0.3)):
        v1 = verts.clone().requires_grad_(True)
        l1 = fit.laplacian_penalty(v1, topo, 7.5, **kw)
        (l1 * up).backward() if up != 1.0 else l1.backward()
        torch.cuda.synchronize()
        assert float(l1) == float(l0), kw
        assert rel_l2(v1.grad * (0.3 / up), v0.grad) < 1e-6, kw
    # the fan: vertex 0 in the middle of a 200-gon
    n = 200
    faces = np.asarray([[0, 1 + k, 1 + (k + 1) % n] for k in range(n)], dtype=np.int32)
    topo2 = fit.MeshTopology(faces, n + 1, 'cuda')
    assert int((topo2.nbr[0] < n + 1).sum()) == n
    ang = torch.arange(n, dtype=torch.float32) * (2 * np.pi / n)
    ring = torch.stack([torch.cos(ang), torch.sin(ang), 0.05 * torch.randn(n, generator=g)], dim=1)
    xyz = torch.cat([torch.tensor([[0.1, -0.2, 0.3]]), ring])[None].repeat(2, 1, 1)
    xyz = (xyz + 0.05 * torch.randn(xyz.shape, generator=g)).cuda()
    v1 = xyz.clone().requires_grad_(True)
    l1 = fit.laplacian_penalty(v1, topo2, 3.0, eager_grad=True)
    l1.backward()
    v2 = xyz.clone().requires_grad_(True)
    l2 = 3.0 * (fit.mesh_laplacian_smoothing(v2, topo2, per_mesh=True) ** 2).mean()
    l2.backward()
    assert abs(float(l1) - float(l2)) < 2e-6 * abs(float(l2))
    assert rel_l2(v1.grad, v2.grad) < 1e-5


@pytest.mark.gpu
def test_fitter_stream_overlap_options_do_not_change_the_step():
    """FitConfig.overlap_regularisers (the Laplacian term on a second stream) and ops.OVERLAP_SIL (the silhouette bits beside the
    set-up kernel) are scheduling options: loss and gradients of a step equal the single-stream default's."""
    from fpc_diffrend_amd import fit, scene
    import fpc_diffrend_amd.ops as dr
    sc = scene.cfg('cfg1', n_frames=2)
    results = []
    try:
        for overlap in (False, True):
            dr.OVERLAP_SIL = overlap
            cfg = fit.FitConfig(max_iter=10, cam_idxs=(0, 5), weight_laplacian=10.0, weight_meshedge=1.0 if overlap else 0.0,
                                overlap_regularisers=overlap)
            ft = fit.Fitter(sc, cfg, device='cuda')
            ft.init_near_truth(0.7)
            ft.cfg.weight_meshedge = 0.0 if not overlap else 1e-30      # (a torch-chain term keeps the side-stream branch alive, at no weight)
            loss = ft.loss_and_backward(torch.arange(0, 2, device='cuda'))
            torch.cuda.synchronize()
            results.append((float(loss), ft.maps_intermediate['local'].grad.clone(), ft.tex_opt.grad.clone(), ft.per_frame_t.grad.clone()))
    finally:
        dr.OVERLAP_SIL = False
    a, b = results
    assert abs(a[0] - b[0]) < 1e-5 * abs(a[0])
    for x, y in zip(a[1:], b[1:]):
        assert rel_l2(x, y) < 1e-5


@pytest.mark.gpu
def test_rig_weights_kernel_matches_the_torch_products():
    """fit.rig_weights (fpcdr_rig_weights_fwd / _bwd) == (mi @ maps[:, ids]).t() in value and in both gradients: every frame (a slice
    over all columns), a contiguous sub-range (a rank's shard), an index tensor, an index tensor that draws a frame twice."""
    from fpc_diffrend_amd import fit
    g = torch.Generator().manual_seed(3)
    K, F = 150, 32
    # (negative entries count from the end, as maps[:, ids] takes them)
    for ids in (slice(0, F), slice(8, 24), torch.tensor([5, 0, 31, 17]), torch.tensor([2, 9, 2, 2, 30]), torch.tensor([-1, 3, -32, -7])):
        dev_ids = ids if isinstance(ids, slice) else ids.cuda()
        mi = torch.randn(K, F, generator=g).cuda().requires_grad_(True)
        maps = torch.randn(F, F, generator=g).cuda().requires_grad_(True)
        w = fit.rig_weights(mi, maps, dev_ids)
        up = torch.randn(w.shape, generator=g).cuda()
        (w * up).sum().backward()
        mi2, maps2 = mi.detach().double().requires_grad_(True), maps.detach().double().requires_grad_(True)
        w2 = torch.matmul(mi2, maps2[:, dev_ids]).t()
        (w2 * up.double()).sum().backward()
        assert w.shape == w2.shape and w.is_contiguous()
        assert rel_l2(w, w2) < 1e-6, ids
        assert rel_l2(mi.grad, mi2.grad) < 1e-6 and rel_l2(maps.grad, maps2.grad) < 1e-6, ids
    # an index outside [-F, F): IndexError like the torch form; with the host check skipped the kernels neither read out of bounds
    # nor stay silent -- NaN in that row forward, no gradient from it backward
    for bad in (torch.tensor([0, F]), torch.tensor([-F - 1, 1])):
        with pytest.raises(IndexError):
            fit.rig_weights(mi.detach(), maps.detach(), bad.cuda())
        mi3, maps3 = mi.detach().clone().requires_grad_(True), maps.detach().clone().requires_grad_(True)
        w3 = fit.rig_weights(mi3, maps3, bad.cuda(), validate=False)
        okrow = 0 if int(bad[0]) == 0 else 1
        assert torch.isnan(w3[1 - okrow]).all() and torch.isfinite(w3[okrow]).all()
        assert rel_l2(w3[okrow], (mi3.detach() @ maps3.detach()[:, int(bad[okrow])])) < 1e-6
        w3[okrow].sum().backward()
        assert torch.isfinite(mi3.grad).all() and torch.isfinite(maps3.grad).all()


def test_caller_supplied_indices_are_checked_on_the_host():
    """the indexed MVP / rig-weight kernels gather from and scatter-add into the full parameter tables without bounds
    checks (the index_select calls they replaced raised).  A caller's own index tensors are checked at the public entry points."""
    from fpc_diffrend_amd import fit, scene
    sc = scene.cfg('cfg1', n_frames=4)
    ft = fit.Fitter(sc, fit.FitConfig(max_iter=10, cam_idxs=(0, 3, 6)), device='cuda')
    dev = 'cuda'
    ok = ft.loss_and_backward(torch.tensor([1, 3], device=dev), torch.tensor([0, 2], device=dev))
    assert bool(torch.isfinite(ok))
    for frames, views in ((torch.tensor([1, 4], device=dev), None), (torch.tensor([-1, 2], device=dev), None),
                          (torch.tensor([0, 1], device=dev), torch.tensor([0, 3], device=dev)), (slice(2, 5), None),
                          (torch.tensor([0.0, 1.0], device=dev), None)):
        with pytest.raises(IndexError):
            ft.loss_and_backward(frames, views)
        if torch.is_tensor(frames) and frames.dtype == torch.int64:
            with pytest.raises(IndexError):
                ft.mvp(frames, views)
    with pytest.raises(IndexError):
        ft.vertices(torch.tensor([0, 9], device=dev))
    shard = fit.Fitter(sc, fit.FitConfig(max_iter=10, cam_idxs=(0, 3)), device='cuda', rank=1, world=2, targets=ft.targets[2:, :2].contiguous())
    with pytest.raises(IndexError):
        shard.loss_and_backward(torch.tensor([1], device=dev))      # a frame of the take, but of the other rank's shard
    assert bool(torch.isfinite(shard.loss_and_backward(torch.tensor([3], device=dev))))
