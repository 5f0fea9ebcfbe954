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
    sc = scene.cfg('cfg1', n_frames=2)
    res = fit.smoke_step(sc, device='cuda:0')
    pos_clip = res['pos_clip'].cpu()
    # (1) the four ops + loss on bit-identical input (the clip positions the GPU chain produced)
    ref = ofit.smoke_from_clip(sc, pos_clip)
    assert torch.equal(res['ids'].cpu(), ref['ids'])
    assert rel_l2(res['image'], ref['image']) < TOL
    assert abs(float(res['loss']) - float(ref['loss'])) < 1e-4 * float(ref['loss'])
    # Gradients of this chain are ill-conditioned in float32 (clip coordinates ~170 with sub-pixel differences):
    # the float32 ORACLE itself sits well above 1e-4 from a float64 evaluation.  Bar: within 1e-4 of the float64
    # truth, or no worse than 2x the float32 oracle's own distance from it.  (Per-op gradients on well-conditioned
    # inputs meet 1e-4: test_gpu_parity.)
    ref64 = ofit.smoke_from_clip(sc, pos_clip, dtype=torch.float64, ids=ref['ids'])
    for k in ('grad_pos_clip', 'grad_tex'):
        floor = rel_l2(ref[k], ref64[k])
        assert rel_l2(res[k], ref64[k]) < max(TOL, 2.0 * floor), (k, rel_l2(res[k], ref64[k]), floor)
    # (2) upstream of the ops (transform_clip, MVP chain, MFMA blend): chain the GPU's d loss / d pos_clip through
    # the CPU restatement in float64 and compare the parameter gradients
    up = ofit.smoke_upstream(sc, res['grad_pos_clip'].cpu())
    for k in ('grad_w', 'grad_pose'):
        assert rel_l2(res[k], up[k]) < TOL, (k, rel_l2(res[k], up[k]))
    # (3) the end-to-end oracle (its own CPU matmul for the positions) agrees up to depth near-ties at folds
    e2e = ofit.smoke_step(sc)
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


@pytest.mark.parametrize("C,boundary", [(1, 'wrap'), (3, 'clamp')])
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
    # (the corner texel collects the random gradients of every empty pixel: float32 atomic order noise ~1e-4)
    assert rel_l2(t2.grad, t1.grad) < 1e-3
    assert rel_l2(p2.grad, p1.grad) < 1e-4


@pytest.mark.parametrize("C", [1, 3])
def test_fitter_fused_and_unfused_paths_agree(C):
    """The three execution paths of the pixel term -- one-shot objective (3 kernels), fused render + separate
    antialias / loss, and the four nvdiffrast-style ops + reference loss chain -- give the same loss and gradients."""
    from fpc_diffrend_amd import fit, scene
    sc = scene.cfg('cfg1', n_frames=2)
    if C != 1:
        sc.texture = np.repeat(sc.texture, C, axis=2) * np.linspace(1.0, 0.6, C, dtype=np.float32)
    results = []
    for kw in (dict(), dict(sparse_objective=False), dict(fused_objective=False),
               dict(fused_objective=False, fused_render=False, fused_loss=False)):
        cfg = fit.FitConfig(max_iter=10, cam_idxs=(0, 5), weight_laplacian=10.0, **kw)
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


@pytest.mark.parametrize("mode,fps,shading", [("prior", 0, "texture"), ("combined", 2, "texture"), ("prior", 0, "vertex")])
def test_hip_graph_steps_equal_eager_steps(mode, fps, shading):
    """FitConfig.hip_graph replays forward+backward and the Adam update as two HIP graphs: the loss trajectory is the
    eager one (same kernels, same arguments; capturable Adam keeps its step count on the device, so allow rounding)."""
    from fpc_diffrend_amd import fit, scene
    traj = {}
    for graph in (False, True):
        sc = scene.cfg('cfg1', n_frames=4)
        sc.q_gt[:] = (0.0, 0.0, 0.0, 1.0)
        cfg = fit.FitConfig(max_iter=14, cam_idxs=(0, 3), lr_base=5e-3, lr_t=5e-3, lr_q=1e-5, mode=mode, frames_per_step=fps,
                            shading=shading, optimize_texture=(shading == "texture"), hip_graph=graph, weight_laplacian=20.0)
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


@pytest.mark.parametrize("C,res,boundary", [(1, (150, 200), 'wrap'), (3, (97, 131), 'clamp'), (4, (64, 320), 'wrap')])
def test_objective_sparse_dense_chain_agree_at_odd_sizes(C, res, boundary):
    """pixel_objective (candidate-based sparse forward, dense forward) == the operator chain + pixel loss when neither
    image side is a multiple of the 32-pixel bin or the 64-pixel flag word, for every channel count it supports."""
    import fpc_diffrend_amd.ops as dr
    from fpc_diffrend_amd import fit, scene
    from helpers import clip_positions
    sc = scene.cfg('cfg1', n_frames=2)
    sc.resolution = res
    pos, _ = clip_positions(sc, [0, 3, 7], frames=[0, 1])
    dev = 'cuda'
    tri = torch.tensor(sc.pos_idx, device=dev)
    uv = torch.tensor(sc.uv, device=dev) * 1.2 - 0.05
    uv_idx = torch.tensor(sc.uv_idx, device=dev)
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
