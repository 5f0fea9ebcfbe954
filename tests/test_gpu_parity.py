"""
GPU parity tests: HIP kernels (through the C ABI, via fpc_diffrend_amd.ops) against the CPU oracle on
the same seeded inputs.  Bars (BASELINE.json north_star): integer triangle-id / coverage buffers
bit-exact; float tensors within 1e-4 relative L2.  The oracle is this build's own restatement
(parity unpinned vs nvdiffrast).
"""
import numpy as np
import pytest
import torch

from helpers import clip_positions, random_soup, rel_l2

pytestmark = pytest.mark.gpu

TOL = 1e-4


@pytest.fixture(scope="module")
def dr():
    import fpc_diffrend_amd.ops as dr
    return dr


@pytest.fixture(scope="module")
def ctx(dr):
    return dr.RasterizeGLContext(device='cuda')


def _ids(rast):
    return rast[..., 3].to(torch.int32).cpu()


@pytest.mark.parametrize("res,T,seed", [((64, 64), 40, 0), ((200, 136), 300, 1), ((97, 131), 1500, 2), ((256, 256), 5000, 3)])
def test_rasterize_soup_ids_bit_exact(dr, ctx, oracle_ops, res, T, seed):
    pos, tri = random_soup(2, T, seed)
    rast, db = dr.rasterize(ctx, pos.cuda(), tri.cuda(), res)
    ids_ref = oracle_ops.rasterize_ids(pos, tri, res)
    assert torch.equal(_ids(rast), ids_ref)
    r_ref, db_ref = oracle_ops.rasterize(pos, tri, res)
    assert rel_l2(rast[..., :3], r_ref[..., :3]) < TOL
    assert rel_l2(db, db_ref) < TOL


def test_rasterize_mesh_cfg1(dr, ctx, oracle_ops):
    from fpc_diffrend_amd import scene
    sc = scene.cfg('cfg1')
    pos, _ = clip_positions(sc, [0, 3, 8], frames=[0, 1])
    tri = torch.tensor(sc.pos_idx)
    rast, db = dr.rasterize(ctx, pos.cuda(), tri.cuda(), sc.resolution)
    assert torch.equal(_ids(rast), oracle_ops.rasterize_ids(pos, tri, sc.resolution))
    r_ref, db_ref = oracle_ops.rasterize(pos, tri, sc.resolution)
    assert rel_l2(rast, r_ref) < TOL
    assert rel_l2(db, db_ref) < TOL


def test_rasterize_backward(dr, ctx, oracle_ops):
    pos, tri = random_soup(2, 200, 5)
    res = (96, 80)
    g = torch.Generator().manual_seed(1)
    gy = torch.randn(2, res[0], res[1], 4, generator=g)
    gdb = torch.randn(2, res[0], res[1], 4, generator=g) * 0.1
    p_ref = pos.clone().requires_grad_(True)
    r, d = oracle_ops.rasterize(p_ref, tri, res)
    ((r * gy).sum() + (d * gdb).sum()).backward()
    p_gpu = pos.cuda().requires_grad_(True)
    r2, d2 = dr.rasterize(ctx, p_gpu, tri.cuda(), res)
    ((r2 * gy.cuda()).sum() + (d2 * gdb.cuda()).sum()).backward()
    assert rel_l2(p_gpu.grad, p_ref.grad) < TOL
    # without db gradient
    p_ref.grad = None
    p_gpu.grad = None
    r, d = oracle_ops.rasterize(p_ref, tri, res, grad_db=False)
    ((r * gy).sum() + (d * gdb).sum()).backward()
    r2, d2 = dr.rasterize(ctx, p_gpu, tri.cuda(), res, grad_db=False)
    ((r2 * gy.cuda()).sum() + (d2 * gdb.cuda()).sum()).backward()
    assert rel_l2(p_gpu.grad, p_ref.grad) < TOL


@pytest.mark.parametrize("A,Ba,diff", [(2, 1, None), (2, 1, 'all'), (3, 2, [2, 0]), (5, 1, 'all')])
def test_interpolate_fwd_bwd(dr, ctx, oracle_ops, A, Ba, diff):
    pos, tri = random_soup(2, 150, 7)
    res = (72, 88)
    g = torch.Generator().manual_seed(2)
    Vt = 3 * 150
    attr = torch.randn(Ba, Vt, A, generator=g)
    r_ref, db_ref = oracle_ops.rasterize(pos, tri, res)
    r_ref = r_ref.detach().requires_grad_(True)
    db_ref = db_ref.detach().requires_grad_(True)
    a_ref = attr.clone().requires_grad_(True)
    o, oda = oracle_ops.interpolate(a_ref, r_ref, tri, rast_db=db_ref if diff else None, diff_attrs=diff)
    gy = torch.randn(o.shape, generator=g)
    gda = torch.randn(oda.shape, generator=g)
    ((o * gy).sum() + (oda * gda).sum()).backward()

    r_gpu = r_ref.detach().cuda().requires_grad_(True)
    db_gpu = db_ref.detach().cuda().requires_grad_(True)
    a_gpu = attr.cuda().requires_grad_(True)
    o2, oda2 = dr.interpolate(a_gpu, r_gpu, tri.cuda(), rast_db=db_gpu if diff else None, diff_attrs=diff)
    assert o2.shape == o.shape and oda2.shape == oda.shape
    ((o2 * gy.cuda()).sum() + (oda2 * gda.cuda()).sum()).backward()
    assert rel_l2(o2, o) < TOL
    if oda.numel():
        assert rel_l2(oda2, oda) < TOL
    assert rel_l2(a_gpu.grad, a_ref.grad) < TOL
    assert rel_l2(r_gpu.grad, r_ref.grad) < TOL
    if diff:
        assert rel_l2(db_gpu.grad, db_ref.grad) < TOL


@pytest.mark.parametrize("mode,C,Bt,boundary", [('linear', 1, 1, 'wrap'), ('linear', 3, 2, 'clamp'), ('nearest', 1, 1, 'wrap'),
                                               ('linear-mipmap-linear', 1, 1, 'wrap'), ('linear-mipmap-linear', 3, 1, 'clamp'),
                                               ('linear-mipmap-nearest', 2, 1, 'wrap'), ('linear', 1, 1, 'zero'), ('nearest', 3, 2, 'zero'),
                                               ('linear-mipmap-linear', 2, 1, 'zero')])
def test_texture_fwd_bwd(dr, oracle_ops, mode, C, Bt, boundary):
    g = torch.Generator().manual_seed(3)
    B, H, W = 2, 40, 56
    tex = torch.rand(Bt, 32, 64, C, generator=g)
    uv = torch.rand(B, H, W, 2, generator=g) * 1.6 - 0.3
    uv_da = (torch.rand(B, H, W, 4, generator=g) - 0.5) * 0.2
    mip = 'mipmap' in mode
    gy = torch.randn(B, H, W, C, generator=g)
    gy[0, :5] = 0.0  # exercises the zero-gradient skip
    t_ref = tex.clone().requires_grad_(True)
    uv_ref = uv.clone().requires_grad_(True)
    da_ref = uv_da.clone().requires_grad_(True)
    kw = dict(filter_mode=mode, boundary_mode=boundary)
    if mip:
        kw['max_mip_level'] = 4
    o = oracle_ops.texture(t_ref, uv_ref, da_ref if mip else None, **kw)
    (o * gy).sum().backward()
    t_gpu = tex.cuda().requires_grad_(True)
    uv_gpu = uv.cuda().requires_grad_(True)
    da_gpu = uv_da.cuda().requires_grad_(True)
    o2 = dr.texture(t_gpu, uv_gpu, da_gpu if mip else None, **kw)
    (o2 * gy.cuda()).sum().backward()
    assert rel_l2(o2, o) < TOL
    assert rel_l2(t_gpu.grad, t_ref.grad) < TOL
    if mode != 'nearest':
        assert rel_l2(uv_gpu.grad, uv_ref.grad) < TOL
    if mode == 'linear-mipmap-linear':
        assert rel_l2(da_gpu.grad, da_ref.grad) < TOL


def _aa_inputs(sc_name='cfg1', cams=(0, 4), C=1, seed=0):
    from fpc_diffrend_amd import scene
    sc = scene.cfg(sc_name)
    pos, _ = clip_positions(sc, list(cams), frames=[0])
    tri = torch.tensor(sc.pos_idx)
    g = torch.Generator().manual_seed(seed)
    color = torch.rand(pos.shape[0], sc.resolution[0], sc.resolution[1], C, generator=g)
    return sc, pos, tri, color


@pytest.mark.parametrize("C", [1, 3, 2])
def test_antialias_fwd_bwd_mesh(dr, ctx, oracle_ops, C):
    sc, pos, tri, color = _aa_inputs(C=C)
    rast_ref, _ = oracle_ops.rasterize(pos, tri, sc.resolution)
    rast_ref = rast_ref.detach()
    g = torch.Generator().manual_seed(9)
    gy = torch.randn(color.shape, generator=g)
    c_ref = color.clone().requires_grad_(True)
    p_ref = pos.clone().requires_grad_(True)
    o, flags = oracle_ops.antialias(c_ref, rast_ref, p_ref, tri, return_flags=True)
    (o * gy).sum().backward()
    assert int((flags > 0).sum()) > 20, "test scene must contain silhouette pairs"

    c_gpu = color.cuda().requires_grad_(True)
    p_gpu = pos.cuda().requires_grad_(True)
    o2 = dr.antialias(c_gpu, rast_ref.cuda(), p_gpu, tri.cuda())
    (o2 * gy.cuda()).sum().backward()
    assert rel_l2(o2, o) < TOL
    assert rel_l2(c_gpu.grad, c_ref.grad) < TOL
    assert rel_l2(p_gpu.grad, p_ref.grad) < TOL


def test_antialias_soup(dr, ctx, oracle_ops):
    # open geometry: every edge is a boundary edge -> many silhouette pairs, incl. against background
    pos, tri = random_soup(2, 60, 11, size=0.6)
    res = (120, 104)
    rast_ref, _ = oracle_ops.rasterize(pos, tri, res)
    rast_ref = rast_ref.detach()
    g = torch.Generator().manual_seed(4)
    color = torch.rand(2, res[0], res[1], 3, generator=g)
    gy = torch.randn(color.shape, generator=g)
    c_ref = color.clone().requires_grad_(True)
    p_ref = pos.clone().requires_grad_(True)
    o = oracle_ops.antialias(c_ref, rast_ref, p_ref, tri)
    (o * gy).sum().backward()
    c_gpu = color.cuda().requires_grad_(True)
    p_gpu = pos.cuda().requires_grad_(True)
    o2 = dr.antialias(c_gpu, rast_ref.cuda(), p_gpu, tri.cuda())
    (o2 * gy.cuda()).sum().backward()
    assert float((o - color).abs().sum()) > 1.0
    assert rel_l2(o2, o) < TOL
    assert rel_l2(c_gpu.grad, c_ref.grad) < TOL
    assert rel_l2(p_gpu.grad, p_ref.grad) < TOL


def test_topology_matches_oracle(dr, oracle_ops):
    from fpc_diffrend_amd import scene
    sc = scene.cfg('cfg1')
    tri = torch.tensor(sc.pos_idx)
    # open the mesh (drop triangles) and add a non-manifold fin
    tri = torch.cat([tri[40:], torch.tensor([[1, 2, 500], [1, 2, 501]], dtype=torch.int32)])
    adj = dr.antialias_construct_topology_hash(tri.cuda()).cpu()
    cnt, oth = oracle_ops.edge_table(tri)
    expect = torch.where(cnt == 1, torch.full_like(oth, -1), torch.where(cnt == 2, oth, torch.full_like(oth, -2)))
    assert torch.equal(adj, expect)


@pytest.mark.parametrize("res,T,size,seed", [((256, 256), 20000, 0.03, 5), ((192, 320), 8000, 0.08, 6)])
def test_rasterize_small_triangle_soup_ids_bit_exact(dr, ctx, oracle_ops, res, T, size, seed):
    """Thousands of small overlapping triangles: the one-lane-per-triangle path with its LDS depth buffer (heavy overdraw,
    bins with several batches) against the oracle's plain loops."""
    pos, tri = random_soup(2, T, seed, size=size)
    rast, _ = dr.rasterize(ctx, pos.cuda(), tri.cuda(), res)
    assert torch.equal(_ids(rast), oracle_ops.rasterize_ids(pos, tri, res))


def test_rasterize_depth_ties_go_to_the_smaller_index(dr, ctx, oracle_ops):
    """Every triangle twice (identical vertices, so identical depth planes): rule R6 gives the pixel to the smaller index
    although triangles reach a bin in no particular order; plus a triangle with a vertex at w <= 0 (clipped against the near
    plane or dropped, rule R1) and one with a NaN vertex (dropped)."""
    pos, tri = random_soup(1, 600, 7, size=0.2)
    T = tri.shape[0]
    perm = torch.randperm(2 * T, generator=torch.Generator().manual_seed(0))
    tri2 = torch.cat([tri, tri], dim=0)[perm].contiguous()          # duplicates scattered over the index range
    pos = pos.clone()
    pos[0, 30, 3] = -0.5            # w <= 0
    pos[0, 91, 0] = float('nan')    # NaN x
    rast, _ = dr.rasterize(ctx, pos.cuda(), tri2.cuda(), (160, 160))
    ids = _ids(rast)
    assert torch.equal(ids, oracle_ops.rasterize_ids(pos, tri2, (160, 160)))
    # direct statement of the rule: a covered pixel's winner is the smaller of the two copies' indices
    first = torch.full((T,), 2 * T, dtype=torch.long)
    src = perm % T                                                  # original triangle of each entry of tri2
    for j in range(2 * T):
        first[src[j]] = min(first[src[j]], j)
    win = ids[ids > 0].long() - 1
    assert torch.equal(first[src[win]], win)


@pytest.mark.parametrize("C,res", [(1, (150, 200)), (3, (97, 131)), (1, (75, 101))])
def test_region_hints_do_not_change_the_operator_chain(dr, oracle_ops, C, res):
    """rasterize() tags its output with the map of bins no triangle touches; interpolate / texture / antialias and the
    backward kernels then skip the reads of those bins (include/fpcdr.h, REGION HINTS).  With and without the mechanism the
    chain of fit.py:151-160 gives bit-identical images and the same gradients -- also when a caller edits a tensor in
    between (the hint is then dropped) or feeds the operators tensors that never carried one."""
    from fpc_diffrend_amd import scene
    sc = scene.cfg('cfg1', n_frames=2)
    pos, _ = clip_positions(sc, [0, 3, 7], frames=[0, 1])
    dev = 'cuda'
    tri = torch.tensor(sc.pos_idx, device=dev)
    uv = torch.tensor(sc.uv, device=dev) * 1.1 + 0.03
    uv_idx = torch.tensor(sc.uv_idx, device=dev)
    g = torch.Generator().manual_seed(2)
    tex0 = torch.rand(48, 64, C, generator=g)
    gy = torch.randn(pos.shape[0], res[0], res[1], C, generator=g).to(dev)

    def chain(hints, edit=False):
        dr.region_hints = hints
        try:
            ctx = dr.RasterizeGLContext(device=dev)
            p = pos.to(dev).clone().requires_grad_(True)
            t = tex0.to(dev).clone().requires_grad_(True)
            if edit:
                with torch.no_grad():                   # (no graph through the edited tensor)
                    rast, _ = dr.rasterize(ctx, p, tri, res)
                    assert (dr._hint_of(rast, 'rast') is not None) == hints
                    rast[0, :40, :40, 3] = 0.0          # wipe a corner: the version counter moves, the hint must go
                assert dr._hint_of(rast, 'rast') is None
            else:
                rast, _ = dr.rasterize(ctx, p, tri, res)
            tagged = dr._hint_of(rast, 'rast') is not None
            texc, _ = dr.interpolate(uv[None], rast, uv_idx)
            col = dr.texture(t[None], texc, filter_mode='linear')
            aa = dr.antialias(col, rast, p, tri)
            (aa * gy).sum().backward()
            return tagged, [x.detach().clone() for x in (rast, texc, col, aa)], p.grad.clone(), t.grad.clone()
        finally:
            dr.region_hints = True

    on, imgs_on, gp_on, gt_on = chain(True)
    off, imgs_off, gp_off, gt_off = chain(False)
    assert on and not off
    cov = (imgs_on[0][..., 3] > 0).float().mean().item()
    assert 0.02 < cov < 0.9          # both empty and covered bins exist
    for a, b in zip(imgs_on, imgs_off):
        assert torch.equal(a, b)
    # (the corner texels collect the random gradients of every empty pixel: float32 atomic-order noise ~1e-5)
    assert rel_l2(gp_on, gp_off) < 1e-6 and rel_l2(gt_on, gt_off) < 1e-4
    # an edited rast: same result as the dense path on the same edit
    _, imgs_e, gp_e, gt_e = chain(True, edit=True)
    _, imgs_d, gp_d, gt_d = chain(False, edit=True)
    for a, b in zip(imgs_e, imgs_d):
        assert torch.equal(a, b)
    assert rel_l2(gp_e, gp_d) < 1e-6 and rel_l2(gt_e, gt_d) < 1e-4
    assert not torch.equal(imgs_e[3], imgs_on[3])
    # foreign tensors (a clone carries no hint) take the dense path and agree
    dr.region_hints = True
    ctx = dr.RasterizeGLContext(device=dev)
    rast, _ = dr.rasterize(ctx, pos.to(dev), tri, res)
    texc, _ = dr.interpolate(uv[None], rast.clone(), uv_idx)
    assert dr._hint_of(texc, 'zero') is None and torch.equal(texc, imgs_on[1])
    # ... and against the oracle
    r_ref, _ = oracle_ops.rasterize(pos, tri.cpu(), res)
    assert torch.equal(imgs_on[0][..., 3].cpu(), r_ref[..., 3])


@pytest.mark.gpu
def test_range_mode_matches_oracle(dr, oracle_ops):
    """nvdiffrast's range mode: ONE vertex array pos [V,4] and per image a slice (first, count) of the triangle list (SURVEY
    section 8b lists `ranges` among the optional arguments of the boundary).  Ids (indices into the whole `tri`) bit-exact,
    rast floats, the interpolate / antialias chain on the shared arrays, and the gradient of the shared positions (the sum
    over the images) against the oracle."""
    from fpc_diffrend_amd import scene
    sc = scene.cfg('cfg1', n_frames=1)
    pos3, _ = clip_positions(sc, [2], frames=[0])
    pos = pos3[0]                                          # [V,4]
    tri = torch.tensor(sc.pos_idx)
    T = tri.shape[0]
    ranges = torch.tensor([[0, T], [T // 3, T // 2], [5, 0], [T - 40, 40]], dtype=torch.int32)
    res = (96, 128)
    g = torch.Generator().manual_seed(5)
    attr = torch.rand(pos.shape[0], 3, generator=g)
    gy = torch.randn(ranges.shape[0], res[0], res[1], 3, generator=g)
    p_ref = pos.clone().requires_grad_(True)
    a_ref = attr.clone().requires_grad_(True)
    rast_o, _ = oracle_ops.rasterize(p_ref, tri, res, ranges=ranges)
    col_o, _ = oracle_ops.interpolate(a_ref, rast_o, tri)
    aa_o = oracle_ops.antialias(col_o, rast_o, p_ref, tri)
    (aa_o * gy).sum().backward()
    ctx = dr.RasterizeGLContext(device='cuda')
    p_gpu = pos.cuda().requires_grad_(True)
    a_gpu = attr.cuda().requires_grad_(True)
    rast, _ = dr.rasterize(ctx, p_gpu, tri.cuda(), res, ranges=ranges)
    col, _ = dr.interpolate(a_gpu, rast, tri.cuda())
    aa = dr.antialias(col, rast, p_gpu, tri.cuda())
    (aa * gy.cuda()).sum().backward()
    assert torch.equal(rast[..., 3].cpu().int(), rast_o[..., 3].int())
    ids = rast[..., 3].cpu().int()
    assert int(ids[2].max()) == 0                                          # an empty range draws nothing
    assert int(ids[1][ids[1] > 0].min()) > T // 3 and int(ids[1].max()) <= T // 3 + T // 2     # ids index the whole list
    assert int(ids[3][ids[3] > 0].min()) > T - 40 if int(ids[3].max()) > 0 else True
    assert rel_l2(rast, rast_o) < TOL and rel_l2(aa, aa_o) < TOL
    assert rel_l2(p_gpu.grad, p_ref.grad) < TOL and rel_l2(a_gpu.grad, a_ref.grad) < TOL
    with pytest.raises(ValueError):
        dr.rasterize(ctx, p_gpu, tri.cuda(), res)                           # range mode without ranges
    with pytest.raises(ValueError):
        dr.rasterize(ctx, p_gpu, tri.cuda(), res, ranges=torch.tensor([[0, T + 1]], dtype=torch.int32))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [0, 1, 2, 3, 4, 5])
def test_rasterize_fuzz_bin_population(dr, ctx, oracle_ops, seed):
    """Ids bit-exact for soups built to exercise every batch shape of the bin kernel: a few / ~100 / several hundred tiny
    triangles inside one 32 x 32-pixel bin (the lane path's 4-, 2- and 1-thread-per-triangle forms and more than one batch),
    mixed with large ones (tile path), depth ties (z quantised to 1/8), back faces and vertices behind the camera (clipped); odd
    resolutions; through the operator and through the fused forward (which carries the same rasteriser in its list form)."""
    import fpc_diffrend_amd.ops as ops
    g = torch.Generator().manual_seed(100 + seed)
    res = [(64, 96), (97, 131), (150, 200), (33, 65), (128, 128), (200, 72)][seed]
    B = 2
    n_small = [40, 110, 300, 700, 64, 129][seed]
    H, W = res
    # clusters of tiny triangles (2-6 px) around a few bin centres, in NDC
    centres = (torch.rand(B, 3, 2, generator=g) * 1.6 - 0.8)
    which = torch.randint(0, 3, (B, n_small), generator=g)
    c = torch.gather(centres, 1, which[..., None].expand(-1, -1, 2))                      # [B,n,2]
    half = torch.tensor([32.0 / W, 32.0 / H])                                             # half a bin in NDC
    c = c + (torch.rand(B, n_small, 2, generator=g) * 2 - 1) * half
    small = c[:, :, None, :] + (torch.rand(B, n_small, 3, 2, generator=g) * 2 - 1) * torch.tensor([6.0 / W, 6.0 / H])
    n_big = 12
    big = (torch.rand(B, n_big, 1, 2, generator=g) * 2 - 1) + (torch.rand(B, n_big, 3, 2, generator=g) * 2 - 1) * 0.9
    xy = torch.cat([small, big], dim=1)
    T = n_small + n_big
    z = torch.round((torch.rand(B, T, 3, 1, generator=g) * 2 - 1) * 0.9 * 8) / 8            # many exact depth ties
    z[:, ::7] = z[:, ::7, :1]                                                             # whole triangles at one depth
    w = torch.rand(B, T, 3, 1, generator=g) * 2.5 + 0.5
    w[:, 5::31, 0] = -0.3                                                                 # a vertex behind the camera: near-plane clipping (R1)
    pos = torch.cat([xy * w, z * w, w], dim=-1).reshape(B, T * 3, 4).contiguous()
    tri = torch.arange(T * 3, dtype=torch.int32).reshape(T, 3)
    ids_ref = oracle_ops.rasterize_ids(pos, tri, res)
    rast, _ = dr.rasterize(ctx, pos.cuda(), tri.cuda(), res)
    assert torch.equal(_ids(rast), ids_ref)
    # the fused forward (dense grid and sparse list forms of the same body)
    uv = torch.rand(T * 3, 2, generator=g).cuda()
    tex = torch.rand(16, 16, 1, generator=g).cuda()
    col, rast2 = ops.render_textured(ctx, pos.cuda(), tri.cuda(), uv, tri.cuda(), tex, res)
    assert torch.equal(_ids(rast2), ids_ref)
    assert int((ids_ref > 0).sum()) > 0
