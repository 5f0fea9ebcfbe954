"""GPU parity of what earlier rounds advertised but never compared with the oracle ("Next round" item 1):
the optional nvdiffrast arguments (mip_level_bias, a prebuilt / custom mip stack, pos_gradient_boost, an explicit
topology_hash), the reference's mip chain end to end (fit.py:153-155), one image at the reference's own run shape
(main.py:28-30: 1600 x 1200, 1024^2 x 1 texture), the strided sweep kernels at 1080p, a last-bin-row case of the bin-shaped
interpolate backward, and one collective through RCCL itself."""
import os
import socket
import subprocess
import sys

import pytest
import torch

from helpers import clip_positions, random_soup, rel_l2

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = 1e-4


@pytest.fixture(scope="module")
def dr():
    import fpc_diffrend_amd.ops as dr
    return dr


# ---------------------------------------------------------------------------------------------------------------------
# optional arguments of dr.texture
# ---------------------------------------------------------------------------------------------------------------------

def _tex_inputs(C=2, seed=13):
    g = torch.Generator().manual_seed(seed)
    B, H, W = 2, 36, 52
    tex = torch.rand(1, 32, 64, C, generator=g)
    uv = torch.rand(B, H, W, 2, generator=g) * 1.4 - 0.2
    uv_da = (torch.rand(B, H, W, 4, generator=g) - 0.5) * 0.15
    bias = (torch.rand(B, H, W, generator=g) - 0.4) * 3.0        # pushes levels below 0 and above the chain too
    gy = torch.randn(B, H, W, C, generator=g)
    return tex, uv, uv_da, bias, gy


@pytest.mark.parametrize("with_da,mode", [(True, 'linear-mipmap-linear'), (False, 'linear-mipmap-linear'), (True, 'linear-mipmap-nearest'),
                                          (False, 'auto')])
def test_texture_mip_level_bias_matches_oracle(dr, oracle_ops, with_da, mode):
    """mip_level_bias [B,H,W] added to the level of detail (alone: the level IS the bias), forward and the gradients to the
    texture, uv, uv_da and the bias itself."""
    tex, uv, uv_da, bias, gy = _tex_inputs()
    kw = dict(filter_mode=mode, max_mip_level=4)
    ref = [t.clone().requires_grad_(True) for t in (tex, uv, uv_da, bias)]
    o = oracle_ops.texture(ref[0], ref[1], ref[2] if with_da else None, mip_level_bias=ref[3], **kw)
    (o * gy).sum().backward()
    gpu = [t.cuda().requires_grad_(True) for t in (tex, uv, uv_da, bias)]
    o2 = dr.texture(gpu[0], gpu[1], gpu[2] if with_da else None, mip_level_bias=gpu[3], **kw)
    (o2 * gy.cuda()).sum().backward()
    assert rel_l2(o2, o) < TOL
    assert rel_l2(gpu[0].grad, ref[0].grad) < TOL and rel_l2(gpu[1].grad, ref[1].grad) < TOL
    if with_da and mode != 'linear-mipmap-nearest':
        assert rel_l2(gpu[2].grad, ref[2].grad) < TOL
    if mode != 'linear-mipmap-nearest':          # (nearest level: piecewise constant in the level)
        assert float(ref[3].grad.abs().sum()) > 0 and rel_l2(gpu[3].grad, ref[3].grad) < TOL


def test_texture_prebuilt_and_custom_mip_stacks_match_oracle(dr, oracle_ops):
    """mip=texture_construct_mip(tex): same values and the same texture gradient as the internally built chain.
    mip=[own tensors]: the levels are sampled as given and receive their own gradients, none of which reaches tex
    (nvdiffrast's documented behaviour for a custom stack); the oracle differentiates the same list with autograd."""
    tex, uv, uv_da, _, gy = _tex_inputs(C=1, seed=17)
    kw = dict(filter_mode='linear-mipmap-linear', max_mip_level=3)
    t_ref, uv_ref, da_ref = (t.clone().requires_grad_(True) for t in (tex, uv, uv_da))
    o = oracle_ops.texture(t_ref, uv_ref, da_ref, **kw)
    (o * gy).sum().backward()
    t_gpu, uv_gpu, da_gpu = (t.cuda().requires_grad_(True) for t in (tex, uv, uv_da))
    stack = dr.texture_construct_mip(t_gpu, max_mip_level=3)
    assert len(stack) == 3 and tuple(stack[2].shape) == (1, 4, 8, 1)
    for lvl, want in zip(stack, oracle_ops.build_mip_chain(tex, 3)[1:]):
        assert rel_l2(lvl, want) < 1e-6
    o2 = dr.texture(t_gpu, uv_gpu, da_gpu, mip=stack, **kw)
    (o2 * gy.cuda()).sum().backward()
    assert rel_l2(o2, o) < TOL and rel_l2(t_gpu.grad, t_ref.grad) < TOL
    assert rel_l2(uv_gpu.grad, uv_ref.grad) < TOL and rel_l2(da_gpu.grad, da_ref.grad) < TOL
    # a custom stack: unrelated level contents
    g = torch.Generator().manual_seed(5)
    own = [torch.rand(1, 32 >> l, 64 >> l, 1, generator=g) for l in (1, 2, 3)]
    t_ref.grad = None
    own_ref = [m.clone().requires_grad_(True) for m in own]
    o = oracle_ops.texture(t_ref, uv, uv_da, mip=own_ref, **kw)
    (o * gy).sum().backward()
    t_gpu.grad = None
    own_gpu = [m.cuda().requires_grad_(True) for m in own]
    o2 = dr.texture(t_gpu, uv.cuda(), uv_da.cuda(), mip=own_gpu, **kw)
    (o2 * gy.cuda()).sum().backward()
    assert rel_l2(o2, o) < TOL and rel_l2(t_gpu.grad, t_ref.grad) < TOL
    for a, b in zip(own_gpu, own_ref):
        assert float(b.grad.abs().sum()) > 0 and rel_l2(a.grad, b.grad) < TOL
    with pytest.raises(ValueError):
        dr.texture(t_gpu, uv.cuda(), uv_da.cuda(), mip=[own_gpu[1]], **kw)      # level 1 of the wrong size


# ---------------------------------------------------------------------------------------------------------------------
# optional arguments of dr.antialias
# ---------------------------------------------------------------------------------------------------------------------

def test_antialias_pos_gradient_boost_and_explicit_topology_hash(dr, oracle_ops):
    """pos_gradient_boost scales the position gradient and nothing else; topology_hash=<antialias_construct_topology_hash(tri)>
    gives the result of the cached, implicit one; a hash of another index buffer is refused."""
    pos, tri = random_soup(2, 50, 21, size=0.55)
    res = (88, 104)
    rast, _ = oracle_ops.rasterize(pos, tri, res)
    rast = rast.detach()
    g = torch.Generator().manual_seed(6)
    color = torch.rand(2, res[0], res[1], 3, generator=g)
    gy = torch.randn(color.shape, generator=g)
    out = {}
    for boost in (1.0, 3.5):
        c_ref, p_ref = color.clone().requires_grad_(True), pos.clone().requires_grad_(True)
        o = oracle_ops.antialias(c_ref, rast, p_ref, tri, pos_gradient_boost=boost)
        (o * gy).sum().backward()
        c_gpu, p_gpu = color.cuda().requires_grad_(True), pos.cuda().requires_grad_(True)
        topo = dr.antialias_construct_topology_hash(tri.cuda())
        o2 = dr.antialias(c_gpu, rast.cuda(), p_gpu, tri.cuda(), topology_hash=topo, pos_gradient_boost=boost)
        (o2 * gy.cuda()).sum().backward()
        assert rel_l2(o2, o) < TOL and rel_l2(c_gpu.grad, c_ref.grad) < TOL and rel_l2(p_gpu.grad, p_ref.grad) < TOL
        out[boost] = (o2.detach(), c_gpu.grad.clone(), p_gpu.grad.clone())
        # the implicit (cached) topology: identical
        c3, p3 = color.cuda().requires_grad_(True), pos.cuda().requires_grad_(True)
        o3 = dr.antialias(c3, rast.cuda(), p3, tri.cuda(), pos_gradient_boost=boost)
        (o3 * gy.cuda()).sum().backward()
        assert torch.equal(o3, o2) and rel_l2(p3.grad, p_gpu.grad) < 1e-6
    assert torch.equal(out[1.0][0], out[3.5][0]) and rel_l2(out[3.5][1], out[1.0][1]) < 1e-6
    assert rel_l2(out[3.5][2], 3.5 * out[1.0][2]) < 1e-6
    with pytest.raises(ValueError):
        dr.antialias(color.cuda(), rast.cuda(), pos.cuda(), tri.cuda(), topology_hash=topo[:-1])


# ---------------------------------------------------------------------------------------------------------------------
# the reference's mip chain end to end (fit.py:153-155, 160-161, 579) at cfg1
# ---------------------------------------------------------------------------------------------------------------------

def test_mip_chain_end_to_end_matches_oracle(dr, oracle_ops):
    """rasterize (with rast_db) -> interpolate(diff_attrs='all') -> texture('linear-mipmap-linear', max_mip_level) ->
    antialias -> background -> pixel loss: image, loss, d loss / d pos_clip and d loss / d tex against
    oracle.fit.forward_from_clip(enable_mip=True) on the same clip positions; ids bit-exact."""
    from fpc_diffrend_amd import fit, scene
    from oracle import fit as ofit
    sc = scene.cfg('cfg1', n_frames=2)
    cams = (0, 4, 7)
    pos, _ = clip_positions(sc, list(cams), frames=[1])
    H, W = sc.resolution
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing='ij')
    targets = (70 + 60 * torch.sin(0.05 * xx + 0.3) * torch.cos(0.07 * yy)).clamp(0, 140).to(torch.uint8)
    targets = targets.reshape(1, 1, H, W).expand(1, len(cams), H, W).contiguous()
    st = ofit.State(sc, cams)
    p_ref = pos.clone().requires_grad_(True)
    loss_o, image_o, rast_o = ofit.forward_from_clip(st, p_ref, targets, enable_mip=True, max_mip_level=4)
    loss_o.backward()
    dev = 'cuda'
    ctx = dr.RasterizeGLContext(device=dev)
    tri, uv, uv_idx = (torch.tensor(a, device=dev) for a in (sc.pos_idx, sc.uv, sc.uv_idx))
    p = pos.to(dev).requires_grad_(True)
    tex = torch.tensor(sc.texture, device=dev).requires_grad_(True)
    colour, rast = fit.render_from_clip(ctx, p, tri, uv, uv_idx, tex, sc.resolution, True, 4)
    image = torch.where(rast[..., 3:] > 0, colour, torch.tensor(fit.BACKGROUND, device=dev))       # fit.py:161
    ref = targets.reshape(len(cams), H, W, 1).to(dev).float()
    loss = torch.mean((ref - image * 255) ** 2)                                                    # fit.py:579
    loss.backward()
    assert torch.equal(rast[..., 3].int().cpu(), rast_o[..., 3].int())
    assert rel_l2(image, image_o) < TOL
    assert abs(float(loss) - float(loss_o)) < TOL * float(loss_o)
    assert rel_l2(p.grad, p_ref.grad) < TOL, rel_l2(p.grad, p_ref.grad)
    assert rel_l2(tex.grad, st.tex.grad) < TOL, rel_l2(tex.grad, st.tex.grad)


@pytest.mark.parametrize("res,max_mip,boundary,C", [(None, 4, 'wrap', 1), ((97, 131), None, 'wrap', 1), ((150, 200), 2, 'clamp', 1),
                                                    ((64, 96), 0, 'wrap', 1), ((70, 96), 3, 'zero', 1), ((97, 131), 3, 'wrap', 3),
                                                    ((64, 200), 2, 'clamp', 4)])
def test_fused_objective_with_mip_equals_the_operator_chain_and_the_oracle(dr, oracle_ops, res, max_mip, boundary, C):
    """pixel_objective(enable_mip=True) -- the reference's enable_mip branch (fit.py:153-155) inside the three fused kernels, for 1, 3
    and 4 colour channels: the
    footprint from the barycentrics' screen derivatives recomputed per pixel, 'linear-mipmap-linear' over the box-filtered chain,
    gradients to every level folded back into the texture and through the derivative outputs of the rasteriser into the
    vertices -- equals the chain rasterize(output_db) -> interpolate(diff_attrs='all') -> texture(texd, max_mip_level) ->
    antialias -> background -> pixel loss (loss, d/d pos, d/d tex), and at the scene's own size the oracle's."""
    from fpc_diffrend_amd import fit, scene
    sc = scene.cfg('cfg1', n_frames=2)
    if res is not None:
        sc.resolution = res
    cams = (0, 4, 7)
    pos, _ = clip_positions(sc, list(cams), frames=[1])
    H, W = sc.resolution
    dev = 'cuda'
    g = torch.Generator().manual_seed(3)
    ref8 = torch.randint(0, 141, (len(cams), H, W), generator=g, dtype=torch.uint8)
    ctx = dr.RasterizeGLContext(device=dev)
    tri, uv, uv_idx = (torch.tensor(a, device=dev) for a in (sc.pos_idx, sc.uv, sc.uv_idx))
    if boundary != 'wrap':
        uv = uv * 1.2 - 0.1
    # the chain of separate operators + the reference's torch loss
    tex0 = torch.tensor(sc.texture, device=dev)
    if C != 1:
        tex0 = (tex0.repeat(1, 1, C) * torch.linspace(1.0, 0.5, C, device=dev)).contiguous()
    p1 = pos.to(dev).requires_grad_(True)
    t1 = tex0.clone().requires_grad_(True)
    rast, rast_db = dr.rasterize(ctx, p1, tri, sc.resolution)
    texc, texd = dr.interpolate(uv[None], rast, uv_idx, rast_db=rast_db, diff_attrs='all')
    col = dr.texture(t1[None], texc, texd, filter_mode='linear-mipmap-linear', boundary_mode=boundary, max_mip_level=max_mip)
    col = dr.antialias(col, rast, p1, tri)
    img = torch.where(rast[..., 3:] > 0, col, torch.tensor(fit.BACKGROUND, device=dev))
    l1 = torch.mean((ref8.to(dev).float()[..., None] - img * 255) ** 2)
    l1.backward()
    # the fused objective: the two-call form, then the one-pass form (value and gradient from one call; the one the checks below keep)
    for one_pass in (False, True):
        p2 = pos.to(dev).requires_grad_(True)
        t2 = tex0.clone().requires_grad_(True)
        l2 = dr.pixel_objective(ctx, p2, tri, uv, uv_idx, t2, ref8.to(dev), sc.resolution, boundary_mode=boundary, enable_mip=True,
                                max_mip_level=max_mip, one_pass=one_pass)
        l2.backward()
        assert abs(float(l2) - float(l1)) <= 2e-6 * abs(float(l1)), (one_pass, float(l2), float(l1))
        assert rel_l2(p2.grad, p1.grad) < TOL, (one_pass, rel_l2(p2.grad, p1.grad))
        assert rel_l2(t2.grad, t1.grad) < TOL, (one_pass, rel_l2(t2.grad, t1.grad))
    assert float(t1.grad.abs().max()) > 0 and float(p1.grad.abs().max()) > 0
    if res is None and boundary == 'wrap' and C == 1:
        from oracle import fit as ofit
        st = ofit.State(sc, cams)
        p_ref = pos.clone().requires_grad_(True)
        loss_o, _, _ = ofit.forward_from_clip(st, p_ref, ref8.reshape(1, len(cams), H, W), enable_mip=True, max_mip_level=max_mip)
        loss_o.backward()
        assert abs(float(l2) - float(loss_o)) < TOL * float(loss_o)
        assert rel_l2(p2.grad, p_ref.grad) < TOL and rel_l2(t2.grad, st.tex.grad) < TOL
    # the dense two-call form has no mip variant: refused, not silently something else
    with pytest.raises(NotImplementedError):
        dr.pixel_objective(ctx, p2.detach(), tri, uv, uv_idx, t2.detach(), ref8.to(dev), sc.resolution, enable_mip=True, sparse=False)


# ---------------------------------------------------------------------------------------------------------------------
# the reference's own run shape: ONE 1600 x 1200 image, 1024^2 x 1 texture (main.py:28-30, fit.py:525-526)
# ---------------------------------------------------------------------------------------------------------------------

def test_reference_run_shape_one_image_matches_oracle(dr, oracle_ops):
    """H = 1600 is 50 bins, W = 1200 is 37.5: a ragged right edge over the full height.  Operators and fused objective,
    forward + backward, against the float32 oracle (ids and antialias flags bit-exact, floats 1e-4)."""
    from fpc_diffrend_amd import scene
    from test_gpu_large import _one_image_against_oracle
    sc = scene.cfg('ref', n_frames=2)
    assert tuple(sc.resolution) == (1600, 1200) and sc.texture.shape == (1024, 1024, 1)
    pos, _ = clip_positions(sc, [6], frames=[1])
    _one_image_against_oracle(dr, sc, pos, torch.tensor(sc.pos_idx), cam=6, seed=8)


# ---------------------------------------------------------------------------------------------------------------------
# the strided sweep kernels at 1080p (the residual shape of the r2 work-queue faults)
# ---------------------------------------------------------------------------------------------------------------------

def test_1080p_sweep_kernels_equal_the_hinted_launch():
    """k_bins_queue / k_aa_fix_queue / k_render_aa_bwd_queue loop over thousands of bins per workgroup when the launch hints
    are far too small: 18 images of 1920 x 1080 with the hints forced to (3, 2, 5) and the list-form backward give the loss
    and gradients of the default launch."""
    import fpc_diffrend_amd.ops as dr
    from fpc_diffrend_amd import scene
    sc = scene.cfg('cfg3', n_frames=2)
    pos, _ = clip_positions(sc, list(range(9)), frames=[0, 1])
    dev = 'cuda'
    tri, uv, uv_idx = (torch.tensor(a, device=dev) for a in (sc.pos_idx, sc.uv, sc.uv_idx))
    g = torch.Generator().manual_seed(11)
    ref = torch.randint(0, 141, (pos.shape[0],) + tuple(sc.resolution), generator=g, dtype=torch.uint8).to(dev)
    ctx = dr.RasterizeGLContext(device=dev)

    def run(**kw):
        p = pos.to(dev).clone().requires_grad_(True)
        t = torch.tensor(sc.texture, device=dev).clone().requires_grad_(True)
        loss = dr.pixel_objective(ctx, p, tri, uv, uv_idx, t, ref, sc.resolution, **kw)
        loss.backward()
        torch.cuda.synchronize()
        return float(loss), p.grad.double().cpu(), t.grad.double().cpu()

    for one_pass in (True, False):      # the one-call form (two lists: live bins, occupied bins) and the two-call form (three)
        dr.clear_hints()
        try:
            base = run(launch_hints=False, one_pass=one_pass)
            run(queued_backward=True, one_pass=one_pass)                     # leaves its counts behind
            hints = dr._list_hints[next(iter(dr._list_hints))]
            caps = hints.poll()
            assert min(caps[:2] if one_pass else caps) > 1000, caps         # thousands of bins per list at this size
            hints.event = None
            hints.caps, hints.frozen = (3, 2, 5), True      # (frozen: the one-call form's hints; event / update: the two-call form's)
            hints.update = lambda counts: None
            swept = run(queued_backward=True, one_pass=one_pass)
            assert abs(swept[0] - base[0]) <= 1e-6 * abs(base[0])
            assert rel_l2(swept[1], base[1]) < 1e-5 and rel_l2(swept[2], base[2]) < 1e-5
        finally:
            dr.clear_hints()


def test_cfg3_batch_of_288_images_equals_its_chunks(oracle_ops):
    """The headline batch itself (32 frames x 9 views of 1920 x 1080: 587 520 bins on the lists, launch hints, 17 GB of buffers)
    against the same images in 8 calls of 36: an image's pixels and vertices are its own, so the loss is the sum of the chunks'
    (same n_total), the position gradients are those of the chunks (up to the order of the float atomics), and the texture
    gradient is their sum.  The first
    288-image call runs without hints (full grids), the second with the counts the first left behind.
    Slices of the 288-image call also go STRAIGHT TO THE ORACLE: the id planes the call's rasteriser left (diagnostic output
    id_plane_out) bit-exact against the C rasteriser for eight images spread over the batch, and for two of them the antialias pair
    flags (bit-exact) and d loss / d pos_clip (1e-4) against the float32 oracle's own evaluation of that image."""
    import fpc_diffrend_amd.ops as dr
    from fpc_diffrend_amd import scene
    nf = 32
    sc = scene.cfg('cfg3', n_frames=nf)
    pos, _ = clip_positions(sc, list(range(9)), frames=list(range(nf)))
    dev = 'cuda'
    pos = pos.to(dev)
    B = pos.shape[0]
    assert B == 288
    tri, uv, uv_idx = (torch.tensor(a, device=dev) for a in (sc.pos_idx, sc.uv, sc.uv_idx))
    H, W = sc.resolution
    g = torch.Generator(device=dev).manual_seed(23)
    ref = torch.randint(0, 141, (B, H, W), generator=g, dtype=torch.uint8, device=dev)
    tex0 = torch.tensor(sc.texture, device=dev)
    n_total = B * H * W * tex0.shape[2]
    ctx = dr.RasterizeGLContext(device=dev)

    from fpc_diffrend_amd import _lib
    from helpers import decode_aa_flags, decode_id_planes
    from oracle import fit as ofit
    lib = _lib.load()

    def run(sl, **kw):
        p = pos[sl].clone().requires_grad_(True)
        t = tex0.clone().requires_grad_(True)
        loss = dr.pixel_objective(ctx, p, tri, uv, uv_idx, t, ref[sl], sc.resolution, n_total=n_total, **kw)
        loss.backward()
        return loss.detach().double(), p.grad, t.grad.double()

    dr.clear_hints()
    try:
        whole_first = run(slice(0, B))                 # no hints yet: every list kernel at its full grid
        idp = torch.zeros(lib.fpcdr_idplane_bytes(B, H, W), dtype=torch.uint8, device=dev)
        flags = torch.zeros(lib.fpcdr_antialias_flags_bytes(B, H, W) // 8, dtype=torch.int64, device=dev)
        whole = run(slice(0, B), id_plane_out=idp, aa_flags_out=flags)      # sized from the first call's counts
        torch.cuda.synchronize()
        # slices of this call against the oracle
        OY, OX, Wq = (H + 31) // 32, (W + 31) // 32, (W + 63) // 64
        planes = idp.view(torch.int32).reshape(B, OY * OX * 1024)
        fl = flags.reshape(2, B, H, Wq)
        tri_c = tri.cpu()
        for b in (0, 41, 100, 143, 144, 200, 259, 287):
            ids = decode_id_planes(planes[b], 1, H, W)
            ids_ref = oracle_ops.rasterize_ids(pos[b:b + 1].cpu(), tri_c, sc.resolution)
            assert torch.equal(ids, ids_ref), f"image {b} of the 288-image call: {int((ids != ids_ref).sum())} ids differ from the oracle"
        for b in (100, 259):
            o = ofit.smoke_from_clip(sc, pos[b:b + 1].cpu(), ref[b].cpu().reshape(1, 1, H, W), cams=(b % 9,))
            assert torch.equal(decode_aa_flags(fl[:, b:b + 1].contiguous(), 1, H, W), o['aa_flags']), f"image {b}: antialias pair set"
            # the oracle's loss is the mean over ITS image, the call's over the whole batch: d/d pos differs by the factor B
            assert rel_l2(whole[1][b:b + 1] * B, o['grad_pos_clip']) < 1e-4, (b, rel_l2(whole[1][b:b + 1] * B, o['grad_pos_clip']))
        del idp, flags, planes, fl
        # (gradients are sums of float atomics: equal up to the order of the additions)
        assert abs(float(whole[0]) - float(whole_first[0])) <= 1e-9 * abs(float(whole[0]))
        assert rel_l2(whole[1].double().cpu(), whole_first[1].double().cpu()) < 1e-6
        assert rel_l2(whole[2].cpu(), whole_first[2].cpu()) < 1e-6
        loss = torch.zeros((), dtype=torch.float64, device=dev)
        gtex = torch.zeros_like(whole[2])
        for c in range(0, B, 36):
            l, gp, gt = run(slice(c, c + 36))
            loss += l
            gtex += gt
            assert rel_l2(gp.double().cpu(), whole[1][c:c + 36].double().cpu()) < 1e-6, c      # an image's vertices see only its own pixels
        torch.cuda.synchronize()
        assert abs(float(loss) - float(whole[0])) <= 2e-6 * abs(float(whole[0]))
        assert rel_l2(gtex.cpu(), whole[2].cpu()) < 1e-5
        assert torch.isfinite(whole[1]).all() and float(whole[1].abs().max()) > 0
    finally:
        dr.clear_hints()


# ---------------------------------------------------------------------------------------------------------------------
# bin-shaped interpolate backward: last bin row with H % 32 in 1..7 (rows past the image were read)
# ---------------------------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("res", [(100, 100), (36, 64), (900, 1600)])
def test_interpolate_backward_last_bin_row(dr, oracle_ops, res):
    B = 1 if res[0] > 500 else 2
    pos, tri = random_soup(B, 120, 31, size=0.7)
    g = torch.Generator().manual_seed(8)
    attr = torch.randn(1, 360, 2, generator=g)
    rast, _ = oracle_ops.rasterize(pos, tri, res)
    rast = rast.detach()
    gy = torch.randn(B, res[0], res[1], 2, generator=g)
    r_ref = rast.clone().requires_grad_(True)
    o, _ = oracle_ops.interpolate(attr, r_ref, tri)
    (o * gy).sum().backward()
    # rast is the LAST allocation made before the call: a read past its end is a read past the image batch
    r_gpu = rast.cuda().requires_grad_(True)
    o2, _ = dr.interpolate(attr.cuda(), r_gpu, tri.cuda())
    (o2 * gy.cuda()).sum().backward()
    assert rel_l2(o2, o) < TOL and rel_l2(r_gpu.grad, r_ref.grad) < TOL
    assert torch.equal(r_gpu.grad[:, -1].cpu() != 0, r_ref.grad[:, -1] != 0)


# ---------------------------------------------------------------------------------------------------------------------
# one collective through RCCL (backend "nccl") on the one GPU: a one-rank group in a fresh child process
# ---------------------------------------------------------------------------------------------------------------------

RCCL_CHILD = r"""
import os, sys
sys.path.insert(0, {root!r})
import torch
import torch.distributed as tdist
from fpc_diffrend_amd import dist as fdist, _lib
_lib.load()                                    # libfpcdr.so (bound to torch's HIP runtime) is resident beside RCCL
rank, world, _ = fdist.init(backend="nccl", force_group=True)
assert (rank, world) == (0, 1) and tdist.is_initialized() and tdist.get_backend() == "nccl"
dev = torch.device("cuda", 0)
params = [torch.nn.Parameter(torch.randn(n, device=dev)) for n in (150 * 256, 63, 1024 * 1024)]
for i, p in enumerate(params):
    p.grad = torch.full_like(p, float(i + 1))
bucket = fdist.GradBucket(params, dev, always_reduce=True, timed=True)
bucket()
torch.cuda.synchronize()
for i, p in enumerate(params):
    assert torch.equal(p.grad, torch.full_like(p, float(i + 1))), i       # sum over one rank
ms = bucket.reduce_ms()
assert bucket.calls == 1 and ms is not None and ms >= 0.0
maps = open("/proc/self/maps").read()
assert "librccl" in maps, "RCCL was not loaded"
assert "libfpcdr.so" in maps
print("RCCL_OK bytes", bucket.nbytes, "ms", ms)
tdist.destroy_process_group()
"""


def test_rccl_backend_reduces_the_gradient_bucket(tmp_path):
    """backend='nccl' IS RCCL on ROCm.  RCCL refuses two ranks on one device, so the one-GPU box runs a ONE-rank group: the
    library loads beside libfpcdr.so and torch's bundled HIP runtime, builds a communicator and all-reduces dist.GradBucket's
    flat buffer (4.4 MB, the prior-mode payload) once.  The test process itself never touches the GPU."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "rccl_child.py"
    script.write_text(RCCL_CHILD.format(root=ROOT))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1",
               LOCAL_RANK="0")
    env.pop("FPCDR_DIST_BACKEND", None)
    r = subprocess.run([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    out = r.stdout.decode(errors="replace")
    assert r.returncode == 0 and "RCCL_OK" in out, out[-3000:]


# ---------------------------------------------------------------------------------------------------------------------
# near-plane clipping (rule R1): triangles with vertices behind the camera
# ---------------------------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("res,T,seed", [((96, 128), 60, 3), ((160, 200), 400, 4), ((75, 101), 900, 5), ((64, 96), 3000, 6)])
def test_near_plane_clipping_matches_oracle(dr, oracle_ops, res, T, seed):
    """Triangles crossing the near plane are clipped against it and their one or two pieces drawn under the triangle's own index
    (the second piece through the overflow slots of the raster scratch): ids bit-exact, rast / rast_db floats from the ORIGINAL
    vertices to 1e-4, and the gradient of the rasterize -> interpolate -> antialias chain to the clip-space positions."""
    from helpers import near_crossing_soup
    pos, tri = near_crossing_soup(2, T, seed)
    behind = (pos[..., 3].reshape(2, -1, 3) <= 0).any(dim=2)
    ids_ref = oracle_ops.rasterize_ids(pos, tri, res)
    shown = torch.zeros_like(behind)
    for b in range(2):
        shown[b, (ids_ref[b][ids_ref[b] > 0] - 1).long().unique()] = True
    assert int((shown & behind).sum()) >= 5, "clipped triangles must be visible"
    ctx = dr.RasterizeGLContext(device='cuda')
    g = torch.Generator().manual_seed(seed)
    attr = torch.rand(1, pos.shape[1], 3, generator=g)
    gy = torch.randn(2, res[0], res[1], 3, generator=g)
    gdb = torch.randn(2, res[0], res[1], 4, generator=g) * 0.05
    p_ref = pos.clone().requires_grad_(True)
    r_o, db_o = oracle_ops.rasterize(p_ref, tri, res)
    c_o, _ = oracle_ops.interpolate(attr, r_o, tri)
    aa_o = oracle_ops.antialias(c_o, r_o, p_ref, tri)
    ((aa_o * gy).sum() + (db_o * gdb).sum()).backward()
    p_gpu = pos.cuda().requires_grad_(True)
    rast, db = dr.rasterize(ctx, p_gpu, tri.cuda(), res)
    col, _ = dr.interpolate(attr.cuda(), rast, tri.cuda())
    aa = dr.antialias(col, rast, p_gpu, tri.cuda())
    ((aa * gy.cuda()).sum() + (db * gdb.cuda()).sum()).backward()
    assert torch.equal(rast[..., 3].int().cpu(), ids_ref)
    assert rel_l2(rast, r_o) < TOL and rel_l2(db, db_o) < TOL and rel_l2(aa, aa_o) < TOL
    assert rel_l2(p_gpu.grad, p_ref.grad) < TOL, rel_l2(p_gpu.grad, p_ref.grad)
    # the fused forward carries the same rasteriser in its list form
    uv = torch.rand(pos.shape[1], 2, generator=g).cuda()
    tex = torch.rand(16, 16, 1, generator=g).cuda()
    _, rast2 = dr.render_textured(ctx, pos.cuda(), tri.cuda(), uv, tri.cuda(), tex, res)
    assert torch.equal(rast2[..., 3].int().cpu(), ids_ref)
    # ... and so does the one-pass objective, whose set-up kernel files the clipped triangles on per-image lists for k_setup_clip<true>
    # (first pieces appended to the per-bin triangle lists, chunk boxes widened by compare-and-swap; with thousands of triangles on a
    # few bins the lists overflow and the bins fall back to scanning those chunk boxes): value and gradients of the operator chain
    from fpc_diffrend_amd import fit
    ref_img = torch.randint(0, 141, (2,) + tuple(res), generator=g, dtype=torch.uint8).cuda()
    out = []
    for one_pass in (None, True, False):
        p, t = pos.cuda().requires_grad_(True), tex.clone().requires_grad_(True)
        if one_pass is None:
            r3, _ = dr.rasterize(ctx, p, tri.cuda(), res)
            tc, _ = dr.interpolate(uv[None], r3, tri.cuda())
            c3 = dr.antialias(dr.texture(t[None], tc, filter_mode='linear'), r3, p, tri.cuda())
            img = torch.where(r3[..., 3:] > 0, c3, torch.tensor(fit.BACKGROUND, device='cuda'))
            loss = torch.mean((ref_img[..., None].float() - img * 255) ** 2)
        else:
            loss = dr.pixel_objective(ctx, p, tri.cuda(), uv, tri.cuda(), t, ref_img, res, one_pass=one_pass)
        loss.backward()
        out.append((float(loss), p.grad.double().cpu(), t.grad.double().cpu()))
    for l, gp, gt in out[1:]:
        assert abs(l - out[0][0]) <= 1e-5 * abs(out[0][0]), (l, out[0][0])
        assert rel_l2(gp, out[0][1]) < TOL and rel_l2(gt, out[0][2]) < TOL, (rel_l2(gp, out[0][1]), rel_l2(gt, out[0][2]))


def test_stale_mip_stack_is_refused_and_hints_refresh_without_backward(dr):
    """(i) a stack from texture_construct_mip() whose base texture was modified in place (an optimiser step) must not be
    demoted to a custom stack silently -- its levels are stale and their gradient would no longer reach tex; (ii) the launch hints of
    the two-call objective are refreshed also when no backward pass will run (torch.no_grad(), evaluation loops); (iii) ref_bg_sumsq
    may be a Python float."""
    from fpc_diffrend_amd import scene
    sc = scene.cfg('cfg1', n_frames=1)
    dev = 'cuda'
    pos, _ = clip_positions(sc, [0, 4], frames=[0])
    tri, uv, uv_idx = (torch.tensor(a, device=dev) for a in (sc.pos_idx, sc.uv, sc.uv_idx))
    ctx = dr.RasterizeGLContext(device=dev)
    tex = torch.rand(1, 64, 64, 1, device=dev).requires_grad_(True)
    rast, rast_db = dr.rasterize(ctx, pos.to(dev), tri, sc.resolution)
    texc, texd = dr.interpolate(uv[None], rast, uv_idx, rast_db=rast_db, diff_attrs='all')
    stack = dr.texture_construct_mip(tex, max_mip_level=3)
    opt = torch.optim.SGD([tex], lr=0.1)
    for step in range(2):
        if step == 1:
            with pytest.raises(RuntimeError, match="modified in place"):
                dr.texture(tex, texc, texd, mip=stack, filter_mode='linear-mipmap-linear', max_mip_level=3)
            stack = dr.texture_construct_mip(tex, max_mip_level=3)
        col = dr.texture(tex, texc, texd, mip=stack, filter_mode='linear-mipmap-linear', max_mip_level=3)
        opt.zero_grad()
        col.sum().backward()
        assert float(tex.grad.abs().sum()) > 0
        opt.step()
    # (ii) + (iii)
    ref = torch.full((pos.shape[0],) + tuple(sc.resolution), 90, dtype=torch.uint8, device=dev)
    bg = float(dr.reference_background_sumsq(ref).sum())
    p = pos.to(dev).clone().requires_grad_(True)
    t = torch.tensor(sc.texture, device=dev).clone().requires_grad_(True)
    for one_pass in (False, True):
        dr.clear_hints()
        with torch.no_grad():
            a = dr.pixel_objective(ctx, p, tri, uv, uv_idx, t, ref, sc.resolution, one_pass=one_pass, ref_bg_sumsq=bg)
        torch.cuda.synchronize()
        hints = dr._list_hints[next(iter(dr._list_hints))]
        assert hints.poll()[0] > 0, "the counts of a forward-only call were never read back"
        b = dr.pixel_objective(ctx, p, tri, uv, uv_idx, t, ref, sc.resolution, one_pass=one_pass)
        assert abs(float(a) - float(b)) <= 1e-6 * abs(float(b))
    dr.clear_hints()


def test_region_hint_is_dropped_for_anything_but_the_producers_own_tensor(dr):
    """ops._hint_of honours a hint only for the very tensor rasterize() returned, owning its storage, unmodified (version counter);
    a view, a re-pointed tensor, an in-place edit, drop_hints() or no_region_hints() all fall back to the dense path."""
    from fpc_diffrend_amd import scene
    sc = scene.cfg('cfg1', n_frames=1)
    pos, _ = clip_positions(sc, [0, 4], frames=[0])
    tri = torch.tensor(sc.pos_idx, device='cuda')
    ctx = dr.RasterizeGLContext(device='cuda')
    rast, _ = dr.rasterize(ctx, pos.cuda(), tri, sc.resolution)
    assert dr._hint_of(rast, 'rast') is not None
    assert dr._hint_of(rast[0:1], 'rast') is None and dr._hint_of(rast.view(-1, 4), 'rast') is None      # views
    assert dr._hint_of(rast.clone(), 'rast') is None
    with dr.no_region_hints():
        assert dr._hint_of(rast, 'rast') is None
    assert dr._hint_of(rast, 'rast') is not None
    other = torch.zeros_like(rast)
    keep = rast.data
    rast.data = other                       # re-pointed at foreign storage
    assert dr._hint_of(rast, 'rast') is None
    rast.data = keep
    rast.mul_(1.0)                          # an in-place edit bumps the version counter
    assert dr._hint_of(rast, 'rast') is None
    rast2, _ = dr.rasterize(ctx, pos.cuda(), tri, sc.resolution)
    dr.drop_hints(rast2)
    assert dr._hint_of(rast2, 'rast') is None
