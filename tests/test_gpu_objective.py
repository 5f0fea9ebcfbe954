"""The one-pass pixel objective (fpcdr_objective_fwd: value and gradient from one call, the shading kernel chains every pixel's
gradient back itself; reference fit.py:151-161, 579, 611) against the two-call form and the chain of the four operators."""
import pytest
import torch

from helpers import rel_l2, clip_positions, random_soup

pytestmark = pytest.mark.gpu


def _inputs(geom, C, res, seed=1):
    from fpc_diffrend_amd import scene
    dev = 'cuda'
    sc = scene.cfg('cfg1', n_frames=2)
    sc.resolution = res
    if geom == 'mesh':
        pos, _ = clip_positions(sc, [0, 3, 7], frames=[0, 1])
        tri = torch.tensor(sc.pos_idx, device=dev)
        uv = torch.tensor(sc.uv, device=dev) * 1.2 - 0.05
        uv_idx = torch.tensor(sc.uv_idx, device=dev)
    else:      # an open soup: every triangle edge is a silhouette edge, overlaps everywhere -> many deferred pixels, depth pairs
        n = {'soup': 60, 'few': 5}[geom]
        pos, tri = random_soup(3, n, seed=21 + seed, spread=0.8, size=0.5 if geom == 'soup' else 0.9)
        tri = tri.to(dev)
        g0 = torch.Generator().manual_seed(8)
        uv = (torch.rand(3 * n, 2, generator=g0) * 1.2 - 0.1).to(dev)
        uv_idx = tri.clone()
    g = torch.Generator().manual_seed(seed)
    tex = torch.rand(48, 64, C, generator=g) * 0.5
    ref = torch.randint(0, 141, (pos.shape[0], res[0], res[1]), generator=g, dtype=torch.uint8).to(dev)
    return pos.to(dev), tri, uv, uv_idx, tex.to(dev), ref


def _chain(ctx, p, tri, uv, uv_idx, t, ref, res, boundary):
    import fpc_diffrend_amd.ops as dr
    from fpc_diffrend_amd import fit
    rast, _ = dr.rasterize(ctx, p, tri, res)
    texc, _ = dr.interpolate(uv[None], rast, uv_idx)
    col = dr.antialias(dr.texture(t[None], texc, filter_mode='linear', boundary_mode=boundary), rast, p, tri)
    img = torch.where(rast[..., 3:] > 0, col, torch.tensor(fit.BACKGROUND, device=p.device))
    return torch.mean((ref[..., None].float() - img * 255) ** 2)


@pytest.mark.parametrize("geom,C,res,boundary", [('mesh', 1, (150, 200), 'wrap'), ('soup', 1, (97, 131), 'wrap'), ('soup', 3, (128, 160), 'clamp'),
                                                ('few', 4, (64, 320), 'zero'), ('soup', 1, (33, 65), 'zero'), ('mesh', 3, (256, 256), 'wrap'),
                                                ('few', 1, (32, 32), 'wrap'), ('soup', 1, (31, 17), 'clamp'),
                                                ('few', 1, (97, 131), 'wrap')])      # (big triangles over a ragged border: the tile path)
def test_one_pass_equals_two_call_form_and_operator_chain(geom, C, res, boundary):
    import fpc_diffrend_amd.ops as dr
    pos, tri, uv, uv_idx, tex, ref = _inputs(geom, C, res)
    ctx = dr.RasterizeGLContext(device='cuda')
    out = {}
    for name in ("one", "two", "chain"):
        p, t = pos.clone().requires_grad_(True), tex.clone().requires_grad_(True)
        if name == "chain":
            loss = _chain(ctx, p, tri, uv, uv_idx, t, ref, res, boundary)
        else:
            loss = dr.pixel_objective(ctx, p, tri, uv, uv_idx, t, ref, res, boundary_mode=boundary, one_pass=(name == "one"), launch_hints=False)
        loss.backward()
        out[name] = (float(loss), p.grad.double().cpu(), t.grad.double().cpu())
    for other in ("two", "chain"):
        assert abs(out["one"][0] - out[other][0]) <= 2e-6 * abs(out[other][0]), (other, out["one"][0], out[other][0])
        assert rel_l2(out["one"][1], out[other][1]) < 1e-4, other
        assert rel_l2(out["one"][2], out[other][2]) < 1e-4, other


def test_one_pass_value_only_and_single_gradients_and_upstream():
    """Without gradients the call computes the value alone; with one input requiring a gradient only that one is produced; the
    upstream scalar multiplies both (unit_upstream=True hands the buffers over as they are)."""
    import fpc_diffrend_amd.ops as dr
    pos, tri, uv, uv_idx, tex, ref = _inputs('soup', 1, (97, 131))
    ctx = dr.RasterizeGLContext(device='cuda')
    res = (97, 131)
    p, t = pos.clone().requires_grad_(True), tex.clone().requires_grad_(True)
    full = dr.pixel_objective(ctx, p, tri, uv, uv_idx, t, ref, res)
    full.backward()
    with torch.no_grad():
        v = dr.pixel_objective(ctx, p, tri, uv, uv_idx, t, ref, res)
    assert float(v) == float(full) and not v.requires_grad
    v2 = dr.pixel_objective(ctx, pos, tri, uv, uv_idx, tex, ref, res)
    assert float(v2) == float(full)
    p2 = pos.clone().requires_grad_(True)
    dr.pixel_objective(ctx, p2, tri, uv, uv_idx, tex, ref, res).backward()
    assert rel_l2(p2.grad, p.grad) < 1e-6
    t2 = tex.clone().requires_grad_(True)
    dr.pixel_objective(ctx, pos, tri, uv, uv_idx, t2, ref, res).backward()
    assert rel_l2(t2.grad, t.grad) < 1e-6
    p3, t3 = pos.clone().requires_grad_(True), tex.clone().requires_grad_(True)
    (dr.pixel_objective(ctx, p3, tri, uv, uv_idx, t3, ref, res) * 3.5).backward()
    assert rel_l2(p3.grad, 3.5 * p.grad) < 1e-6 and rel_l2(t3.grad, 3.5 * t.grad) < 1e-6
    p4, t4 = pos.clone().requires_grad_(True), tex.clone().requires_grad_(True)
    dr.pixel_objective(ctx, p4, tri, uv, uv_idx, t4, ref, res, unit_upstream=True).backward()
    assert rel_l2(p4.grad, p.grad) < 1e-6 and rel_l2(t4.grad, t.grad) < 1e-6


def test_one_pass_launch_hints_do_not_change_the_result():
    import fpc_diffrend_amd.ops as dr
    pos, tri, uv, uv_idx, tex, ref = _inputs('mesh', 1, (150, 200))
    ctx = dr.RasterizeGLContext(device='cuda')

    def run(**kw):
        p, t = pos.clone().requires_grad_(True), tex.clone().requires_grad_(True)
        loss = dr.pixel_objective(ctx, p, tri, uv, uv_idx, t, ref, (150, 200), **kw)
        loss.backward()
        torch.cuda.synchronize()
        return float(loss), p.grad.double().cpu(), t.grad.double().cpu()

    base = run(launch_hints=False)
    dr.clear_hints()
    first = run()
    key = next(k for k in dr._list_hints if k[0] == 'onepass')
    hints = dr._list_hints[key]
    caps = hints.poll()
    assert caps[0] >= 256 and caps[1] >= 256
    second = run()
    hints.caps, hints.frozen = (3, 2, 1), True      # absurdly small: almost every bin goes through the strided sweeps
    third = run()
    for r in (first, second, third):
        assert abs(r[0] - base[0]) <= 1e-6 * abs(base[0])
        assert rel_l2(r[1], base[1]) < 1e-5 and rel_l2(r[2], base[2]) < 1e-5
    dr.clear_hints()


def test_one_pass_edge_cases_clipping_empty_images_and_the_indirect_uv_path(monkeypatch):
    """(i) triangles crossing the near plane (rule R1: clipped pieces under the triangle's id) and images that show nothing at all;
    (ii) the C ABI's tri_uv = NULL path (uv looked up through uv_tri inside the kernels; the binding always pre-gathers);
    (iii) the mip branch with absurdly small launch hints (its strided sweep kernels do the work): all against the operator chain."""
    import fpc_diffrend_amd.ops as dr
    from helpers import near_crossing_soup
    dev = 'cuda'
    ctx = dr.RasterizeGLContext(device=dev)
    res = (96, 128)
    pos, tri = near_crossing_soup(3, 40, seed=5)
    pos[2, :, 3] = -1.0                                   # image 2: everything behind the camera -> an empty image
    pos, tri = pos.to(dev), tri.to(dev)
    g0 = torch.Generator().manual_seed(4)
    uv = (torch.rand(tri.shape[0] * 3, 2, generator=g0) * 1.1).to(dev)
    uv_idx = tri.clone()
    tex = (torch.rand(32, 32, 1, generator=g0) * 0.5).to(dev)
    ref = torch.randint(0, 141, (3,) + res, generator=g0, dtype=torch.uint8).to(dev)

    def both(**kw):
        out = []
        for name in ("one", "chain"):
            p, t = pos.clone().requires_grad_(True), tex.clone().requires_grad_(True)
            if name == "chain":
                if kw.get("enable_mip"):
                    from fpc_diffrend_amd import fit
                    rast, rdb = dr.rasterize(ctx, p, tri, res)
                    texc, texd = dr.interpolate(uv[None], rast, uv_idx, rast_db=rdb, diff_attrs='all')
                    col = dr.antialias(dr.texture(t[None], texc, texd, filter_mode='linear-mipmap-linear', max_mip_level=2), rast, p, tri)
                    img = torch.where(rast[..., 3:] > 0, col, torch.tensor(fit.BACKGROUND, device=dev))
                    loss = torch.mean((ref[..., None].float() - img * 255) ** 2)
                else:
                    loss = _chain(ctx, p, tri, uv, uv_idx, t, ref, res, 'wrap')
            else:
                loss = dr.pixel_objective(ctx, p, tri, uv, uv_idx, t, ref, res, **kw)
            loss.backward()
            out.append((float(loss), p.grad.double().cpu(), t.grad.double().cpu()))
        (l1, gp1, gt1), (l2, gp2, gt2) = out
        assert abs(l1 - l2) <= 2e-6 * abs(l2), (l1, l2)
        assert rel_l2(gp1, gp2) < 1e-4 and rel_l2(gt1, gt2) < 1e-4, (rel_l2(gp1, gp2), rel_l2(gt1, gt2))
        assert float(gp1[2].abs().max()) == 0.0            # the empty image contributes no gradient

    both()
    monkeypatch.setattr(dr, "_cached_tri_uv", lambda uv_, idx_: None)      # (ii)
    both()
    monkeypatch.undo()
    dr.clear_hints()                                 # (iii)
    both(enable_mip=True, max_mip_level=2)
    h = dr._list_hints[next(k for k in dr._list_hints if k[0] == 'onepass')]
    h.caps, h.frozen = (1, 1, 1), True
    both(enable_mip=True, max_mip_level=2)
    dr.clear_hints()


@pytest.mark.parametrize("geom,C,res", [('soup', 1, (97, 131)), ('mesh', 3, (256, 256)), ('few', 4, (64, 320))])
def test_compact_records_equal_records_by_pixel_and_a_short_pool_says_so(geom, C, res):
    """fpcdr_objective_params.rec_slots: the deferred pixels' records in slots of 1 024 (one per bin that shows a silhouette triangle)
    instead of addressed by pixel -- same value, gradients and antialias flags; the call counts the slots it needs (counts_out[1]), a
    counting call reports the same number without shading, and a pool that is too small raises the overflow flag (counts_out[5])."""
    import ctypes
    import fpc_diffrend_amd.ops as dr
    from fpc_diffrend_amd import _lib
    pos, tri, uv, uv_idx, tex, ref = _inputs(geom, C, res)
    ctx = dr.RasterizeGLContext(device='cuda')
    B = pos.shape[0]
    nflag = _lib.load().fpcdr_antialias_flags_bytes(B, res[0], res[1]) // 8

    def run(slots):
        dr.clear_hints()
        p, t = pos.clone().requires_grad_(True), tex.clone().requires_grad_(True)
        flags = torch.zeros(nflag, dtype=torch.int64, device='cuda')
        loss = dr.pixel_objective(ctx, p, tri, uv, uv_idx, t, ref, res, record_slots=slots, aa_flags_out=flags)
        loss.backward()
        torch.cuda.synchronize()
        h = dr._list_hints[next(k for k in dr._list_hints if k[0] == 'onepass')]
        counts = [int(v) for v in h.host[:6].tolist()]
        return float(loss), p.grad.double().cpu(), t.grad.double().cpu(), flags.cpu(), counts

    dense = run(0)
    need = dense[4][1]
    assert need == 0 and dense[4][5] == 0      # (records by pixel: no slot taken)
    roomy = run(4096)
    need = roomy[4][1]
    assert need > 0 and roomy[4][5] == 0
    exact = run(need)
    assert exact[4][1] == need and exact[4][5] == 0
    for r in (roomy, exact):
        assert abs(r[0] - dense[0]) <= 1e-6 * abs(dense[0])
        assert rel_l2(r[1], dense[1]) < 1e-5 and rel_l2(r[2], dense[2]) < 1e-5
        assert torch.equal(r[3], dense[3])
    if need > 1:
        short = run(need - 1)
        assert short[4][1] == need and short[4][5] == 1
    dr.clear_hints()


def test_large_batch_sizes_its_record_pool_from_a_counting_call_then_from_the_last_call(monkeypatch):
    """Beyond SMALL_BATCH_BINS bins the binding runs compact by itself: the first call on a shape counts (count_only), later ones take 1.5 x
    the last call's slots; results equal the records-by-pixel form; an overflow seen at the next call raises."""
    import fpc_diffrend_amd.ops as dr
    pos, tri, uv, uv_idx, tex, ref = _inputs('soup', 1, (97, 131))
    ctx = dr.RasterizeGLContext(device='cuda')
    res = (97, 131)

    def run(**kw):
        p, t = pos.clone().requires_grad_(True), tex.clone().requires_grad_(True)
        loss = dr.pixel_objective(ctx, p, tri, uv, uv_idx, t, ref, res, **kw)
        loss.backward()
        torch.cuda.synchronize()
        return float(loss), p.grad.double().cpu(), t.grad.double().cpu()

    dr.clear_hints()
    dense = run(record_slots=0)
    dr.clear_hints()
    monkeypatch.setattr(dr, "SMALL_BATCH_BINS", 0)
    monkeypatch.setattr(dr, "RECORD_SLOT_MARGIN", 1)
    first = run()
    h = dr._list_hints[next(k for k in dr._list_hints if k[0] == 'onepass')]
    assert h.sil_bins > 0 and h.slots == h.sil_bins + max(1, h.sil_bins // 2)
    second = run()
    for r in (first, second):
        assert abs(r[0] - dense[0]) <= 1e-6 * abs(dense[0])
        assert rel_l2(r[1], dense[1]) < 1e-5 and rel_l2(r[2], dense[2]) < 1e-5
    h.poll()
    h.slots, h.frozen = 1, True      # (a pool that cannot hold the batch)
    run()
    h.frozen = False
    with pytest.raises(RuntimeError, match="ran out of record slots"):
        run()
    third = run()      # (the shape counts afresh and carries on)
    assert abs(third[0] - dense[0]) <= 1e-6 * abs(dense[0])
    dr.clear_hints()


@pytest.mark.parametrize("tex_hw", [(4, 4), (2, 8), (16, 4), (64, 64)])
@pytest.mark.parametrize("boundary", ['wrap', 'clamp', 'zero'])
def test_texel_windows_with_tiny_textures_and_coordinates_far_outside_the_unit_square(tex_hw, boundary):
    """The shading kernels sum texel gradients in LDS windows shaped by a bin's footprint: rectangles in unwrapped texel
    coordinates, cells found modulo the texture's size.  Textures smaller than a window, footprints across many periods and all three
    boundary modes, with and without mip levels, against the operator chain."""
    import fpc_diffrend_amd.ops as dr
    from fpc_diffrend_amd import fit
    dev = 'cuda'
    ctx = dr.RasterizeGLContext(device=dev)
    res, B, nt = (96, 128), 2, 40
    Ht, Wt = tex_hw
    pos, tri = random_soup(B, nt, seed=Ht * 31 + Wt, spread=0.8, size=0.7)
    tri = tri.to(dev)
    g = torch.Generator().manual_seed(3)
    tex0 = torch.rand(Ht, Wt, 1, generator=g) * 0.6
    ref = torch.randint(0, 141, (B,) + res, generator=g, dtype=torch.uint8).to(dev)
    for scale in (0.3, 6.0):
        uv = ((torch.rand(3 * nt, 2, generator=g) - 0.3) * scale).to(dev)
        for mip in (False, True):
            out = {}
            for name in ('chain', 'one'):
                p, t = pos.to(dev).clone().requires_grad_(True), tex0.to(dev).clone().requires_grad_(True)
                if name == 'chain':
                    rast, rdb = dr.rasterize(ctx, p, tri, res)
                    if mip:
                        texc, texd = dr.interpolate(uv[None], rast, tri, rast_db=rdb, diff_attrs='all')
                        col = dr.texture(t[None], texc, texd, filter_mode='linear-mipmap-linear', boundary_mode=boundary, max_mip_level=1)
                    else:
                        texc, _ = dr.interpolate(uv[None], rast, tri)
                        col = dr.texture(t[None], texc, filter_mode='linear', boundary_mode=boundary)
                    img = torch.where(rast[..., 3:] > 0, dr.antialias(col, rast, p, tri), torch.tensor(fit.BACKGROUND, device=dev))
                    loss = torch.mean((ref[..., None].float() - img * 255) ** 2)
                else:
                    loss = dr.pixel_objective(ctx, p, tri, uv, tri, t, ref, res, boundary_mode=boundary, enable_mip=mip, max_mip_level=1)
                loss.backward()
                out[name] = (float(loss.detach()), p.grad.double().cpu(), t.grad.double().cpu())
            assert abs(out['one'][0] - out['chain'][0]) <= 3e-6 * abs(out['chain'][0]), (scale, mip)
            assert rel_l2(out['one'][1], out['chain'][1]) < 1e-4 and rel_l2(out['one'][2], out['chain'][2]) < 1e-4, (scale, mip)
