"""GPU: a fit step whose pixel objective ran out of record slots is SKIPPED ON THE DEVICE (include/fpcdr.h ABI v11:
fpcdr_objective_params.skip_out -> dist.GradBucket.flag -> fpcdr_adam_params.skip_flag / skipped): parameters and Adam moments stay as
they were, nothing raises, the run continues, and what follows is the run that never drew the skipped iteration (reference
fit.py:610-613 except for that iteration).  The shortage is forced through the launch-hint record of the batch shape."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _onepass_hints(dr):
    return dr._list_hints[next(k for k in dr._list_hints if k[0] == 'onepass')]


def _make(targets=None):
    from fpc_diffrend_amd import fit, scene
    sc = scene.cfg('cfg1', n_frames=2)
    sc.q_gt[:] = (0.0, 0.0, 0.0, 1.0)
    # max_iter = 40: LambdaLR's lr_ramp^(i / max_iter) falls by 12 % per step -- a skipped step that left the schedule one step ahead
    # would show at once
    cfg = fit.FitConfig(max_iter=40, cam_idxs=(0, 3, 6), lr_base=5e-3, lr_t=5e-3, lr_q=1e-5, weight_laplacian=40.0, init_texture='truth')
    ft = fit.Fitter(sc, cfg, device='cuda', targets=targets)
    ft.init_near_truth(0.8)
    return ft


def _snapshot(ft):
    state = [p.detach().clone() for p in ft.params]
    for p in ft.params:
        st = ft.optimizer.state.get(p, {})
        if 'exp_avg' in st:
            state += [st['exp_avg'].clone(), st['exp_avg_sq'].clone()]
    return state


def _close(ft_a, ft_b, steps_apart=1):
    """Two Fitters that took the same step from the same state: equal up to the order of the float atomics inside a step -- every entry
    of every tensor to 1e-5, except that Adam turns a texel gradient that cancels to ~0 with no history into a step of the learning
    rate whose sign is that order's: the texture and its moments may hold a vanishing share of such entries."""
    sa, sb = _snapshot(ft_a), _snapshot(ft_b)
    assert len(sa) == len(sb)
    n_par = len(ft_a.params)
    tex_like = {i for i, t in enumerate(sa) if t.shape == ft_a.tex_opt.shape}
    for i, (a, b) in enumerate(zip(sa, sb)):
        d = (a - b).abs()
        tol = 1e-5 * max(1.0, float(b.abs().max())) if i < n_par else 1e-4 * max(float(b.abs().max()), 1e-30)
        if i in tex_like:
            assert float((d > tol).float().mean()) <= 2e-4 and float(d.max()) <= 2.0 * 2.5e-3 * steps_apart, (i, float(d.max()))
        else:
            assert float(d.max()) <= tol, (i, tuple(a.shape), float(d.max()), tol)


def test_short_record_pool_skips_the_update_on_the_device_and_the_run_continues(tmp_path, monkeypatch):
    import fpc_diffrend_amd.ops as dr
    dr.clear_hints()
    monkeypatch.setattr(dr, "SMALL_BATCH_BINS", 0)       # compact records for this small batch too
    monkeypatch.setattr(dr, "RECORD_SLOT_MARGIN", 1)
    k = 3
    ft = _make()
    for _ in range(k):
        assert math.isfinite(float(ft.step()))
    ft.save_checkpoint(str(tmp_path / "before.pt"))      # the state in front of iteration k
    before = _snapshot(ft)
    h = _onepass_hints(dr)
    h.poll()
    need = h.sil_bins
    assert need > 1
    h.slots, h.frozen = 1, True                          # iteration k: a pool that cannot hold the batch
    bad = ft.step()
    h.frozen = False
    assert math.isnan(float(bad))                        # the call says so in its value ...
    assert ft.skipped_steps == 1                         # ... the update kernel counted the skip ...
    for a, b in zip(_snapshot(ft), before):
        assert torch.equal(a, b)                         # ... and touched neither parameters nor moments (nor the quaternion division)
    # the run continues without an exception; the next call is sized from the demand the short call counted
    assert math.isfinite(float(ft.step()))
    h.poll()
    assert h.skipped_calls == 1 and h.overflowed is None and h.slots >= need
    # ... as the run that never drew the skipped iteration: a second Fitter resumes from the state in front of iteration k and takes
    # its step k with host counters and a schedule that never saw the skipped one -- the same update (the kernel re-forms the bias
    # corrections for step - skipped and the learning rate of one schedule step earlier, in double).  max_iter = 40 puts 12 % between two
    # consecutive learning rates and Adam's first bias corrections differ by 20 % from step to step: an update formed for the wrong
    # step would be off by 1e-3, not 1e-5
    plain = _make(targets=ft.targets)
    plain.load_checkpoint(str(tmp_path / "before.pt"))
    for a, b in zip(_snapshot(plain), before):
        assert torch.equal(a, b)
    plain.step()
    assert plain.skipped_steps == 0
    _close(ft, plain)
    ft.step()
    plain.step()
    _close(ft, plain, steps_apart=2)
    assert ft.skipped_steps == 1 and ft.iteration == plain.iteration + 1
    dr.clear_hints()


def test_skip_counter_survives_a_checkpoint(tmp_path, monkeypatch):
    import fpc_diffrend_amd.ops as dr
    dr.clear_hints()
    monkeypatch.setattr(dr, "SMALL_BATCH_BINS", 0)
    monkeypatch.setattr(dr, "RECORD_SLOT_MARGIN", 1)
    a = _make()
    a.step()
    h = _onepass_hints(dr)
    h.poll()
    h.slots, h.frozen = 1, True
    a.step()
    h.frozen = False
    a.step()
    a.save_checkpoint(str(tmp_path / "ck.pt"))
    want = [float(a.step()) for _ in range(3)]
    b = _make(targets=a.targets)
    b.load_checkpoint(str(tmp_path / "ck.pt"))
    assert b.skipped_steps == 1
    got = [float(b.step()) for _ in range(3)]
    assert np.allclose(got, want, rtol=1e-5), (got, want)
    dr.clear_hints()


def test_a_caller_without_skip_out_still_learns_of_the_overflow(monkeypatch):
    """ops.pixel_objective without skip_out= (nobody to skip the update): the short call's value is NaN and the NEXT call on the shape
    raises -- also when another call on the shape finished in between (the device's overflow count is cumulative)."""
    import fpc_diffrend_amd.ops as dr
    from test_gpu_objective import _inputs
    pos, tri, uv, uv_idx, tex, ref = _inputs('soup', 1, (97, 131))
    ctx = dr.RasterizeGLContext(device='cuda')
    res = (97, 131)
    dr.clear_hints()
    monkeypatch.setattr(dr, "SMALL_BATCH_BINS", 0)
    monkeypatch.setattr(dr, "RECORD_SLOT_MARGIN", 1)

    def run(**kw):
        p, t = pos.clone().requires_grad_(True), tex.clone().requires_grad_(True)
        loss = dr.pixel_objective(ctx, p, tri, uv, uv_idx, t, ref, res, **kw)
        loss.backward()
        torch.cuda.synchronize()
        return float(loss.detach())

    def enqueue(**kw):      # (no synchronisation: the host runs ahead of the device, as a fit loop does)
        p, t = pos.clone().requires_grad_(True), tex.clone().requires_grad_(True)
        return dr.pixel_objective(ctx, p, tri, uv, uv_idx, t, ref, res, **kw)

    good = run()
    h = _onepass_hints(dr)
    h.poll()
    need = h.sil_bins
    h.slots, h.frozen = 1, True
    assert math.isnan(run())
    h.frozen = False
    with pytest.raises(RuntimeError, match="ran out of record slots"):
        run()
    assert abs(run() - good) <= 1e-6 * abs(good)         # (the shape counts afresh and carries on)
    # the short call followed at once by a good one, nobody polling in between: the device's overflow count is cumulative, so the
    # second call's counters do not wipe it (they did: the flag was overwritten by every call)
    h.poll()
    h.slots, h.frozen = 1, True
    short = enqueue()
    h.frozen, h.slots = False, need + 8
    with pytest.raises(RuntimeError, match="ran out of record slots"):
        later = enqueue()                                # (raises here only if the short call has already landed)
        torch.cuda.synchronize()
        assert math.isnan(float(short)) and abs(float(later) - good) <= 1e-6 * abs(good)
        run()
    assert abs(run() - good) <= 1e-6 * abs(good)
    # with skip_out the same shortage is reported through the flag, in the same call, and nothing raises later
    flag = torch.full((1,), 7.0, device='cuda')
    assert abs(run(skip_out=flag) - good) <= 1e-6 * abs(good) and float(flag) == 0.0
    h.poll()
    h.slots, h.frozen = 1, True
    assert math.isnan(run(skip_out=flag)) and float(flag) == 1.0
    h.frozen = False
    assert abs(run(skip_out=flag) - good) <= 1e-6 * abs(good) and float(flag) == 0.0
    dr.clear_hints()
