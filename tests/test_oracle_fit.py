"""CPU: the oracle fit step (BASELINE.json configs[0]: 1-view 256x256, 10 blendshapes, 1k triangles, PyTorch-CPU
software raster) runs end to end and its loss goes down -- the plumbing the CPU baseline and smoke() rely on."""
import torch

from fpc_diffrend_amd import scene
from oracle import fit as ofit


def test_oracle_fit_cfg1_descends(oracle_ops):
    sc = scene.cfg('cfg1', n_frames=2)
    st = ofit.State(sc, cams=[3])
    # targets: ground truth rendered by the oracle itself, 8 bit, clipped to [0,140] (reference fit.py:531)
    gt = ofit.State(sc, cams=[3])
    with torch.no_grad():
        gt.M1.copy_(torch.eye(2))
        gt.M2.copy_(torch.tensor(sc.weights_gt).t())
        gt.per_frame_t.copy_(torch.tensor(sc.t_gt))
        gt.per_frame_q.copy_(torch.tensor(sc.q_gt))
        _, img, _ = ofit.forward(gt, torch.arange(2), torch.zeros(2, 1, 256, 256, dtype=torch.uint8))
        targets = torch.clamp(torch.round(img[..., 0] * 255), 0, 140).to(torch.uint8).reshape(2, 1, 256, 256)
    with torch.no_grad():
        # start inside the basin of the checkered texture: 80 % of the true activations / translation
        st.M1.copy_(torch.eye(2))
        st.M2.copy_(0.8 * torch.tensor(sc.weights_gt).t())
        st.per_frame_t.copy_(0.8 * torch.tensor(sc.t_gt))
        st.per_frame_q.copy_(torch.tensor(sc.q_gt))
        at_truth, _, _ = ofit.forward(gt, torch.arange(2), targets)
    assert float(at_truth) < 0.1          # only 8-bit quantisation separates the ground truth from its target
    opt = torch.optim.Adam([st.M2, st.per_frame_t], lr=5e-3)
    losses = []
    for _ in range(8):
        opt.zero_grad()
        loss, _, _ = ofit.forward(st, torch.arange(2), targets)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert losses[-1] < 0.7 * losses[0], losses
    from fpc_diffrend_amd import fit
    r = ofit.smoke_step(sc, fit.smoke_targets(sc, (0, 4)))
    assert all(torch.isfinite(v).all() for k, v in r.items() if v is not None)
    assert r['grad_w'].abs().max() > 0 and r['grad_tex'].abs().max() > 0 and r['grad_pose'].abs().max() > 0


def test_oracle_camera_matches_reference_golden():
    """The oracle restates the reference's camera matrices itself (it imports nothing from the product): pinned against
    the outputs of the reference's own camera.py captured in tests/golden/camera_golden.json."""
    import json
    import os
    import numpy as np
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "camera_golden.json")))
    for name, c in gold["cameras"].items():
        P = ofit.intrinsic_to_projection(np.asarray(c["intrinsic"], dtype=np.float32))
        MV = ofit.extrinsic_to_modelview(np.asarray(c["rotation"], dtype=np.float32), np.asarray(c["translation"], dtype=np.float32))
        assert P.dtype == np.float32 and str(MV.dtype) == c["MV_dtype"]
        np.testing.assert_array_equal(P, np.asarray(c["P"], dtype=np.float32))
        np.testing.assert_array_equal(MV, np.asarray(c["MV"], dtype=np.float32))
    np.testing.assert_array_equal(ofit.translate(0.0, 170.0, 0.0), np.asarray(gold["translate_0_170_0"], dtype=np.float32))


def test_oracle_blend_matches_reference_golden():
    """oracle.fit.vertices (the three blend modes, reference fit.py:47-129) against tests/golden/blend_golden.json: the outputs
    and autograd gradients of the reference's OWN blend / blend_free / blend_combined, run on the CPU by tests/golden/make_golden.py
    (the functions are taken out of the reference file's syntax tree; the module itself cannot be imported)."""
    import json
    import os
    import types
    import numpy as np
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "blend_golden.json")))
    inp = {k: (torch.tensor(np.asarray(v, dtype=np.float32)) if isinstance(v, list) else v) for k, v in gold["inputs"].items()}
    assert inp["learned_coefficient"] == 0.5        # the literal of fit.py:562, which oracle.fit.vertices hard-codes
    F = inp["M1"].shape[0]
    for mode, case in (("prior", "blend"), ("free", "blend_free"), ("combined", "blend_combined")):
        leaves = gold["cases"][case]["grads"].keys()
        t = {k: inp[k].clone().requires_grad_(k in leaves) for k in ("v_base", "Bmat", "M1", "M2", "m1", "m2", "m3")}
        st = types.SimpleNamespace(mode=mode, **t)
        out = ofit.vertices(st, torch.arange(F))                    # [F, 3V]: all frames as one batch
        want = torch.tensor(np.asarray(gold["cases"][case]["out"], dtype=np.float32))
        assert out.shape == want.shape
        assert float((out.detach() - want).abs().max()) <= 2e-6 * float(want.abs().max()), case
        (out * inp["gy"]).sum().backward()
        for k, g in gold["cases"][case]["grads"].items():
            g = torch.tensor(np.asarray(g, dtype=np.float32))
            err = float((t[k].grad - g).norm() / g.norm())
            assert err < 1e-5, (case, k, err)


def test_oracle_trainer_follows_the_reference_update_rules(oracle_ops):
    """reference fit.py:493-505, 603-618: ten Adam groups in the reference's order and learning rates, lr * ramp^(i/max_iter),
    whole-tensor quaternion division (quirk Q3), and the combined mode's learned basis receiving its first gradient in the
    iteration AFTER the first i > max_iter / 2."""
    import numpy as np
    from fpc_diffrend_amd import fit
    sc = scene.cfg('cfg1', n_frames=2)
    cams = (3,)
    st, F = ofit.perturbed_state(sc, cams, mode='combined')
    tr = ofit.Trainer(st, max_iter=2, lr_base=1e-3, lr_tex_coef=0.5, lr_ramp=0.005, lr_t=1e-5, lr_q=2e-5, weight_laplacian=50.0)
    lrs = [g['lr'] for g in tr.optimizer.param_groups]
    assert np.allclose(lrs, [1e-4, 1e-4, 1e-4, 1e-3, 1e-3, 1e-5, 2e-5, 1e-5, 2e-5, 5e-4])
    assert [p is q for p, q in zip((g['params'][0] for g in tr.optimizer.param_groups), st.params())] == [True] * 10
    targets = fit.smoke_targets(sc, cams)
    m3_before = st.m3.detach().clone()
    seen = []
    for i in range(4):
        tr.step(torch.arange(F), targets)
        seen.append(bool(st.m3.requires_grad) and not torch.equal(st.m3.detach(), m3_before))
        assert abs(tr.optimizer.param_groups[3]['lr'] - 1e-3 * 0.005 ** ((i + 1) / 2)) < 1e-12
        # Q3: after the division the WHOLE tensor has unit Frobenius norm (rows: 1/3 resp. 1/sqrt(F))
        assert abs(float(st.q_opt.norm()) - 1.0) < 1e-6 and abs(float(st.per_frame_q.norm()) - 1.0) < 1e-6
    assert seen == [False, False, False, True]      # i = 2 switches after its forward; i = 3 is the first update of m3
    assert abs(float(st.q_opt[0].norm()) - 1.0 / 3.0) < 1e-4
