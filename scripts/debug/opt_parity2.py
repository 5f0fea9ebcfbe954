import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np, torch
from fpc_diffrend_amd import fit, scene
import fpc_diffrend_amd.ops as dr
from oracle import fit as ofit, ops as O
from helpers import rel_l2
O.build()
mode = 'prior'
cams = (0, 4)
sc = scene.cfg('cfg1', n_frames=2)
targets = fit.smoke_targets(sc, cams)
hp = dict(max_iter=100, lr_base=2e-3, lr_tex_coef=0.5, lr_ramp=0.005, lr_t=1e-3, lr_q=1e-4, weight_laplacian=0.0)
st, F = ofit.perturbed_state(sc, cams, mode=mode)
start = [p.detach().clone() for p in st.params()]
for variant in ('default', 'dense', 'unfused'):
    kw = {'default': {}, 'dense': dict(sparse_objective=False), 'unfused': dict(fused_objective=False, fused_render=False, fused_loss=False)}[variant]
    ft = fit.Fitter(sc, fit.FitConfig(cam_idxs=cams, mode=mode, **hp, **kw), device='cuda', targets=targets.cuda())
    with torch.no_grad():
        for p, v in zip(ft.params, start):
            p.copy_(v.cuda())
    grabbed = {}
    orig = fit.transform_clip_batched
    def spy(mvp, verts):
        out = orig(mvp, verts)
        out.register_hook(lambda g: grabbed.__setitem__('g', g.detach().clone()))
        grabbed['pos'] = out.detach()
        return out
    fit.transform_clip_batched = spy
    loss = ft.loss_and_backward(slice(0, 2))
    fit.transform_clip_batched = orig
    pos = grabbed['pos'].cpu()
    ref = ofit.smoke_from_clip(sc, pos, targets, cams, texture=start[9].numpy())
    # oracle state identical to start
    st2, _ = ofit.perturbed_state(sc, cams, mode=mode)
    pc, _ = ofit.clip_positions(st2, torch.arange(F))
    print(variant, 'loss', float(loss), float(ref['loss']), 'pos rel', rel_l2(pos, pc.detach()), 'max abs', float((pos - pc.detach()).abs().max()))
    print('   grad_pos_clip vs oracle on same pos', rel_l2(grabbed['g'], ref['grad_pos_clip']), 'tex', rel_l2(ft.tex_opt.grad, ref['grad_tex']))
    e2e = ofit.smoke_from_clip(sc, pc.detach(), targets, cams, texture=start[9].numpy())
    print('   ids mismatches gpu-pos vs cpu-pos', int((ref['ids'] != e2e['ids']).sum()), 'flags mism', int((ref['aa_flags'] != e2e['aa_flags']).sum()),
          'grad_pos_clip cpu-pos vs gpu-pos', rel_l2(e2e['grad_pos_clip'], ref['grad_pos_clip']))
    up = ofit.smoke_upstream(sc, grabbed['g'].cpu(), cams)
    print('   upstream: M2', rel_l2(ft.maps_intermediate['local'].grad, up['grad_w']), 'pose', rel_l2(torch.cat([ft.per_frame_t.grad.reshape(-1), ft.per_frame_q.grad.reshape(-1), ft.t_opt.grad.reshape(-1), ft.q_opt.grad.reshape(-1)]), up['grad_pose']))
    up2 = ofit.smoke_upstream(sc, e2e['grad_pos_clip'], cams)
    print('   M2 grad e2e-oracle vs upstream(gpu g):', rel_l2(up2['grad_w'], up['grad_w']))
