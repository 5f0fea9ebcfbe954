"""One row of the float-rules table ("Accuracy bars"): for the library FPCDR_LIB_PATH names (default: the in-tree build,
-ffp-contract=off) the distance of the HIP path's image and gradients from the float32 oracle and from the float64 oracle on the
smoke scene, and the cfg3 time of the objective call.  Run once per build variant, e.g.
    bash scripts/build_variant.sh contract -ffp-contract=fast
    python scripts/float_rules_table.py; FPCDR_LIB_PATH=$PWD/fpc_diffrend_amd/libfpcdr_contract.so python scripts/float_rules_table.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from fpc_diffrend_amd import _lib, fit, scene
from oracle import fit as ofit

def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / max(float(b.norm()), 1e-30))

cams = (0, 4)
sc = scene.cfg('cfg1', n_frames=2)
targets = fit.smoke_targets(sc, cams)
res = fit.smoke_step(sc, device='cuda:0', cams=cams)
pos_clip = res['pos_clip'].cpu()
ref = ofit.smoke_from_clip(sc, pos_clip, targets, cams)
ref64 = ofit.smoke_from_clip(sc, pos_clip, targets, cams, dtype=torch.float64, ids=ref['ids'])
row = {"lib": os.path.basename(_lib.LIB_PATH), "ids_equal": bool(torch.equal(res['ids'].cpu(), ref['ids']))}
for k in ("grad_pos_clip", "grad_tex"):
    row[k] = {"operators_vs_f32": rel(res[k], ref[k]), "objective_vs_f32": rel(res[k + "_fused"], ref[k]),
              "operators_vs_f64": rel(res[k], ref64[k]), "objective_vs_f64": rel(res[k + "_fused"], ref64[k]),
              "f32_oracle_vs_f64": rel(ref[k], ref64[k])}
row["image"] = {"operators_vs_f32": rel(res['image'], ref['image']), "operators_vs_f64": rel(res['image'], ref64['image']),
                "f32_oracle_vs_f64": rel(ref['image'], ref64['image'])}
# cfg3 time of the objective call
sc3 = scene.cfg("cfg3", n_frames=32)
ft = fit.Fitter(sc3, fit.FitConfig(max_iter=80000, init_texture="random"), device="cuda")
for _ in range(3):
    ft.step()
torch.cuda.synchronize()
t = _lib.KernelTimer(names=["fpcdr_objective_fwd"]); _lib.TIMER = t
for _ in range(5):
    ft.step()
_lib.TIMER = None
row["cfg3_objective_ms"] = {k: v[1] / v[0] for k, v in t.summary().items()}
print(json.dumps(row))
