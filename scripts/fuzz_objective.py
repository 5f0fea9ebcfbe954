"""Random shapes through the fused objective against the chain of separate operators + the reference's torch loss: image count,
resolution (ragged against the 32-pixel bin and the 64-pixel flag word), triangle soup size, channels, boundary mode, mip,
launch hints on a second call.  Prints one line per case and fails on the first mismatch.   python scripts/fuzz_objective.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, torch
import fpc_diffrend_amd.ops as dr
from fpc_diffrend_amd import fit
from helpers import random_soup, rel_l2
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
dev = 'cuda'
ctx = dr.RasterizeGLContext(device=dev)
for case in range(n_cases):
    B = int(rng.integers(1, 12)); H = int(rng.integers(20, 300)); W = int(rng.integers(20, 420))
    nt = int(rng.choice([3, 17, 120, 900])); C = int(rng.choice([1, 1, 3, 4]))
    boundary = str(rng.choice(['wrap', 'clamp', 'zero'])); mip = bool(rng.random() < 0.4)
    pos, tri = random_soup(B, nt, seed=int(rng.integers(1 << 30)), spread=float(rng.uniform(0.4, 1.1)), size=float(rng.uniform(0.05, 0.9)))
    tri = tri.to(dev)
    g = torch.Generator().manual_seed(case)
    uv = (torch.rand(3 * nt, 2, generator=g) * 1.3 - 0.15).to(dev); uv_idx = tri.clone()
    tex0 = torch.rand(32, 64, C, generator=g) * 0.6
    ref = torch.randint(0, 141, (B, H, W), generator=g, dtype=torch.uint8).to(dev)
    out = {}
    dr.clear_hints()
    # fused = the one-pass objective (value + gradient in one call); compact = with the deferred pixels' records in slots sized by a counting
    # call, then by the previous call (what batches beyond ops.SMALL_BATCH_BINS bins do by themselves)
    small_batch_bins = dr.SMALL_BATCH_BINS
    for name in ("chain", "fused", "fused again (launch hints)", "compact records", "compact again", "two-call form"):
        dr.SMALL_BATCH_BINS = 0 if name.startswith("compact") else small_batch_bins
        if name == "compact records":
            dr.clear_hints()
        p = pos.to(dev).clone().requires_grad_(True); t = tex0.to(dev).clone().requires_grad_(True)
        if name == "chain":
            rast, rdb = dr.rasterize(ctx, p, tri, (H, W))
            if mip:
                texc, texd = dr.interpolate(uv[None], rast, uv_idx, rast_db=rdb, diff_attrs='all')
                col = dr.texture(t[None], texc, texd, filter_mode='linear-mipmap-linear', boundary_mode=boundary, max_mip_level=3)
            else:
                texc, _ = dr.interpolate(uv[None], rast, uv_idx)
                col = dr.texture(t[None], texc, filter_mode='linear', boundary_mode=boundary)
            col = dr.antialias(col, rast, p, tri)
            img = torch.where(rast[..., 3:] > 0, col, torch.tensor(fit.BACKGROUND, device=dev))
            loss = torch.mean((ref[..., None].float() - img * 255) ** 2)
        else:
            loss = dr.pixel_objective(ctx, p, tri, uv, uv_idx, t, ref, (H, W), boundary_mode=boundary, enable_mip=mip, max_mip_level=3,
                                      one_pass=(name != "two-call form"), queued_backward=True)
        loss.backward()
        torch.cuda.synchronize()
        out[name] = (float(loss), p.grad.double().cpu(), t.grad.double().cpu())
    ok = True
    for name in list(out)[1:]:
        dl = abs(out[name][0] - out["chain"][0]) / max(abs(out["chain"][0]), 1e-30)
        gp = rel_l2(out[name][1], out["chain"][1]) if float(out["chain"][1].abs().max()) > 0 else float(out[name][1].abs().max())
        gt = rel_l2(out[name][2], out["chain"][2]) if float(out["chain"][2].abs().max()) > 0 else float(out[name][2].abs().max())
        ok &= dl <= 3e-6 and gp < 1e-4 and gt < 1e-4
        print(f"case {case:2d} B={B} {H}x{W} T={nt} C={C} {boundary:5s} mip={int(mip)} {name[:13]:13s} dloss {dl:.1e} dpos {gp:.1e} dtex {gt:.1e}", flush=True)
    assert ok, "mismatch"
print("all cases agree")
