import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import torch
import fpc_diffrend_amd.ops as dr
from fpc_diffrend_amd import fit, scene
from helpers import clip_positions, rel_l2
sc = scene.cfg('cfg1', n_frames=2)
pos, _ = clip_positions(sc, [0, 3, 7], frames=[0, 1])
dev = 'cuda'
tri = torch.tensor(sc.pos_idx, device=dev)
uv = torch.tensor(sc.uv, device=dev); uv_idx = torch.tensor(sc.uv_idx, device=dev)
ref = torch.full((pos.shape[0],) + tuple(sc.resolution), 90, dtype=torch.uint8, device=dev)
ctx = dr.RasterizeGLContext(device=dev)
out = {}
for name in ("dense", "sparse"):
    p = pos.to(dev).clone().requires_grad_(True)
    t = torch.tensor(sc.texture, device=dev).clone().requires_grad_(True)
    print(name, "forward...", flush=True)
    loss = dr.pixel_objective(ctx, p, tri, uv, uv_idx, t, ref, sc.resolution, sparse=(name == "sparse"))
    torch.cuda.synchronize()
    print(name, "forward ok", float(loss), flush=True)
    if os.environ.get("FPCDR_DEBUG_STAGE"):
        cm = loss.grad_fn.dbg_cmask.cpu()
        B, (H, W) = pos.shape[0], sc.resolution
        nb = B * ((H + 31) // 32) * ((W + 31) // 32)
        hdr_off = nb * 1152
        import numpy as np
        hdr = np.frombuffer(cm.numpy().tobytes()[hdr_off:hdr_off + 64], dtype=np.int32)
        lst = np.frombuffer(cm.numpy().tobytes()[hdr_off + 64:hdr_off + 64 + 4 * nb], dtype=np.int32)
        live = np.frombuffer(cm.numpy().tobytes()[hdr_off + 64 + 8 * nb:hdr_off + 64 + 9 * nb], dtype=np.uint8)
        if os.environ.get("FPCDR_DEBUG_FIXMODE") == "4":
            d = np.frombuffer(cm.numpy().tobytes()[:16 * 8], dtype=np.int32).reshape(8, 4)
            print("dbg rows (n, popped, *count, cursor-count):", d.tolist())
        print("nb", nb, "hdr", hdr.tolist(), "live sum", int(live.sum()), "live max", int(live.max()), "list[:n] range", lst[:max(hdr[0],1)].min(), lst[:max(hdr[0],1)].max(),
              "unique", len(set(lst[:hdr[0]].tolist())))
        sys.exit(0)
    loss.backward()
    torch.cuda.synchronize()
    print(name, "backward ok", flush=True)
    out[name] = (float(loss), p.grad.clone(), t.grad.clone())
print("loss", out["dense"][0], out["sparse"][0], "gp", rel_l2(out["sparse"][1], out["dense"][1]), "gt", rel_l2(out["sparse"][2], out["dense"][2]))
