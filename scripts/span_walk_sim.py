"""Instruction-count model of two forms of the rasteriser's lane walk (rasterize.hip, process_batch) on the cfg3 rig, with the exact
coverage of every (triangle, bin, row): the box walk as it is -- per column step 8 instructions for every lane plus 10 more whenever
ANY active lane of the wave is covered -- against a span walk -- per row ~57 instructions to find the first and last covered column
of each lane exactly (three reciprocal estimates with integer fix-ups), then 11 per step over the longest span among the wave's
lanes.  DESIGN.md 8 quotes its output.  No GPU.   python scripts/span_walk_sim.py [camera] [every n-th bin]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from fpc_diffrend_amd import scene
from helpers import clip_positions
cam = int(sys.argv[1]) if len(sys.argv) > 1 else 4
every = int(sys.argv[2]) if len(sys.argv) > 2 else 3
sc = scene.cfg('cfg3', n_frames=2)
H, W = sc.resolution
tri = np.asarray(sc.pos_idx)
pos, _ = clip_positions(sc, [cam], frames=[1]); p = pos[0].double().numpy()
w = p[:, 3]
X = np.floor((p[:, 0] / w * 0.5 + 0.5) * (W * 256) + 0.5).astype(np.int64); Y = np.floor((p[:, 1] / w * 0.5 + 0.5) * (H * 256) + 0.5).astype(np.int64)
tx, ty = X[tri], Y[tri]
D = (tx[:, 1] - tx[:, 0]) * (ty[:, 2] - ty[:, 0]) - (ty[:, 1] - ty[:, 0]) * (tx[:, 2] - tx[:, 0])
x0 = np.maximum((tx.min(1) - 128 + 255) // 256, 0); x1 = np.minimum((tx.max(1) - 128) // 256, W - 1)
y0 = np.maximum((ty.min(1) - 128 + 255) // 256, 0); y1 = np.minimum((ty.max(1) - 128) // 256, H - 1)
ok = (D != 0) & (x0 <= x1) & (y0 <= y1)
bins = {}
for t in np.nonzero(ok)[0]:
    sg = 1 if D[t] > 0 else -1
    e = []
    for (a, b) in ((1, 2), (2, 0), (0, 1)):      # edge e: A = -(Yb - Ya) s, B = (Xb - Xa) s, anchored at vertex a
        A = -(ty[t, b] - ty[t, a]) * sg; Bc = (tx[t, b] - tx[t, a]) * sg
        n = 0 if ((-A > 0) or (A == 0 and Bc < 0)) else 1
        e.append((A, Bc, tx[t, a], ty[t, a], n))
    for by in range(y0[t] // 32, y1[t] // 32 + 1):
        for bx in range(x0[t] // 32, x1[t] // 32 + 1):
            if (by * 1000 + bx) % every:
                continue
            cx0, cx1 = max(x0[t], bx * 32), min(x1[t], bx * 32 + 31); cy0, cy1 = max(y0[t], by * 32), min(y1[t], by * 32 + 31)
            px = (np.arange(cx0, cx1 + 1) * 256 + 128)[None, :]; py = (np.arange(cy0, cy1 + 1) * 256 + 128)[:, None]
            cov = np.ones((cy1 - cy0 + 1, cx1 - cx0 + 1), dtype=bool)
            for (A, Bc, xa, ya, n) in e:
                cov &= (A * (px - xa) + Bc * (py - ya) - n) >= 0
            bins.setdefault((by, bx), []).append(cov)

def walk(covs, span):
    n = len(covs); tot = 0
    for base in range(0, n, 256):
        b = covs[base:base + 256]; m = len(b)
        split = 4 if m <= 64 else (2 if m <= 128 else 1)
        thr = [(c, part) for part in range(split) for c in b]
        for w0 in range(0, len(thr), 64):
            wv = thr[w0:w0 + 64]
            rows = [c[part::split] for c, part in wv]      # the rows each lane walks, in order
            iters = max(r.shape[0] for r in rows)
            cost = 110      # record fetch + edge set-up
            for j in range(iters):
                act = [r[j] for r in rows if r.shape[0] > j]
                if span:
                    cost += 57 + 12 + 11 * max(int(a.sum()) for a in act)
                else:
                    bw = max(a.shape[0] for a in act)
                    anyc = np.zeros(bw, dtype=bool)
                    for a in act:
                        anyc[:a.shape[0]] |= a
                    cost += 12 + 8 * bw + 10 * int(anyc.sum())
            tot += cost
    return tot

nb = len(bins)
cur = sum(walk(v, False) for v in bins.values()); spn = sum(walk(v, True) for v in bins.values())
cov = sum(int(c.sum()) for v in bins.values() for c in v); box = sum(c.size for v in bins.values() for c in v)
print(f"camera {cam}: {nb} bins sampled, {sum(len(v) for v in bins.values()) / nb:.0f} triangles per bin, {cov / box:.2f} of the box samples covered")
print(f"wave-instructions per bin: box walk {cur / nb:.0f} (per wave {cur / nb / 4:.0f}; measured 495), span walk {spn / nb:.0f} = {100 * (spn / cur - 1):+.0f} %")
