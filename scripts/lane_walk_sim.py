"""Instruction-count model of the rasteriser's lane path (rasterize.hip, process_batch) on the cfg3 rig: wave-instructions per bin
for the rows-outside / columns-inside box walk as it is (one lane per triangle, 2 / 4 lanes when a batch is small), with the
triangles sorted by box size, with one lane per (triangle, row), and for perfectly packed lanes.  DESIGN.md 4.5 (round 3) quotes
its output.  No GPU, no oracle: scene + camera only.   python scripts/lane_walk_sim.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from fpc_diffrend_amd import scene
from helpers import clip_positions
sc = scene.cfg('cfg3', n_frames=2)
H,W = sc.resolution
tri = np.asarray(sc.pos_idx)
res=[]
allw=[]
for cam in (0,4,7):
    pos,_ = clip_positions(sc,[cam],frames=[1]); p = pos[0].double().numpy()
    w = p[:,3]; X = np.floor((p[:,0]/w*0.5+0.5)*(W*256)+0.5); Y = np.floor((p[:,1]/w*0.5+0.5)*(H*256)+0.5)
    tx = X[tri]; ty = Y[tri]
    D = (tx[:,1]-tx[:,0])*(ty[:,2]-ty[:,0]) - (ty[:,1]-ty[:,0])*(tx[:,2]-tx[:,0])
    ok = D!=0
    x0 = np.floor((tx.min(1)-128+255)/256).astype(int); x1=np.floor((tx.max(1)-128)/256).astype(int)
    y0 = np.floor((ty.min(1)-128+255)/256).astype(int); y1=np.floor((ty.max(1)-128)/256).astype(int)
    x0=np.maximum(x0,0); y0=np.maximum(y0,0); x1=np.minimum(x1,W-1); y1=np.minimum(y1,H-1)
    ok &= (x0<=x1)&(y0<=y1)
    idx = np.nonzero(ok)[0]
    bins={}
    for t in idx:
        for by in range(y0[t]//32, y1[t]//32+1):
            for bx in range(x0[t]//32, x1[t]//32+1):
                cx0=max(x0[t],bx*32); cx1=min(x1[t],bx*32+31); cy0=max(y0[t],by*32); cy1=min(y1[t],by*32+31)
                bins.setdefault((by,bx),[]).append((cx1-cx0+1, cy1-cy0+1))
    # cost models
    def cur(lst):
        n=len(lst); tot=0
        for base in range(0,n,256):
            b = lst[base:base+256]; m=len(b)
            split = 4 if m<=64 else (2 if m<=128 else 1)
            thr=[(bw,bh,part) for part in range(split) for (bw,bh) in b]
            for w0 in range(0,len(thr),64):
                wv=thr[w0:w0+64]
                iters=max((bh-part+split-1)//split for bw,bh,part in wv)
                c=0
                for j in range(iters):
                    act=[bw for bw,bh,part in wv if (bh-part+split-1)//split>j]
                    c+= max(act)*8+12
                tot+=c+110
        return tot
    def sortd(lst):   # same but triangles sorted by area (lanes of a wave alike)
        return cur(sorted(lst,key=lambda a:(a[1],a[0])))
    def rows(lst):    # item = (triangle,row): lanes take rows
        items=[bw for bw,bh in lst for _ in range(bh)]
        items.sort()
        tot=0
        for p0 in range(0,len(items),256):
            chunk=items[p0:p0+256]
            for w0 in range(0,len(chunk),64):
                wv=chunk[w0:w0+64]; tot+=max(wv)*8+40
        return tot
    def rows_unsorted(lst):
        items=[bw for bw,bh in lst for _ in range(bh)]
        tot=0
        for p0 in range(0,len(items),256):
            chunk=items[p0:p0+256]
            for w0 in range(0,len(chunk),64):
                wv=chunk[w0:w0+64]; tot+=max(wv)*8+40
        return tot
    c1=sum(cur(v) for v in bins.values()); c2=sum(sortd(v) for v in bins.values()); c3=sum(rows(v) for v in bins.values()); c4=sum(rows_unsorted(v) for v in bins.values())
    nb=len(bins); ntr=sum(len(v) for v in bins.values()); samples=sum(bw*bh for v in bins.values() for bw,bh in v)
    print(f"cam {cam}: bins {nb} tri-bin pairs {ntr} ({ntr/nb:.0f}/bin) samples {samples} ({samples/ntr:.1f}/tri) ; wave-instr per bin: current {c1/nb:.0f} sorted {c2/nb:.0f} rows-sorted {c3/nb:.0f} rows-unsorted {c4/nb:.0f}; ideal {samples*8/64/nb:.0f}")
    allw += [bw*bh for v in bins.values() for bw,bh in v]
a=np.array(allw); print('area pct 50/90/99/max', np.percentile(a,[50,90,99]), a.max(), 'mean',a.mean())
# alternative splits
def model(lst, split, setup=110, inner=8, outer=12, order='part-major'):
    tot=0; per=256//split
    for base in range(0,len(lst),per):
        b=lst[base:base+per]
        if order=='part-major': thr=[(bw,bh,part) for part in range(split) for (bw,bh) in b]
        else: thr=[(bw,bh,part) for (bw,bh) in b for part in range(split)]
        for w0 in range(0,len(thr),64):
            wv=thr[w0:w0+64]
            iters=max((bh-part+split-1)//split for bw,bh,part in wv)
            c=0
            for j in range(iters):
                act=[bw for bw,bh,part in wv if (bh-part+split-1)//split>j]
                c+=max(act)*inner+outer
            tot+=c+setup
    return tot
# column split: each of `split` lanes takes every split-th column? (inner loop shorter)
def model_cols(lst, split, setup=110, inner=8, outer=12):
    tot=0; per=256//split
    for base in range(0,len(lst),per):
        b=lst[base:base+per]
        thr=[(( bw-part+split-1)//split,bh) for (bw,bh) in b for part in range(split)]
        for w0 in range(0,len(thr),64):
            wv=thr[w0:w0+64]
            iters=max(bh for bw,bh in wv); c=0
            for j in range(iters):
                act=[bw for bw,bh in wv if bh>j]
                c+=max(act)*inner+outer
            tot+=c+setup
    return tot

nb = len(bins)
for sp in (1, 2, 4, 8):
    print('every triangle split over', sp, 'lanes:', round(sum(model(v, sp) for v in bins.values()) / nb), 'wave-instructions per bin (part-major),',
          round(sum(model(v, sp, order='lane') for v in bins.values()) / nb), '(lane-major)')

# r4: the (triangle, row) form with its real costs.  Phase 1 is what the lane path does today per triangle (record + box fetch, edge set-up:
# 110 per wave of triangles) plus 14 dwords of set-up stored to LDS and a block-wide prefix sum of the box heights (~40 per wave, 4 waves);
# an item then finds its triangle by bisection of the prefix array (8 dependent LDS reads, ~32 instructions), reads the set-up back
# (four ds_read_b128) and moves the edge functions to its row (~12): ~60 per wave of items, not the 40 the r3 model assumed.
def rows_real(lst, item_setup=60):
    tot = 0
    for base in range(0, len(lst), 256):
        b = lst[base:base + 256]
        tot += -(-len(b) // 64) * (110 + 20) + 4 * 40
        items = [bw for bw, bh in b for _ in range(bh)]
        for w0 in range(0, len(items), 64):
            tot += max(items[w0:w0 + 64]) * 8 + item_setup
    return tot
for s in (40, 60, 80):
    print(f'(triangle, row) items with phase 1 and {s} instructions of item set-up:', round(sum(rows_real(v, s) for v in bins.values()) / nb),
          'wave-instructions per bin; current', round(sum(cur(v) for v in bins.values()) / nb))
