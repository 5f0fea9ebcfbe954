import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fpc_diffrend_amd import _lib, scene, camera
import fpc_diffrend_amd.ops as dr
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from helpers import clip_positions
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 8
sc = scene.cfg('cfg3', n_frames=nf)
pos, _ = clip_positions(sc, list(range(9)), frames=list(range(nf)))
pos = pos.cuda(); tri = torch.tensor(sc.pos_idx).cuda()
ctx = dr.RasterizeGLContext(output_db=False)
for _ in range(2): dr.rasterize(ctx, pos, tri, sc.resolution)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): dr.rasterize(ctx, pos, tri, sc.resolution)
torch.cuda.synchronize(); print(os.environ.get('FPCDR_LIB_PATH', 'default'), 'rasterize_fwd ms', (time.perf_counter() - t0) / 5 * 1e3)
