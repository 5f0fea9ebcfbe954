"""Share of the shaded pixels of the plain (non-mip) k_shade whose four texel adds fall inside the bin's LDS window (a -DFPCDR_MIPSTAT build:
bash scripts/build_one_variant.sh STAT objective -DFPCDR_MIPSTAT; FPCDR_LIB_PATH=fpc_diffrend_amd/libfpcdr_STAT.so python scripts/tex_window_stats.py)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fpc_diffrend_amd import fit, scene
sc = scene.cfg('cfg3', n_frames=32)
ft = fit.Fitter(sc, fit.FitConfig(max_iter=80000, init_texture="random"), device="cuda")
ft.step()
torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 16)()
lib = ctypes.CDLL(os.environ["FPCDR_LIB_PATH"])
assert lib.fpcdr_debug_mipstat(out) == 0
v = list(out)
print(f"pixels whose taps go through the window: {v[13]} ({100 * v[13] / (v[13] + v[14]):.1f} %), to memory: {v[14]} ({100 * v[14] / (v[13] + v[14]):.1f} %)")
print(f"  of those, outside a window that held the whole footprint of pass 0: {v[15]}")
