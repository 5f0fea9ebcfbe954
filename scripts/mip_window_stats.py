"""Where the texel-gradient taps of the MIP shading kernel go (a -DFPCDR_MIPSTAT build of objective.hip: bash scripts/build_one_variant.sh mipSTAT
objective -DFPCDR_MIPSTAT; FPCDR_LIB_PATH=fpc_diffrend_amd/libfpcdr_mipSTAT.so python scripts/mip_window_stats.py): per level offset from the
bin's finest level, inside / outside its LDS window."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fpc_diffrend_amd import fit, scene, _lib
sc = scene.cfg('cfg3', n_frames=32)
ft = fit.Fitter(sc, fit.FitConfig(max_iter=80000, init_texture="random", enable_mip=True, max_mip_level=6), device="cuda")
ft.step()
torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 16)()
lib = ctypes.CDLL(os.environ["FPCDR_LIB_PATH"])
assert lib.fpcdr_debug_mipstat(out) == 0
v = list(out)
tot = sum(v[:7])
print("pixels with a gradient:", v[7])
for j in range(3):
    print(f"level lb+{j}: in window {v[j]:>12d} ({100 * v[j] / tot:5.1f} %)   outside {v[3 + j]:>12d} ({100 * v[3 + j] / tot:5.1f} %)")
print(f"beyond lb+2: {v[6]:>12d} ({100 * v[6] / tot:5.1f} %)")
print("outside a window that fitted the sampled footprint:", v[8:11], " finer than lb:", v[11], " no sampled pixel in the bin:", v[12])
