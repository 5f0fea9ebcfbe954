"""Wave-cycles of k_shade by phase (a -DFPCDR_OPROF build of objective.hip: scripts/build_obj_variant.sh oprof -DFPCDR_OPROF; run with
FPCDR_LIB_PATH=fpc_diffrend_amd/libfpcdr_oprof.so).  Prints the share of each phase."""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fpc_diffrend_amd import _lib, fit, scene
sc = scene.cfg("cfg3", n_frames=32)
ft = fit.Fitter(sc, fit.FitConfig(max_iter=80000, init_texture="random"), device="cuda")
for _ in range(3):
    ft.step()
torch.cuda.synchronize()
lib = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * 16)()
lib.fpcdr_debug_oprof(buf, 1)
n = 5
for _ in range(n):
    ft.step()
torch.cuda.synchronize()
lib.fpcdr_debug_oprof(buf, 0)
v = list(buf)
# (the bin is shaded as two halves, each: first pass -> window set-up with one barrier -> kept taps + second pass; the timers of the
#  set-up are taken in the SECOND half, so the first half lies wholly inside the third interval)
names = ["prologue: id plane, apron, tables", "barrier (plane in place)", "first half (2 passes, window, flush) + second half's first pass",
         "barrier (second half's window)", "second half: kept taps + second pass", "barrier (end of shading)", "epilogue: vertex table + window flush"]
tot = sum(v[:7])
print(json.dumps({"k_fix pass0: pixels, bins with deferred pixels (per launch)": [v[8] / n, v[9] / n],
                  "k_fix pass1: pixels, bins": [v[10] / n, v[11] / n]}))
print(json.dumps({"k_fix pass1 per working wave: cycles to list, cycles in pixel loop, waves": [v[12] / max(v[14], 1), v[13] / max(v[14], 1), v[14] / n]}))
print(json.dumps({"waves_per_launch": v[7] / n, "cycles_per_wave": tot / max(v[7], 1),
                  "share": {k: round(x / tot, 4) for k, x in zip(names, v[:7])},
                  "cycles_per_wave_by_phase": {k: round(x / max(v[7], 1), 1) for k, x in zip(names, v[:7])}}))
