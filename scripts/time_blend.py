"""HIP-event time of the MFMA blend kernels at the cfg3 size (M = 45006, K = 150, F = 32)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fpc_diffrend_amd import fit
M, K, F = 3 * 15002, 150, 32
g = torch.Generator().manual_seed(0)
vb, Bm, w = torch.randn(M, generator=g).cuda(), torch.randn(M, K, generator=g).cuda(), torch.randn(F, K, generator=g).cuda().requires_grad_(True)
go = torch.randn(F, M, generator=g).cuda()
big = torch.empty(64 * 1024 * 1024, device='cuda')   # 256 MB: evicts Bmat from the Infinity Cache between repetitions
def timed(fn, reps=20):
    ts = []
    for _ in range(reps):
        big.fill_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]
out = fit.blend_batched(vb, Bm, w)
t_f = timed(lambda: fit.blend_batched(vb, Bm, w.detach()))
def bwd():
    w.grad = None
    out = fit.blend_batched(vb, Bm, w)
    out.backward(go)
t_fb = timed(bwd)
print(f"blend fwd {t_f:.1f} us ({M * K * 4 / t_f / 1e6:.2f} TB/s of Bmat, {2 * M * K * F / t_f / 1e6:.1f} TFLOP/s); fwd+bwd_w {t_fb:.1f} us")
# the backward-to-weights kernel alone (fpcdr_blend_bwd_w), straight through the C ABI
import ctypes
from fpc_diffrend_amd import _lib
gw = torch.zeros(F, K, device='cuda')
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr())
t_bw = timed(lambda: _lib.call("fpcdr_blend_bwd_w", P(Bm), P(go), P(gw), M, K, F, st))
ref = (go.double() @ Bm.double())
gw.zero_(); _lib.call("fpcdr_blend_bwd_w", P(Bm), P(go), P(gw), M, K, F, st)
print(f"blend bwd_w {t_bw:.1f} us ({M * K * 4 / t_bw / 1e6:.2f} TB/s of Bmat); rel err {float((gw.double() - ref).norm() / ref.norm()):.1e}")
