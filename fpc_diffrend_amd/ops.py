"""
nvdiffrast-compatible operator API over the gfx950 HIP kernels (C ABI: include/fpcdr.h).

The reference's fit loop calls exactly these names on `import nvdiffrast.torch as dr`
(reference src/torch/fit.py:13):

    dr.RasterizeGLContext(device='cuda')                                   fit.py:484
    dr.rasterize(glctx, pos_clip, pos_idx, resolution=(H, W))              fit.py:151
    dr.interpolate(uv[None], rast, uv_idx[, rast_db=..., diff_attrs='all']) fit.py:154, 157
    dr.texture(tex[None], texc[, texd], filter_mode=..., max_mip_level=..) fit.py:155, 158
    dr.antialias(colour, rast, pos_clip, pos_idx)                          fit.py:160

so `import fpc_diffrend_amd.ops as dr` lets the reference's render() (fit.py:134-162) run
unchanged.  Signatures, argument meaning, defaults and error behaviour follow nvdiffrast's
documented API; arguments the reference never passes are accepted with upstream defaults.
Instanced mode (pos [B,V,4]) and range mode (pos [V,4] + ranges [B,2]) are implemented.

PyTorch is plumbing here: it owns the HBM buffers (including kernel scratch, so lifetimes follow
autograd), supplies the stream, and runs autograd bookkeeping.  All pixel work happens in
libfpcdr.so; there is no CPU or eager-torch fallback.
"""
import ctypes
import weakref
from collections import OrderedDict

import torch

from . import _lib

__all__ = ['RasterizeGLContext', 'RasterizeCudaContext', 'RasterizeHipContext', 'rasterize', 'interpolate', 'texture',
           'texture_construct_mip', 'antialias', 'antialias_construct_topology_hash', 'render_textured', 'pixel_objective']


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _check_tensor(name, t, dtype, dims=None):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if not t.is_cuda:
        raise ValueError(f"{name} must be a GPU tensor (got device {t.device}); the raster ops have no CPU path")
    if t.dtype != dtype:
        raise ValueError(f"{name} must have dtype {dtype} (got {t.dtype})")
    if dims is not None and t.dim() != dims:
        raise ValueError(f"{name} must have {dims} dimensions (got shape {tuple(t.shape)})")


# ----------------------------------------------------------------------------------------------
# region hints (include/fpcdr.h): which 32 x 32-pixel bins of an image tensor are known to be empty
# ----------------------------------------------------------------------------------------------
# rasterize() leaves a byte map of the bins no triangle touches.  It rides with the tensors of the operator chain of the
# reference's render() (fit.py:151-160) -- rast ('rast': zero there), the interpolated texture coordinates ('zero') and the
# sampled colour ('const': one known value there) -- in this registry, keyed by the tensor OBJECT: a consumer that is handed
# the very tensor an operator returned, unmodified (same object, same version counter), does not read it in empty bins.
# Anything else (a clone, a slice, an in-place edit, a tensor from elsewhere) simply finds no hint and takes the dense path.
# Results are identical either way; set `region_hints = False` to switch the mechanism off.
region_hints = True
_hints = {}


def _tag(t, hint, kind, const=None):
    """Register the hint of a tensor THIS library has just produced (a fresh allocation that owns its storage)."""
    if len(_hints) > 256:
        for k in [k for k, e in _hints.items() if e[0]() is None]:
            del _hints[k]
    st = t.untyped_storage()
    _hints[t.data_ptr()] = (weakref.ref(t), t._version, tuple(t.shape), hint, kind, const, st.data_ptr(), st.nbytes())


def _hint_of(t, kind):
    """The hint of `t`, or None.  A hint is honoured only for the very tensor object an operator of this library returned, owning the
    storage the library allocated for it (not a view, not re-pointed with set_() / .data = ...), with an unchanged version counter --
    every in-place torch operation bumps it.  What no host-side check can see is a write that bypasses autograd's bookkeeping: a
    foreign kernel, or `t.data.copy_(...)`; such a caller must call ops.drop_hints(t) (or work under ops.no_region_hints())."""
    if not region_hints:
        return None
    e = _hints.get(t.data_ptr())
    if e is None:
        return None
    ref, version, shape, hint, k, const, st_ptr, st_bytes = e
    if ref() is not t or t._version != version or tuple(t.shape) != shape or k != kind:
        return None
    st = t.untyped_storage()
    if t._base is not None or st.data_ptr() != st_ptr or st.nbytes() != st_bytes or t.storage_offset() != 0 or not t.is_contiguous():
        return None
    return hint, const


def drop_hints(*tensors):
    """Forget the region hints of these tensors (call it after writing to one of them behind torch's back)."""
    for t in tensors:
        _hints.pop(t.data_ptr(), None)


class no_region_hints:
    """Context manager: the operators inside take the dense path (module switch `region_hints` restored on exit)."""

    def __enter__(self):
        global region_hints
        self._old, region_hints = region_hints, False

    def __exit__(self, *exc):
        global region_hints
        region_hints = self._old


def _hint_bytes(B, H, W):
    return 2 * B * ((H + _lib.OCC_BIN - 1) // _lib.OCC_BIN) * ((W + _lib.OCC_BIN - 1) // _lib.OCC_BIN)     # FPCDR_HINT_BYTES


# ----------------------------------------------------------------------------------------------
# contexts
# ----------------------------------------------------------------------------------------------

class RasterizeHipContext:
    """Rasteriser context.  Stateless on the device side (the library keeps no state between calls);
    holds the device and whether rast_db is produced.  `RasterizeGLContext` / `RasterizeCudaContext`
    are aliases so reference code constructing either keeps working (reference fit.py:484)."""

    def __init__(self, output_db=True, mode='automatic', device=None):
        assert output_db is True or output_db is False
        assert mode in ('automatic', 'manual')
        self.output_db = output_db
        self.mode = mode
        if device is None:
            self.device = torch.device('cuda', torch.cuda.current_device()) if torch.cuda.is_available() else None
        else:
            self.device = torch.device(device)
            if self.device.type == 'cuda' and self.device.index is None and torch.cuda.is_available():
                self.device = torch.device('cuda', torch.cuda.current_device())
        _lib.load()  # fail loudly here if the HIP extension is missing

    # nvdiffrast GL-context API surface (no-ops: there is no GL context to bind)
    def set_context(self):
        pass

    def release_context(self):
        pass


RasterizeGLContext = RasterizeHipContext
RasterizeCudaContext = RasterizeHipContext


# ----------------------------------------------------------------------------------------------
# rasterize
# ----------------------------------------------------------------------------------------------

class _rasterize_func(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pos, tri, H, W, output_db, grad_db, hint, ranges):
        lib = _lib.load()
        B, V, _ = pos.shape
        T = tri.shape[0]
        dev = pos.device
        rast = torch.empty(B, H, W, 4, dtype=torch.float32, device=dev)
        rast_db = torch.empty(B, H, W, 4, dtype=torch.float32, device=dev) if output_db else None
        scratch = torch.empty(lib.fpcdr_rasterize_scratch_bytes(B, T), dtype=torch.uint8, device=dev)
        p = _lib.RasterizeFwd(pos=_ptr(pos), tri=_ptr(tri), B=B, V=V, T=T, H=H, W=W, scratch=_ptr(scratch),
                              rast=_ptr(rast), rast_db=_ptr(rast_db), hint=_ptr(hint), ranges=_ptr(ranges))
        _lib.call("fpcdr_rasterize_fwd", ctypes.byref(p), _stream())
        ctx.save_for_backward(pos, tri, rast)
        ctx.hint = hint
        ctx.grad_db = bool(grad_db and output_db)
        if rast_db is None:
            rast_db = torch.zeros(B, H, W, 0, dtype=torch.float32, device=dev)
        return rast, rast_db

    @staticmethod
    def backward(ctx, dy, ddb):
        lib = _lib.load()
        pos, tri, rast = ctx.saved_tensors
        B, V, _ = pos.shape
        _, H, W, _ = rast.shape
        g_pos = torch.zeros_like(pos)
        dy = dy.contiguous()
        ddb = ddb.contiguous() if (ctx.grad_db and ddb is not None and ddb.numel() > 0) else None
        p = _lib.RasterizeBwd(pos=_ptr(pos), tri=_ptr(tri), rast=_ptr(rast), dy=_ptr(dy), ddb=_ptr(ddb), B=B, V=V,
                              T=tri.shape[0], H=H, W=W, grad_pos=_ptr(g_pos), hint=_ptr(ctx.hint))
        _lib.call("fpcdr_rasterize_bwd", ctypes.byref(p), _stream())
        return g_pos, None, None, None, None, None, None, None


def rasterize(glctx, pos, tri, resolution, ranges=None, grad_db=True):
    """Rasterize triangles.  pos [B,V,4] clip space f32, tri [T,3] i32, resolution (H, W).

    Returns (rast [B,H,W,4] = (u, v, z/w, triangle_id + 1), rast_db [B,H,W,4] = (du/dX, du/dY, dv/dX, dv/dY)).
    """
    assert isinstance(glctx, RasterizeHipContext), "glctx must be a Rasterize*Context"
    ranges_dev = None
    if isinstance(pos, torch.Tensor) and pos.dim() == 2:
        # range mode (nvdiffrast): one shared vertex array pos [V,4]; image b renders triangles ranges[b] = (first, count).
        # The vertex array is broadcast over the minibatch (its gradient is the sum over the images); the kernel only needs
        # to know which slice of `tri` each image draws.  Triangle ids in rast stay indices into `tri`.
        if ranges is None:
            raise ValueError("range mode (pos [V,4]) needs ranges [B,2]")
        ranges = torch.as_tensor(ranges)
        if ranges.dim() != 2 or ranges.shape[1] != 2 or ranges.shape[0] < 1 or ranges.dtype != torch.int32:
            raise ValueError("ranges must be an int32 tensor of shape [minibatch, 2]")
        r = ranges.cpu()
        if int(r.min()) < 0 or int((r[:, 0] + r[:, 1]).max()) > tri.shape[0]:
            raise ValueError("ranges reach outside the triangle tensor")
        ranges_dev = ranges.to(pos.device).contiguous()
        pos = pos[None].expand(ranges.shape[0], -1, -1)
    elif ranges is not None:
        raise ValueError("ranges is for range mode only (pos [V,4]); instanced mode takes pos [B,V,4]")
    assert grad_db is True or grad_db is False
    resolution = tuple(int(r) for r in resolution)
    assert len(resolution) == 2 and resolution[0] > 0 and resolution[1] > 0, "resolution must be (height, width)"
    _check_tensor('pos', pos, torch.float32, 3)
    _check_tensor('tri', tri, torch.int32, 2)
    if pos.shape[2] != 4 or pos.shape[0] < 1 or pos.shape[1] < 1:
        raise ValueError(f"pos must have shape [>0, >0, 4] (got {tuple(pos.shape)})")
    if tri.shape[1] != 3 or tri.shape[0] < 1:
        raise ValueError(f"tri must have shape [>0, 3] (got {tuple(tri.shape)})")
    if glctx.device is not None and pos.device != glctx.device:
        raise ValueError(f"pos is on {pos.device} but the context was created for {glctx.device}")
    hint = torch.empty(_hint_bytes(pos.shape[0], *resolution), dtype=torch.uint8, device=pos.device) if region_hints else None
    rast, rast_db = _rasterize_func.apply(pos.contiguous(), tri.contiguous(), resolution[0], resolution[1], glctx.output_db,
                                          grad_db, hint, ranges_dev)
    if hint is not None:
        _tag(rast, hint, 'rast')
    return rast, rast_db


# ----------------------------------------------------------------------------------------------
# fused render (extension; not part of the nvdiffrast surface)
# ----------------------------------------------------------------------------------------------

class _render_textured_func(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pos, tri, uv, uv_tri, tex, H, W, boundary):
        lib = _lib.load_twocall()      # (include/fpcdr_twocall.h: fpcdr_render_fwd / _bwd live in the second library)
        B, V, _ = pos.shape
        T = tri.shape[0]
        Ht, Wt, C = tex.shape
        dev = pos.device
        rast = torch.empty(B, H, W, 4, dtype=torch.float32, device=dev)
        color = torch.empty(B, H, W, C, dtype=torch.float32, device=dev)
        scratch = torch.empty(lib.fpcdr_rasterize_scratch_bytes(B, T), dtype=torch.uint8, device=dev)
        tri_uv = uv[uv_tri.long()].contiguous()      # [T,3,2], static per mesh
        p = _lib.RenderFwd(pos=_ptr(pos), tri=_ptr(tri), B=B, V=V, T=T, H=H, W=W, scratch=_ptr(scratch), uv=_ptr(uv),
                           uv_tri=_ptr(uv_tri), Vt=uv.shape[0], tex=_ptr(tex), Ht=Ht, Wt=Wt, C=C, boundary_mode=boundary,
                           rast=_ptr(rast), color=_ptr(color), tri_uv=_ptr(tri_uv))
        _lib.call("fpcdr_render_fwd", ctypes.byref(p), _stream())
        ctx.save_for_backward(pos, tri, uv, uv_tri, tex, rast)
        ctx.boundary = boundary
        ctx.mark_non_differentiable(rast)
        return color, rast

    @staticmethod
    def backward(ctx, dy, _drast):
        pos, tri, uv, uv_tri, tex, rast = ctx.saved_tensors
        B, V, _ = pos.shape
        _, H, W, _ = rast.shape
        Ht, Wt, C = tex.shape
        g_pos = torch.zeros_like(pos) if ctx.needs_input_grad[0] else None
        g_tex = torch.zeros_like(tex) if ctx.needs_input_grad[4] else None
        if g_pos is None and g_tex is None:
            return (None,) * 8
        dy = dy.contiguous()
        p = _lib.RenderBwd(pos=_ptr(pos), tri=_ptr(tri), uv=_ptr(uv), uv_tri=_ptr(uv_tri), tex=_ptr(tex), rast=_ptr(rast),
                           dy=_ptr(dy), B=B, V=V, T=tri.shape[0], H=H, W=W, Vt=uv.shape[0], Ht=Ht, Wt=Wt, C=C,
                           boundary_mode=ctx.boundary, grad_pos=_ptr(g_pos), grad_tex=_ptr(g_tex), tri_uv=None)
        _lib.call("fpcdr_render_bwd", ctypes.byref(p), _stream())
        return g_pos, None, None, None, g_tex, None, None, None


def render_textured(glctx, pos, tri, uv, uv_tri, tex, resolution, boundary_mode='wrap'):
    """rasterize -> interpolate(uv) -> texture('linear') in one pass (the non-mip branch of the reference's render(),
    fit.py:151,157,158).  pos [B,V,4], tri [T,3], uv [Vt,2], uv_tri [T,3], tex [Ht,Wt,C].  Returns (colour [B,H,W,C],
    rast [B,H,W,4]); values equal the three separate calls, the texture-coordinate image never reaches HBM, and the
    backward pass scatters straight into grad_tex and grad_pos.  rast carries no gradient here (use `rasterize` for that)."""
    assert isinstance(glctx, RasterizeHipContext)
    _check_tensor('pos', pos, torch.float32, 3)
    _check_tensor('tri', tri, torch.int32, 2)
    _check_tensor('uv', uv, torch.float32, 2)
    _check_tensor('uv_tri', uv_tri, torch.int32, 2)
    _check_tensor('tex', tex, torch.float32, 3)
    if uv.shape[1] != 2 or uv_tri.shape != tri.shape or pos.shape[2] != 4 or tri.shape[1] != 3:
        raise ValueError("shapes: pos [B,V,4], tri [T,3], uv [Vt,2], uv_tri [T,3], tex [Ht,Wt,C]")
    if boundary_mode not in _lib.BOUNDARY:
        raise ValueError(f"unknown boundary_mode '{boundary_mode}'")
    H, W = int(resolution[0]), int(resolution[1])
    return _render_textured_func.apply(pos.contiguous(), tri.contiguous(), uv.contiguous(), uv_tri.contiguous(),
                                       tex.contiguous(), H, W, _lib.BOUNDARY[boundary_mode])


class _ListHints:
    """Launch-size hints for the list kernels of the sparse objective (include/fpcdr.h: cap_bins / cap_fix / cap_bwd).
    Every forward call leaves the number of bins each of its three list kernels visited in the occupancy buffer; they are
    copied to pinned host memory asynchronously and the NEXT call on a batch of the same shape sizes its launches from them
    (+ 12 % margin) -- never waiting: a copy that has not landed yet simply means "no hint".  Results do not depend on the
    hints (entries beyond one are swept up on the device)."""

    def __init__(self):
        self.host = torch.zeros(4, dtype=torch.int32).pin_memory()
        self.event = None
        self.caps = (0, 0, 0)       # bins, fix, bwd

    def poll(self):
        if self.event is not None and self.event.query():
            n_bwd, _, n_bins, n_fix = (int(v) for v in self.host.tolist())
            self.caps = tuple(n + max(256, n // 8) if n > 0 else 0 for n in (n_bins, n_fix, n_bwd))
            self.event = None
        return self.caps

    def update(self, counts_dev):
        if self.event is not None or torch.cuda.is_current_stream_capturing():
            return                  # a copy is still in flight / no host read-back from inside a HIP graph
        self.host.copy_(counts_dev, non_blocking=True)
        self.event = torch.cuda.Event()
        self.event.record()


class _MappedHints:
    """The same for the one-pass objective, without a copy in the stream: the call's last kernel writes the counters straight into
    pinned host memory (fpcdr_objective_params.counts_out) as a sequence lock -- the call's sequence number in front of AND behind them;
    poll() uses whatever has landed (a device-to-host copy per step was a blit kernel between two barriers: ~20 us of the step's serial tail)."""

    def __init__(self):
        self.host = torch.zeros(8, dtype=torch.int32)
        if torch.cuda.is_available():      # (device-writable host memory; without a GPU only the bookkeeping below can be exercised)
            self.host = self.host.pin_memory()
        self.seq = 0
        self.seen = 0
        self.caps = (0, 0, 0)       # live bins, occupied bins, bins with a deferred pixel
        self.frozen = False         # tests: keep `caps` as set
        self.sil_bins = -1          # bins that took a record slot in the last call seen (compact records; -1: none seen yet)
        self.slots = 0              # record slots the next call gets (tests may pin it with `frozen`)
        self.overflowed = None      # sequence number of the newest call seen at or before which a call ran out of record slots
        self.overflow_count = 0     # the device's cumulative count of such calls, as last read (host[5]: only clear_hints resets it)
        self.skipped_calls = 0      # ... of them, calls whose caller took the device-side flag (skip_out): handled, not raised

    def poll(self):
        if self.frozen:
            self._poll_overflow(int(self.host[4]))
            return self.caps
        # include/fpcdr.h, counts_out: the device writes [6] = seq, fences, the counters, fences, [4] = seq.  Read the other way round --
        # [4], the counters, [6] -- the counters are ONE call's iff both numbers agree: a later call that has started to write has
        # already changed [6], one that has not finished has not yet changed [4]
        end = int(self.host[4])
        n_def, n_sil, n_bins, n_occ = (int(v) for v in self.host[:4].tolist())
        slots_valid = int(self.host[7])
        begin = int(self.host[6])
        if begin == end and end != self.seen:
            self.seen = end
            self.caps = tuple(n + max(256, n // 8) if n > 0 else 0 for n in (n_bins, n_occ, n_def))
            if slots_valid:      # (a dense call takes no record slots: its zero says nothing about the demand)
                self.sil_bins = n_sil
                self.slots = n_sil + max(RECORD_SLOT_MARGIN, n_sil // 2)
        self._poll_overflow(end)
        return self.caps

    def _poll_overflow(self, seq):
        # cumulative on the device: an overflow in call N is still there when call N + 1 has finished before this poll
        n_over = int(self.host[5])
        if n_over != self.overflow_count:
            self.overflow_count = n_over
            self.overflowed = seq

    def next_seq(self):
        self.seq = self.seq % 0x7ffffff0 + 1
        return self.seq


_list_hints = {}
# Compact records (fpcdr_objective_params.rec_slots): a call gets 1.5 x the slots the last call on the batch shape used, at least this
# many more.  The slot demand is the number of occupied bins that show a silhouette triangle: it moves by a few per cent from one minibatch
# of a take to the next.  A call that runs out regardless says so itself -- NaN value, skip_out = 1 (include/fpcdr.h ABI v11): Fitter skips
# that update on the device --; a caller that did not hand over skip_out gets a RuntimeError at its next call on the shape.
RECORD_SLOT_MARGIN = 1024
COMPACT_RECORDS = True
SMALL_BATCH_BINS = 16384      # (eight full-HD images) batches up to this many 32 x 32 bins run their list kernels unhinted: see _pixel_objective_onepass


def _hints_for(key, cls):
    """The hint record of a batch shape, made on first use -- never under HIP-graph capture (its pinned allocation would be a
    hipHostMalloc inside the capture): a captured call simply runs unhinted then."""
    h = _list_hints.get(key)
    if h is None:
        if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
            return None
        h = _list_hints[key] = cls()
    return h


def clear_hints():
    """Forget every launch hint.  The objective's last kernel writes its counters into the records' pinned host memory through a raw
    pointer the allocator cannot see, so the device is drained first: dropping a record while a call is queued would return its block
    to the pinned cache with a write still in flight."""
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    _list_hints.clear()


class _pixel_objective_func(torch.autograd.Function):
    """mean over n_total of (ref - 255 * where(covered, antialias(render(pos, tex)), bg))^2 as three kernels."""

    @staticmethod
    def forward(ctx, pos, tex, tri, adj, uv, uv_tri, ref, H, W, n_total, bg, boundary, sparse, ref_bg_sumsq, use_hints,
                queued_backward, mip_levels=None, grad_enabled=True):
        lib = _lib.load_twocall()      # (the two-call form: include/fpcdr_twocall.h, libfpcdr_twocall.so)
        B, V, _ = pos.shape
        T = tri.shape[0]
        Ht, Wt, C = tex.shape
        dev = pos.device
        rast = torch.empty(B, H, W, 4, dtype=torch.float32, device=dev)
        color = torch.empty(B, H, W, C, dtype=torch.float32, device=dev)
        scratch = torch.empty(lib.fpcdr_rasterize_scratch_bytes(B, T), dtype=torch.uint8, device=dev)
        # sparse: 32x32-pixel bins that no triangle's bounding box touches are neither written nor read by the three
        # kernels; occ is the map of the others, ecol the colour of an empty pixel
        occ = torch.empty(lib.fpcdr_occ_bytes(B, H, W), dtype=torch.uint8, device=dev) if sparse else None
        ecol = torch.empty(4, dtype=torch.float32, device=dev) if sparse else None
        tri_uv = _cached_tri_uv(uv, uv_tri)          # [T,3,2], static per mesh: saves a dependent load per pixel
        p = _lib.RenderFwd(pos=_ptr(pos), tri=_ptr(tri), B=B, V=V, T=T, H=H, W=W, scratch=_ptr(scratch), uv=_ptr(uv),
                           uv_tri=_ptr(uv_tri), Vt=uv.shape[0], tex=_ptr(tex), Ht=Ht, Wt=Wt, C=C, boundary_mode=boundary,
                           rast=_ptr(rast), color=_ptr(color), tri_uv=_ptr(tri_uv), occ=_ptr(occ), empty_color=_ptr(ecol))
        # mip_levels = n: the reference's enable_mip branch inside the same kernels (the chain is built here, box filter as texture())
        chain = _build_mips(tex[None], mip_levels)[1:] if mip_levels is not None else []
        if mip_levels is not None:
            p.mip, p.n_levels = 1, len(chain)
            for l, t in enumerate(chain):
                p.tex_mip[l] = _ptr(t)
        g_aa = torch.empty_like(color)
        sil = torch.empty(B, T, dtype=torch.uint8, device=dev)
        nflag = lib.fpcdr_antialias_flags_bytes(B, H, W) // 8
        flags = torch.empty(nflag, dtype=torch.int64, device=dev)     # (sparse: zeroed by fpcdr_render_loss_fwd itself, inside its first kernel)
        acc = torch.zeros(_lib.LOSS_SLOTS, dtype=torch.float64, device=dev)
        q = _lib.AaLossFwd(color=_ptr(color), rast=_ptr(rast), pos=_ptr(pos), tri=_ptr(tri), adj=_ptr(adj), ref=_ptr(ref), B=B,
                           H=H, W=W, C=C, V=V, T=T, bg=bg, color_scale=255.0, grad_scale=1.0 / n_total, sil=_ptr(sil),
                           flags=_ptr(flags), grad_aa=_ptr(g_aa), occ=_ptr(occ), empty_color=_ptr(ecol), loss_sum=_ptr(acc))
        ctx.cap_bwd = 0
        ctx.hint_update = None
        if sparse:
            # one call: the rasteriser settles every pixel antialiasing cannot touch, a second kernel the candidates it marks
            cmask = torch.empty(lib.fpcdr_cmask_bytes(B, H, W), dtype=torch.uint8, device=dev)
            hints = _hints_for((dev.index, B, V, T, H, W), _ListHints) if use_hints else None
            if hints is not None:
                q.cap_bins, q.cap_fix, ctx.cap_bwd = hints.poll()
            _lib.call("fpcdr_render_loss_fwd", ctypes.byref(p), ctypes.byref(q), _ptr(cmask), _stream())
            if hints is not None:
                # (the counts are read back after the BACKWARD call has been enqueued: the copy would otherwise sit between the
                # two large kernels on the stream)
                nb = B * ((H + _lib.OCC_BIN - 1) // _lib.OCC_BIN) * ((W + _lib.OCC_BIN - 1) // _lib.OCC_BIN)
                off = (4 * nb + 3) // 4 * 4                       # FPCDR_OCC_COUNTS_OFFSET
                ctx.hint_update = (hints, occ[off:off + 16].view(torch.int32))
        else:
            _lib.call("fpcdr_render_fwd", ctypes.byref(p), _stream())
            _lib.call("fpcdr_aa_loss_fwd", ctypes.byref(q), _stream())
        del scratch
        ctx.save_for_backward(pos, tex, tri, uv, uv_tri, rast, color, g_aa, sil, flags, occ, ecol, tri_uv, *chain)
        ctx.mip = mip_levels is not None
        ctx.boundary = boundary
        # the one-call sparse forward leaves the list of bins the backward visits in `occ`; measured at cfg3 the backward gains
        # nothing from it (2.69 vs 2.64 ms: its dead workgroups' dispatch hides behind the live ones' work), so the grid form is
        # the default and the list form stays selectable
        ctx.queued = 1 if (sparse and queued_backward) else 0
        # (sparse => the one-call forward, which also leaves the per-bin summary of the antialias flags in `occ`)
        bg_sum = None
        if sparse:
            # the kernel summed only the difference to an all-background image; the rest depends on ref alone
            bg_sum = (reference_background_sumsq(ref, bg).sum() if ref_bg_sumsq is None
                      else torch.as_tensor(ref_bg_sumsq, device=dev)).to(torch.float64).contiguous()
        out = torch.empty((), dtype=torch.float32, device=dev)
        _lib.call("fpcdr_objective_value", _ptr(acc), _lib.LOSS_SLOTS, _ptr(bg_sum), float(C), float(n_total), _ptr(out), _stream())
        # forward only (no input wants a gradient, or the caller runs under no_grad -- needs_input_grad stays True there): read the
        # counts back now; otherwise after the backward call has been enqueued
        if not (grad_enabled and (ctx.needs_input_grad[0] or ctx.needs_input_grad[1])) and ctx.hint_update is not None:
            ctx.hint_update[0].update(ctx.hint_update[1])
            ctx.hint_update = None
        return out

    @staticmethod
    def backward(ctx, g):
        pos, tex, tri, uv, uv_tri, rast, color, g_aa, sil, flags, occ, ecol, tri_uv = ctx.saved_tensors[:13]
        chain = ctx.saved_tensors[13:]
        B, V, _ = pos.shape
        _, H, W, _ = rast.shape
        Ht, Wt, C = tex.shape
        g_pos = torch.zeros_like(pos)
        g_tex = torch.zeros_like(tex) if ctx.needs_input_grad[1] else None
        g_chain = [torch.zeros_like(t) for t in chain] if g_tex is not None else []
        g = g.to(torch.float32).contiguous()
        p = _lib.RenderAaBwd(pos=_ptr(pos), tri=_ptr(tri), uv=_ptr(uv), uv_tri=_ptr(uv_tri), tex=_ptr(tex), rast=_ptr(rast),
                             color=_ptr(color), grad_aa=_ptr(g_aa), sil=_ptr(sil), flags=_ptr(flags), occ=_ptr(occ), empty_color=_ptr(ecol), B=B, V=V,
                             T=tri.shape[0],
                             H=H, W=W, Vt=uv.shape[0], Ht=Ht, Wt=Wt, C=C, boundary_mode=ctx.boundary, grad_pos=_ptr(g_pos),
                             grad_tex=_ptr(g_tex), tri_uv=_ptr(tri_uv), upstream=_ptr(g),    # g: applied inside the kernel
                             queued=ctx.queued, cap_bwd=ctx.cap_bwd, binflags=1 if occ is not None else 0)
        if ctx.mip:
            p.mip, p.n_levels = 1, len(chain)
            for l, t in enumerate(chain):
                p.tex_mip[l] = _ptr(t)
                if g_chain:
                    p.grad_tex_mip[l] = _ptr(g_chain[l])
        _lib.call("fpcdr_render_aa_bwd", ctypes.byref(p), _stream())
        if g_chain:      # fold the levels' gradients into the texture's (the box filter's backward, coarse to fine)
            g_all = [g_tex[None]] + g_chain
            for l in range(len(chain), 0, -1):
                _, h, w, _ = g_all[l - 1].shape
                _lib.call("fpcdr_mip_downsample_bwd", _ptr(g_all[l]), _ptr(g_all[l - 1]), 1, h, w, C, _stream())
        if ctx.hint_update is not None:
            ctx.hint_update[0].update(ctx.hint_update[1])
            ctx.hint_update = None
        if not ctx.needs_input_grad[0]:
            g_pos = None
        return (g_pos, g_tex) + (None,) * 16


_side_streams = {}


# The silhouette bits of the one-pass objective on a second stream, beside the rasteriser's set-up kernel (fpcdr_objective_params.sil_ready /
# sil_event)?  Off: at cfg3 the ~60 us the overlap hides are what the set-up kernel -- bound by memory latency -- loses to the company plus the
# ~10 us a cross-stream wait costs the main stream on this runtime, event long complete or not (2.78 ms per step without, 2.80 with;
# profiles/r04_stream_overlap.txt).
OVERLAP_SIL = False


def _side_stream(dev):
    """One helper stream per device (the silhouette bits of the one-pass objective run on it beside the rasteriser's set-up)."""
    st = _side_streams.get(dev.index)
    if st is None:
        st = _side_streams[dev.index] = torch.cuda.Stream(device=dev)
    return st


class _pixel_objective_onepass(torch.autograd.Function):
    """The pixel objective with VALUE AND GRADIENT from one call (include/fpcdr.h, fpcdr_objective_fwd): the objective is a scalar, so
    its gradient is d(objective)/d(input) times the one upstream number -- it is accumulated by the kernel that shades a pixel, while
    barycentrics, taps, texels and vertices are in registers, and backward() only multiplies by what autograd hands over."""

    @staticmethod
    def forward(ctx, pos, tex, tri, adj, uv, uv_tri, ref, H, W, n_total, bg, boundary, ref_bg_sumsq, use_hints, want_grad, unit_upstream,
                flags_out=None, mip_levels=None, zero_extra=None, overlap_sil=None, bin_lists=True, idp_out=None, record_slots=None,
                skip_out=None):
        lib = _lib.load()
        B, V, _ = pos.shape
        T = tri.shape[0]
        Ht, Wt, C = tex.shape
        dev = pos.device
        want_pos = bool(want_grad and ctx.needs_input_grad[0])
        want_tex = bool(want_grad and ctx.needs_input_grad[1])
        u8 = lambda n: torch.empty(n, dtype=torch.uint8, device=dev)
        scratch = u8(lib.fpcdr_rasterize_scratch_bytes(B, T))
        sil = u8(B * T)
        if idp_out is not None:      # (tests / diagnostics: the caller keeps the id planes)
            if idp_out.numel() * idp_out.element_size() != lib.fpcdr_idplane_bytes(B, H, W) or not idp_out.is_contiguous() or idp_out.device != dev:
                raise ValueError("id_plane_out must be a contiguous device tensor of fpcdr_idplane_bytes(B, H, W) bytes")
        idp = idp_out if idp_out is not None else u8(lib.fpcdr_idplane_bytes(B, H, W))
        binlist = u8(lib.fpcdr_binlist_bytes(B, H, W)) if bin_lists else None      # per-bin triangle lists (set-up kernel -> rasteriser)
        occ, cmask = u8(lib.fpcdr_occ_bytes(B, H, W)), u8(lib.fpcdr_objective_cmask_bytes(B, H, W))
        ecol = torch.empty(4, dtype=torch.float32, device=dev)
        # (zero_outputs: the call's first kernel zero-fills its accumulators)
        acc = torch.empty(_lib.LOSS_SLOTS, dtype=torch.float64, device=dev)
        g_pos = torch.empty_like(pos) if want_pos else None
        g_tex = torch.empty_like(tex) if want_tex else None
        tri_uv = _cached_tri_uv(uv, uv_tri)
        p = _lib.Objective(pos=_ptr(pos), tri=_ptr(tri), adj=_ptr(adj), B=B, V=V, T=T, H=H, W=W, scratch=_ptr(scratch), uv=_ptr(uv),
                           uv_tri=_ptr(uv_tri), Vt=uv.shape[0], tri_uv=_ptr(tri_uv), tex=_ptr(tex), Ht=Ht, Wt=Wt, C=C,
                           boundary_mode=boundary, ref=_ptr(ref), bg=bg, color_scale=255.0, grad_scale=1.0 / n_total, sil=_ptr(sil),
                           idp=_ptr(idp), occ=_ptr(occ), cmask=_ptr(cmask),
                           empty_color=_ptr(ecol), loss_sum=_ptr(acc), grad_pos=_ptr(g_pos), grad_tex=_ptr(g_tex), flags=_ptr(flags_out),
                           binlist=_ptr(binlist), zero_outputs=1, skip_out=_ptr(skip_out))
        # mip_levels = n: the reference's enable_mip branch inside the same kernels (the chain is built here, box filter as texture())
        chain = _build_mips(tex[None], mip_levels)[1:] if mip_levels is not None else []
        g_chain = [torch.empty_like(t) for t in chain] if want_tex else []
        if mip_levels is not None:
            p.mip, p.n_levels = 1, len(chain)
            for l, t in enumerate(chain):
                p.tex_mip[l] = _ptr(t)
                if g_chain:
                    p.grad_tex_mip[l] = _ptr(g_chain[l])
        # the silhouette bits need the positions only: on a second stream they run beside the rasteriser's set-up kernel (not inside a
        # graph capture, where the fork would become part of the caller's graph topology).  Forked HERE, behind the zero-fills of the
        # gradient buffers above: started earlier the kernel shares the memory system with them (the 69 MB fill took 68 us instead of 12)
        if overlap_sil is None:
            overlap_sil = OVERLAP_SIL
        side = _side_stream(dev) if (overlap_sil and not torch.cuda.is_current_stream_capturing()) else None
        if side is not None:
            main = torch.cuda.current_stream(dev)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                _lib.call("fpcdr_silhouette_bits", _ptr(pos), _ptr(tri), _ptr(adj), B, V, T, H, W, _ptr(sil), ctypes.c_void_p(side.cuda_stream))
                sil_event = torch.cuda.Event()
                sil_event.record(side)
            for t_ in (pos, tri, adj, sil):
                t_.record_stream(side)
        capturing = torch.cuda.is_current_stream_capturing()
        hints = _hints_for(('onepass', dev.index, B, V, T, H, W), _MappedHints) if use_hints else None
        nbins = B * ((H + _lib.OCC_BIN - 1) // _lib.OCC_BIN) * ((W + _lib.OCC_BIN - 1) // _lib.OCC_BIN)
        if hints is not None:
            p.cap_bins, p.cap_occ, p.cap_def = hints.poll()      # (live bins, occupied bins, bins with a deferred pixel)
            # a batch of few bins -- one image of the reference's run shape has 1 900 -- is launched at its full size whatever the
            # hint says: a sized launch has a strided sweep launched behind it (four per call, 4.6 us each), and the dead workgroups
            # of a full launch cost less than that
            if nbins <= SMALL_BATCH_BINS and not hints.frozen:
                p.cap_bins, p.cap_occ, p.cap_def = 0, 0, 0
            if not capturing:      # (a graph replays fixed launch sizes: nothing to report)
                p.counts_out, p.counts_seq = hints.host.data_ptr(), hints.next_seq()
        # records of the DEFERRED pixels (a pixel pair at a silhouette; a few per cent of the covered pixels -- nothing else of the image
        # ever exists in HBM).  Large batches: COMPACT, a slot of 1 024 records for every bin that shows a silhouette triangle, sized from
        # the last call on the batch shape (24 B per pixel of the batch otherwise: 14 GB at 288 full-HD images, for 0.2 % of them).  The
        # first call on a shape asks the rasteriser for the count (one extra rasterisation and a host wait, once per shape).  Small
        # batches, captured calls and calls without hints address the records by pixel.
        slots = 0
        if hints is not None and hints.overflowed is not None and not capturing:
            # an earlier call on this shape ran out of record slots: its value was NaN and its gradients incomplete.  A caller that hands
            # over skip_out had the device flag of that very call (Fitter: the update was skipped on the device) -- nothing to raise; the
            # demand the call counted (it keeps counting beyond the pool) has sized this call.  Anybody else learns it here.
            seq, hints.overflowed = hints.overflowed, None
            if skip_out is not None:
                hints.skipped_calls += 1
            else:
                hints.sil_bins = -1      # (re-count in front of the next call)
                raise RuntimeError(f"pixel_objective: a call on this batch shape (sequence number <= {seq}) ran out of record slots (the "
                                   "take's silhouette grew by more than half within one step); its value was NaN and its gradients "
                                   "incomplete -- repeat the step, or pass skip_out= and skip the update as Fitter does")
        if record_slots is not None:
            slots = int(record_slots)
        elif COMPACT_RECORDS and hints is not None and not capturing and nbins > SMALL_BATCH_BINS:
            if hints.sil_bins < 0 and not hints.frozen:
                p.count_only = 1
                _lib.call("fpcdr_objective_fwd", ctypes.byref(p), _stream())
                p.count_only = 0
                torch.cuda.current_stream(dev).synchronize()
                p.cap_bins, p.cap_occ, p.cap_def = hints.poll()
                p.counts_seq = hints.next_seq()
                if hints.sil_bins < 0:
                    raise RuntimeError("pixel_objective: the counting call reported nothing")
            slots = max(hints.slots, 1)
        if slots > 0:
            n_rec = slots * _lib.OCC_BIN * _lib.OCC_BIN
            slot_map = torch.empty(nbins, dtype=torch.int32, device=dev)
            p.rec_slots, p.slot_map = slots, _ptr(slot_map)
        else:
            n_rec = B * H * W
        rec = torch.empty(n_rec, 4, dtype=torch.float32, device=dev)
        color = torch.empty(n_rec, C, dtype=torch.float32, device=dev)
        g_aa = torch.empty(n_rec, C, dtype=torch.float32, device=dev)
        p.rec, p.color, p.grad_aa = _ptr(rec), _ptr(color), _ptr(g_aa)
        # the value from the call's last kernel: (loss slots + C * background share) / n_total
        bg_sum = (reference_background_sumsq(ref, bg).sum() if ref_bg_sumsq is None
                  else torch.as_tensor(ref_bg_sumsq, device=dev)).to(torch.float64).contiguous()
        out = torch.empty((), dtype=torch.float32, device=dev)
        p.bg_sumsq, p.bg_coeff, p.n_total, p.value_out = _ptr(bg_sum), float(C), float(n_total), _ptr(out)
        if zero_extra is not None:      # (a caller's own small accumulators, zero-filled by the call's first kernel)
            p.zero_extra, p.zero_extra_bytes = _ptr(zero_extra), zero_extra.numel() * zero_extra.element_size()
        if side is not None:      # (the call waits for the event right before its first kernel that reads the bits)
            p.sil_ready, p.sil_event = 1, ctypes.c_void_p(sil_event.cuda_event)
        _lib.call("fpcdr_objective_fwd", ctypes.byref(p), _stream())
        if g_chain:      # fold the levels' gradients into the texture's (the box filter's backward, coarse to fine)
            g_all = [g_tex[None]] + g_chain
            for l in range(len(chain), 0, -1):
                _, h, w, _ = g_all[l - 1].shape
                _lib.call("fpcdr_mip_downsample_bwd", _ptr(g_all[l]), _ptr(g_all[l - 1]), 1, h, w, C, _stream())
        ctx.save_for_backward(*(t for t in (g_pos, g_tex) if t is not None))
        ctx.have = (want_pos, want_tex)
        ctx.unit = bool(unit_upstream)
        return out

    @staticmethod
    def backward(ctx, g):
        saved = list(ctx.saved_tensors)
        g_pos = saved.pop(0) if ctx.have[0] else None
        g_tex = saved.pop(0) if ctx.have[1] else None
        if (ctx.needs_input_grad[0] and g_pos is None) or (ctx.needs_input_grad[1] and g_tex is None):
            raise RuntimeError("pixel_objective: the forward pass ran with gradients disabled")
        if not ctx.unit:      # (unit_upstream: the caller guarantees d loss / d objective = 1, as the fit loop's `pix + reg` does)
            g = g.to(torch.float32)
            g_pos = g_pos * g if g_pos is not None else None
            g_tex = g_tex * g if g_tex is not None else None
        return (g_pos if ctx.needs_input_grad[0] else None, g_tex if ctx.needs_input_grad[1] else None) + (None,) * 22


def reference_background_sumsq(ref_u8, background=45.0 / 255.0):
    """Per image sum over pixels of (ref - 255 * background)^2 (f64 [B]): the pixel loss (fit.py:579) of an image that
    shows nothing but background.  It depends on the reference images only, so a fit loop computes it once and hands it
    to pixel_objective(ref_bg_sumsq=...), which then reads reference pixels only where geometry is."""
    _check_tensor('ref_u8', ref_u8, torch.uint8, 3)
    ref_u8 = ref_u8.contiguous()
    out = torch.zeros(ref_u8.shape[0], dtype=torch.float64, device=ref_u8.device)
    _lib.call("fpcdr_ref_bg_sumsq", _ptr(ref_u8), ref_u8.shape[0], ref_u8.shape[1] * ref_u8.shape[2], float(background) * 255.0,
              _ptr(out), _stream())
    return out


def pixel_objective(glctx, pos, tri, uv, uv_tri, tex, ref_u8, resolution, n_total=None, background=45.0 / 255.0,
                    boundary_mode='wrap', sparse=True, ref_bg_sumsq=None, launch_hints=True, queued_backward=False,
                    enable_mip=False, max_mip_level=None, one_pass=True, unit_upstream=False, aa_flags_out=None, zero_extra=None,
                    id_plane_out=None, record_slots=None, skip_out=None):
    """The whole pixel term of the reference's loss (fit.py:151-161 + the first term of :579) for a minibatch,
    as three kernels:  mean((ref - 255 * where(rast.w > 0, antialias(texture(interpolate(rasterize(pos)))), bg))^2)
    over n_total elements (default: all of this call's).  pos [B,V,4], tex [Ht,Wt,C] (C in 1,3,4), ref_u8 [B,H,W] uint8.
    Differentiable w.r.t. pos and tex; equals the chain of separate operators + pixel loss.
    sparse=True: 32x32-pixel bins that no triangle's bounding box touches are skipped by all three kernels (they can only
    contribute (ref - 255 bg)^2 to the loss, which is added from ref_bg_sumsq: a scalar f64 tensor, the sum of
    reference_background_sumsq(ref_u8, background) over this call's images; computed here when None); same result.
    launch_hints: size the sparse kernels' launches from the bin counts of the previous call on the same batch shape
    (read back asynchronously, never waited for; the result does not depend on them).  queued_backward: the backward
    kernel also runs over the compact list of occupied bins instead of one workgroup per bin (same result).
    enable_mip (sparse mode): the reference's other branch (fit.py:153-155) -- interpolate with the rasteriser's
    screen-space derivatives and texture 'linear-mipmap-linear' with max_mip_level -- inside the same three kernels; equals the
    chain rasterize(output_db) -> interpolate(diff_attrs='all') -> texture(texd) -> antialias + pixel loss.
    one_pass (sparse mode, with or without mip; the default): value and gradient from ONE call -- the kernel that shades a pixel also chains
    its gradient back (fpcdr_objective_fwd); backward() multiplies by the upstream scalar, or returns the buffers as they are with
    unit_upstream=True (the caller guarantees d loss / d objective = 1).  one_pass=False: the two-call form.
    aa_flags_out (one_pass; tests / diagnostics): a zero-filled int64 tensor of fpcdr_antialias_flags_bytes(B,H,W) / 8 words that
    receives the antialias flag planes (which pixel pairs were blended).  zero_extra (one_pass): a contiguous float32 / int32 tensor of the
    caller's that the call's first kernel zero-fills along with its own buffers (a fit step's small gradient accumulators).
    id_plane_out (one_pass; tests / diagnostics): a zero-filled tensor of fpcdr_idplane_bytes(B,H,W) bytes that is used as the call's id
    planes and so keeps them -- 1024 uint32 per 32 x 32 bin, bin-major, (triangle + 1) | silhouette bits << 24 (include/fpcdr.h).
    record_slots (one_pass): the records of deferred pixels in that many slots of 1 024 (fpcdr_objective_params.rec_slots) instead of the
    default -- by pixel for batches of up to SMALL_BATCH_BINS bins, compact and sized from the last call on the shape beyond; 0 = by pixel.
    skip_out (one_pass): a one-element float32 device tensor that receives 1.0 when the call ran out of record slots -- its value is then
    NaN and its gradients are incomplete -- and 0.0 otherwise, written by the call's last kernel: hand it to GroupedAdam.skip_flag (summed
    over the ranks under data parallelism) and the update of such a step does not happen, without a host read-back.  Without it the
    NEXT call on the batch shape raises."""
    assert isinstance(glctx, RasterizeHipContext)
    _check_tensor('pos', pos, torch.float32, 3)
    _check_tensor('tri', tri, torch.int32, 2)
    _check_tensor('uv', uv, torch.float32, 2)
    _check_tensor('uv_tri', uv_tri, torch.int32, 2)
    _check_tensor('tex', tex, torch.float32, 3)
    _check_tensor('ref_u8', ref_u8, torch.uint8, 3)
    H, W = int(resolution[0]), int(resolution[1])
    if tex.shape[2] not in (1, 3, 4):
        raise ValueError("pixel_objective supports 1, 3 or 4 colour channels")
    if ref_u8.shape != (pos.shape[0], H, W):
        raise ValueError("ref_u8 must have shape [B,H,W]")
    if boundary_mode not in _lib.BOUNDARY:
        raise ValueError(f"unknown boundary_mode '{boundary_mode}'")
    tri = tri.contiguous()
    adj = _cached_topology(tri)
    n_total = n_total or pos.shape[0] * H * W * tex.shape[2]
    mip_levels = None
    if enable_mip:
        if not sparse:
            raise NotImplementedError("pixel_objective(enable_mip=True) runs in sparse mode (use the separate operators otherwise)")
        mip_levels = _num_mip_levels(tex.shape[0], tex.shape[1], max_mip_level)
    if skip_out is not None and not (torch.is_tensor(skip_out) and skip_out.dtype == torch.float32 and skip_out.device == pos.device
                                     and skip_out.numel() >= 1 and skip_out.is_contiguous()):
        raise ValueError("skip_out must be a contiguous float32 tensor of at least one element on the device of pos")
    if zero_extra is not None:
        # (the kernel zero-fills numel * itemsize bytes from data_ptr(): a view with gaps would have other storage overwritten)
        if not (zero_extra.is_contiguous() and zero_extra.dtype in (torch.float32, torch.int32) and zero_extra.device == pos.device):
            raise ValueError("zero_extra must be a contiguous float32 / int32 tensor on the device of pos")
        if not (one_pass and sparse):
            zero_extra.zero_()
    if one_pass and sparse:
        return _pixel_objective_onepass.apply(pos.contiguous(), tex.contiguous(), tri, adj, uv.contiguous(), uv_tri.contiguous(),
                                              ref_u8.contiguous(), H, W, n_total, background, _lib.BOUNDARY[boundary_mode], ref_bg_sumsq,
                                              bool(launch_hints), torch.is_grad_enabled(), bool(unit_upstream), aa_flags_out, mip_levels, zero_extra,
                                              None, True, id_plane_out, record_slots, skip_out)
    if skip_out is not None:
        raise ValueError("skip_out belongs to the one-pass form")
    if record_slots is not None:
        raise ValueError("record_slots belongs to the one-pass form")
    if id_plane_out is not None:
        raise ValueError("id_plane_out is an output of the one-pass form")
    return _pixel_objective_func.apply(pos.contiguous(), tex.contiguous(), tri, adj, uv.contiguous(), uv_tri.contiguous(),
                                       ref_u8.contiguous(), H, W, n_total, background, _lib.BOUNDARY[boundary_mode], bool(sparse), ref_bg_sumsq,
                                       bool(launch_hints), bool(queued_backward), mip_levels, torch.is_grad_enabled())


# ----------------------------------------------------------------------------------------------
# interpolate
# ----------------------------------------------------------------------------------------------

def _diff_array(diff_list):
    arr = (ctypes.c_int32 * _lib.MAX_ATTR)()
    for i, k in enumerate(diff_list):
        arr[i] = k
    return arr


class _interpolate_func(torch.autograd.Function):
    @staticmethod
    def forward(ctx, attr, rast, tri, rast_db, diff_list, hint):
        lib = _lib.load()
        B, H, W, _ = rast.shape
        Ba, Vt, A = attr.shape
        n_diff = len(diff_list)
        dev = rast.device
        out = torch.empty(B, H, W, A, dtype=torch.float32, device=dev)
        out_da = torch.empty(B, H, W, 2 * n_diff, dtype=torch.float32, device=dev)
        p = _lib.InterpolateFwd(attr=_ptr(attr), rast=_ptr(rast), tri=_ptr(tri), rast_db=_ptr(rast_db) if n_diff else None,
                                B=B, H=H, W=W, Ba=Ba, Vt=Vt, A=A, T=tri.shape[0], n_diff=n_diff,
                                diff_idx=_diff_array(diff_list), out=_ptr(out), out_da=_ptr(out_da) if n_diff else None,
                                hint=_ptr(hint))
        _lib.call("fpcdr_interpolate_fwd", ctypes.byref(p), _stream())
        ctx.save_for_backward(attr, rast, tri, rast_db if n_diff else None)
        ctx.diff_list = diff_list
        ctx.hint = hint
        return out, out_da

    @staticmethod
    def backward(ctx, dy, dda):
        lib = _lib.load()
        attr, rast, tri, rast_db = ctx.saved_tensors
        B, H, W, _ = rast.shape
        Ba, Vt, A = attr.shape
        diff_list = ctx.diff_list
        n_diff = len(diff_list)
        dev = rast.device
        need_attr = ctx.needs_input_grad[0]
        g_attr = torch.zeros_like(attr) if need_attr else None
        g_rast = torch.empty_like(rast)
        g_db = torch.empty_like(rast_db) if n_diff else None
        dy = dy.contiguous()
        if n_diff:
            dda = dda.contiguous() if dda is not None else torch.zeros(B, H, W, 2 * n_diff, dtype=torch.float32, device=dev)
        p = _lib.InterpolateBwd(attr=_ptr(attr), rast=_ptr(rast), tri=_ptr(tri), rast_db=_ptr(rast_db) if n_diff else None,
                                dy=_ptr(dy), dda=_ptr(dda) if n_diff else None, B=B, H=H, W=W, Ba=Ba, Vt=Vt, A=A,
                                T=tri.shape[0], n_diff=n_diff, diff_idx=_diff_array(diff_list), grad_attr=_ptr(g_attr),
                                grad_rast=_ptr(g_rast), grad_rast_db=_ptr(g_db), hint=_ptr(ctx.hint))
        _lib.call("fpcdr_interpolate_bwd", ctypes.byref(p), _stream())
        return g_attr, g_rast, None, g_db, None, None


def interpolate(attr, rast, tri, rast_db=None, diff_attrs=None):
    """Interpolate vertex attributes.  attr [1|B,Vt,A], rast [B,H,W,4], tri [T,3] -> (out [B,H,W,A], out_da [B,H,W,2D])."""
    _check_tensor('attr', attr, torch.float32)
    _check_tensor('rast', rast, torch.float32, 4)
    _check_tensor('tri', tri, torch.int32, 2)
    if attr.dim() == 2:     # range mode: one shared attribute array
        attr = attr[None]
    if attr.dim() != 3:
        raise ValueError(f"attr must have shape [1|B,Vt,A] (got {tuple(attr.shape)})")
    if rast.shape[3] != 4:
        raise ValueError("rast must have shape [B,H,W,4]")
    if attr.shape[0] not in (1, rast.shape[0]):
        raise ValueError(f"attr minibatch ({attr.shape[0]}) must be 1 or match rast ({rast.shape[0]})")
    if tri.shape[1] != 3:
        raise ValueError("tri must have shape [T,3]")
    A = attr.shape[2]
    if diff_attrs is None or rast_db is None:
        diff_list = []
        if diff_attrs is not None and rast_db is None:
            raise ValueError("diff_attrs given but rast_db is None")
    elif isinstance(diff_attrs, str):
        if diff_attrs != 'all':
            raise ValueError("diff_attrs must be None, 'all' or a list of attribute indices")
        diff_list = list(range(A))
    else:
        diff_list = [int(i) for i in diff_attrs]
        if any(i < 0 or i >= A for i in diff_list):
            raise ValueError("diff_attrs index out of range")
    if len(diff_list) > _lib.MAX_ATTR:
        raise ValueError(f"at most {_lib.MAX_ATTR} attributes may have pixel differentials")
    if diff_list:
        _check_tensor('rast_db', rast_db, torch.float32, 4)
        if rast_db.shape != rast.shape:
            raise ValueError("rast_db must have the same shape as rast")
        rast_db = rast_db.contiguous()
    else:
        rast_db = None
    h = _hint_of(rast, 'rast')
    out, out_da = _interpolate_func.apply(attr.contiguous(), rast.contiguous(), tri.contiguous(), rast_db, diff_list,
                                          h[0] if h else None)
    if h:
        _tag(out, h[0], 'zero')       # no triangle, no attribute: zeros in empty bins
    return out, out_da


# ----------------------------------------------------------------------------------------------
# texture
# ----------------------------------------------------------------------------------------------

def _num_mip_levels(Ht, Wt, max_mip_level):
    n = 0
    h, w = Ht, Wt
    while (max_mip_level is None or n < max_mip_level) and h % 2 == 0 and w % 2 == 0 and h >= 2 and w >= 2:
        h //= 2
        w //= 2
        n += 1
        if n >= _lib.MAX_MIP:
            break
    return n


def _build_mips(tex, n_levels):
    lib = _lib.load()
    chain = [tex]
    for _ in range(n_levels):
        src = chain[-1]
        N, h, w, C = src.shape
        dst = torch.empty(N, h // 2, w // 2, C, dtype=torch.float32, device=tex.device)
        _lib.call("fpcdr_mip_downsample", _ptr(src), _ptr(dst), N, h, w, C, _stream())
        chain.append(dst)
    return chain


class MipStack(list):
    """Levels 1..n of the box-filtered mip chain of `base`, as texture_construct_mip() returns it.  texture(base, ..., mip=stack)
    recognises the pairing (same tensor object, unmodified) and lets the levels' gradients flow back into `base`, as
    nvdiffrast's opaque mip wrapper does; any other list of tensors passed as `mip` is a CUSTOM stack."""

    def __init__(self, levels, base):
        super().__init__(levels)
        self._base = weakref.ref(base)
        self._version = base._version

    def built_from(self, tex):
        return self._base() is tex and tex._version == self._version

    def stale_for(self, tex):
        """Built from this very tensor, which has been modified in place since (an optimiser step): the levels no longer match it."""
        return self._base() is tex and tex._version != self._version


def texture_construct_mip(tex, max_mip_level=None, cube_mode=False):
    """Pre-build a mip stack for `texture(..., mip=...)`.  Returns a MipStack (a list of the level tensors 1..n)."""
    if cube_mode:
        raise NotImplementedError("cube maps are not implemented")
    _check_tensor('tex', tex, torch.float32, 4)
    n = _num_mip_levels(tex.shape[1], tex.shape[2], max_mip_level)
    with torch.no_grad():
        return MipStack(_build_mips(tex.contiguous(), n)[1:], tex)


def _ptr_array(tensors):
    arr = (ctypes.c_void_p * (_lib.MAX_MIP + 1))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr() if t is not None else None
    return arr


class _texture_func(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tex, uv, uv_da, bias, filter_mode, boundary_mode, n_levels, hint, empty_color, custom, *mips):
        lib = _lib.load()
        B, H, W, _ = uv.shape
        Bt, Ht, Wt, C = tex.shape
        chain = [tex] + list(mips) if mips else _build_mips(tex, n_levels)
        out = torch.empty(B, H, W, C, dtype=torch.float32, device=uv.device)
        p = _lib.TextureFwd(tex=_ptr_array(chain), n_levels=n_levels, uv=_ptr(uv), uv_da=_ptr(uv_da),
                            mip_level_bias=_ptr(bias), B=B, H=H, W=W, Bt=Bt, Ht=Ht, Wt=Wt, C=C,
                            filter_mode=filter_mode, boundary_mode=boundary_mode, out=_ptr(out), hint=_ptr(hint),
                            empty_color=_ptr(empty_color))
        _lib.call("fpcdr_texture_fwd", ctypes.byref(p), _stream())
        ctx.save_for_backward(tex, uv, uv_da, bias, *chain[1:])
        ctx.cfg = (filter_mode, boundary_mode, n_levels)
        ctx.hint = hint
        ctx.custom = bool(custom)     # a caller's own mip tensors: each level keeps its gradient (nvdiffrast: not propagated to tex)
        return out

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        tex, uv, uv_da, bias = ctx.saved_tensors[:4]
        chain = [tex] + list(ctx.saved_tensors[4:])
        filter_mode, boundary_mode, n_levels = ctx.cfg
        B, H, W, _ = uv.shape
        Bt, Ht, Wt, C = tex.shape
        need_tex = ctx.needs_input_grad[0]
        if ctx.custom:
            need = [need_tex] + [bool(ctx.needs_input_grad[10 + l]) for l in range(len(chain) - 1)]
        else:
            need = [need_tex] * len(chain)
        g_levels = [torch.zeros_like(t) if n else None for t, n in zip(chain, need)]
        g_uv = torch.empty_like(uv) if ctx.needs_input_grad[1] else None
        g_da = torch.empty_like(uv_da) if (uv_da is not None and ctx.needs_input_grad[2]) else None
        g_bias = torch.empty_like(bias) if (bias is not None and ctx.needs_input_grad[3]) else None
        dy = dy.contiguous()
        p = _lib.TextureBwd(tex=_ptr_array(chain), n_levels=n_levels, uv=_ptr(uv), uv_da=_ptr(uv_da),
                            mip_level_bias=_ptr(bias), dy=_ptr(dy), B=B, H=H, W=W, Bt=Bt, Ht=Ht, Wt=Wt, C=C,
                            filter_mode=filter_mode, boundary_mode=boundary_mode, grad_tex=_ptr_array(g_levels),
                            grad_uv=_ptr(g_uv), grad_uv_da=_ptr(g_da), grad_mip_level_bias=_ptr(g_bias), hint=_ptr(ctx.hint))
        _lib.call("fpcdr_texture_bwd", ctypes.byref(p), _stream())
        if ctx.custom:
            return (g_levels[0], g_uv, g_da, g_bias, None, None, None, None, None, None) + tuple(g_levels[1:])
        if need_tex:
            # collapse the mip gradients down to level 0
            for l in range(n_levels, 0, -1):
                N, h, w, _ = chain[l - 1].shape
                _lib.call("fpcdr_mip_downsample_bwd", _ptr(g_levels[l]), _ptr(g_levels[l - 1]), N, h, w, C, _stream())
        return (g_levels[0], g_uv, g_da, g_bias, None, None, None, None, None, None) + (None,) * (len(chain) - 1)


def texture(tex, uv, uv_da=None, mip_level_bias=None, mip=None, filter_mode='auto', boundary_mode='wrap',
            max_mip_level=None):
    """Texture lookup.  tex [1|B,Ht,Wt,C], uv [B,H,W,2], uv_da [B,H,W,4] -> [B,H,W,C]."""
    if filter_mode == 'auto':
        filter_mode = 'linear-mipmap-linear' if (uv_da is not None or mip_level_bias is not None) else 'linear'
    if filter_mode not in _lib.FILTER:
        raise ValueError(f"unknown filter_mode '{filter_mode}'")
    if boundary_mode == 'cube':
        raise NotImplementedError("cube maps are not implemented")
    if boundary_mode not in _lib.BOUNDARY:
        raise ValueError(f"unknown boundary_mode '{boundary_mode}'")
    _check_tensor('tex', tex, torch.float32, 4)
    _check_tensor('uv', uv, torch.float32, 4)
    if uv.shape[3] != 2:
        raise ValueError("uv must have shape [B,H,W,2]")
    if tex.shape[0] not in (1, uv.shape[0]):
        raise ValueError(f"tex minibatch ({tex.shape[0]}) must be 1 or match uv ({uv.shape[0]})")
    mipped = filter_mode in ('linear-mipmap-nearest', 'linear-mipmap-linear')
    n_levels = 0
    mips = ()
    custom = False
    if mipped:
        if uv_da is None and mip_level_bias is None:
            raise ValueError("mipmapped filter modes need uv_da and/or mip_level_bias")
        if uv_da is not None:
            _check_tensor('uv_da', uv_da, torch.float32, 4)
            if uv_da.shape != uv.shape[:3] + (4,):
                raise ValueError("uv_da must have shape [B,H,W,4]")
            uv_da = uv_da.contiguous()
        if mip_level_bias is not None:
            _check_tensor('mip_level_bias', mip_level_bias, torch.float32, 3)
            if mip_level_bias.shape != uv.shape[:3]:
                raise ValueError("mip_level_bias must have shape [B,H,W]")
            mip_level_bias = mip_level_bias.contiguous()
        if mip is not None:
            # a stack texture_construct_mip() built from this very tensor behaves like the internal chain (its gradient
            # collapses into tex); any other list of tensors is a custom stack whose levels receive their own gradients
            if isinstance(mip, MipStack) and mip.stale_for(tex):
                # (silently demoting it to a custom stack would sample stale levels AND cut their gradient off from tex, where
                # nvdiffrast's mip wrapper keeps propagating it)
                raise RuntimeError("texture(): `mip` was built by texture_construct_mip() from this tensor, which has been modified in "
                                   "place since (e.g. an optimiser step): rebuild the stack, or pass mip=None to build it per call")
            custom = not (isinstance(mip, MipStack) and mip.built_from(tex))
            mips = tuple(m.contiguous() for m in mip)
            n_levels = min(len(mips), _lib.MAX_MIP)
            if max_mip_level is not None:
                n_levels = min(n_levels, int(max_mip_level))
            mips = mips[:n_levels]
            h, w = tex.shape[1], tex.shape[2]
            for l, m in enumerate(mips):
                _check_tensor(f'mip[{l}]', m, torch.float32, 4)
                h, w = h // 2, w // 2
                if tuple(m.shape) != (tex.shape[0], h, w, tex.shape[3]):
                    raise ValueError(f"mip level {l + 1} must have shape {(tex.shape[0], h, w, tex.shape[3])} (got {tuple(m.shape)})")
        else:
            n_levels = _num_mip_levels(tex.shape[1], tex.shape[2], max_mip_level)
    else:
        uv_da = None
        mip_level_bias = None
    h = _hint_of(uv, 'zero') if (not mipped and tex.shape[0] == 1) else None
    empty_color = torch.empty(tex.shape[3], dtype=torch.float32, device=uv.device) if h else None
    out = _texture_func.apply(tex.contiguous(), uv.contiguous(), uv_da, mip_level_bias, _lib.FILTER[filter_mode],
                              _lib.BOUNDARY[boundary_mode], n_levels, h[0] if h else None, empty_color, custom, *mips)
    if h:
        _tag(out, h[0], 'const', empty_color)     # uv = (0,0) in empty bins: the texture's value there
    return out


# ----------------------------------------------------------------------------------------------
# antialias
# ----------------------------------------------------------------------------------------------

_topology_cache = OrderedDict()
_TOPOLOGY_CACHE_SIZE = 16


def antialias_construct_topology_hash(tri):
    """Edge adjacency of `tri` ([T,3] i32): adj [T,3] i32 (see include/fpcdr.h).  Built on the GPU."""
    _check_tensor('tri', tri, torch.int32, 2)
    lib = _lib.load()
    tri = tri.contiguous()
    T = tri.shape[0]
    scratch = torch.empty(lib.fpcdr_topology_scratch_bytes(T), dtype=torch.uint8, device=tri.device)
    adj = torch.empty(T, 3, dtype=torch.int32, device=tri.device)
    _lib.call("fpcdr_topology_build", _ptr(tri), T, _ptr(scratch), _ptr(adj), _stream())
    return adj


_tri_uv_cache = OrderedDict()


def _cached_tri_uv(uv, uv_tri):
    """uv[uv_tri] as [T,3,2], gathered once per (uv, uv_tri) pair: static per mesh, saves a dependent load per pixel."""
    key = (uv.data_ptr(), uv._version, uv_tri.data_ptr(), uv_tri._version, tuple(uv.shape), tuple(uv_tri.shape), str(uv.device))
    hit = _tri_uv_cache.get(key)
    if hit is not None:
        _tri_uv_cache.move_to_end(key)
        return hit[2]
    out = uv.detach()[uv_tri.long()].contiguous()
    _tri_uv_cache[key] = (uv, uv_tri, out)     # holding the tensors keeps their storage (and so the key) alive
    while len(_tri_uv_cache) > _TOPOLOGY_CACHE_SIZE:
        _tri_uv_cache.popitem(last=False)
    return out


def _cached_topology(tri):
    key = (tri.data_ptr(), tri._version, tuple(tri.shape), str(tri.device))
    hit = _topology_cache.get(key)
    if hit is not None:
        _topology_cache.move_to_end(key)
        return hit[1]
    adj = antialias_construct_topology_hash(tri)
    _topology_cache[key] = (tri, adj)  # holding `tri` keeps its storage (and so the key) alive
    while len(_topology_cache) > _TOPOLOGY_CACHE_SIZE:
        _topology_cache.popitem(last=False)
    return adj


class _antialias_func(torch.autograd.Function):
    @staticmethod
    def forward(ctx, color, rast, pos, tri, adj, boost, hint, empty_color):
        lib = _lib.load()
        B, H, W, C = color.shape
        V, T = pos.shape[1], tri.shape[0]
        dev = color.device
        out = torch.empty_like(color)
        sil = torch.empty(B, T, dtype=torch.uint8, device=dev)
        flags = torch.empty(lib.fpcdr_antialias_flags_bytes(B, H, W) // 8, dtype=torch.int64, device=dev)
        p = _lib.AntialiasFwd(color=_ptr(color), rast=_ptr(rast), pos=_ptr(pos), tri=_ptr(tri), adj=_ptr(adj), B=B, H=H,
                              W=W, C=C, V=V, T=T, sil=_ptr(sil), flags=_ptr(flags), out=_ptr(out), hint=_ptr(hint),
                              empty_color=_ptr(empty_color))
        _lib.call("fpcdr_antialias_fwd", ctypes.byref(p), _stream())
        ctx.save_for_backward(color, rast, pos, tri, adj, sil, flags)
        ctx.boost = float(boost)
        return out

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        color, rast, pos, tri, adj, sil, flags = ctx.saved_tensors
        B, H, W, C = color.shape
        V, T = pos.shape[1], tri.shape[0]
        g_color = torch.empty_like(color)
        g_pos = torch.zeros_like(pos)
        dy = dy.contiguous()
        p = _lib.AntialiasBwd(color=_ptr(color), rast=_ptr(rast), pos=_ptr(pos), tri=_ptr(tri), adj=_ptr(adj), dy=_ptr(dy),
                              B=B, H=H, W=W, C=C, V=V, T=T, sil=_ptr(sil), flags=_ptr(flags),
                              pos_gradient_boost=ctx.boost, grad_color=_ptr(g_color), grad_pos=_ptr(g_pos))
        _lib.call("fpcdr_antialias_bwd", ctypes.byref(p), _stream())
        return g_color, None, g_pos, None, None, None, None, None


def antialias(color, rast, pos, tri, topology_hash=None, pos_gradient_boost=1.0):
    """Silhouette antialiasing.  color [B,H,W,C], rast [B,H,W,4], pos [B,V,4], tri [T,3] -> [B,H,W,C]."""
    _check_tensor('color', color, torch.float32, 4)
    _check_tensor('rast', rast, torch.float32, 4)
    _check_tensor('pos', pos, torch.float32)
    _check_tensor('tri', tri, torch.int32, 2)
    if pos.dim() == 2:      # range mode: one shared vertex array for the whole minibatch (its gradient sums over the images)
        pos = pos[None].expand(color.shape[0], -1, -1)
    if pos.dim() != 3 or pos.shape[2] != 4:
        raise ValueError("pos must have shape [B,V,4]")
    if color.shape[:3] != rast.shape[:3] or rast.shape[3] != 4:
        raise ValueError("color [B,H,W,C] and rast [B,H,W,4] must agree in B, H, W")
    if pos.shape[0] != color.shape[0]:
        raise ValueError("pos minibatch must match color")
    tri = tri.contiguous()
    if topology_hash is None:
        adj = _cached_topology(tri)
    else:
        adj = topology_hash
        _check_tensor('topology_hash', adj, torch.int32, 2)
        if adj.shape != tri.shape:
            raise ValueError("topology_hash does not belong to this tri tensor")
    hr = _hint_of(rast, 'rast')
    hc = _hint_of(color, 'const') if hr else None
    if hc is not None and hc[0] is not hr[0]:     # a colour image from another rasterisation
        hc = None
    return _antialias_func.apply(color.contiguous(), rast.contiguous(), pos.contiguous(), tri, adj, pos_gradient_boost,
                                 hr[0] if hr else None, hc[1] if hc else None)
