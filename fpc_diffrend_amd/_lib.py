"""
ctypes binding of libfpcdr.so (C ABI: include/fpcdr.h).  No CPU fallback exists: if the library is
missing or fails to load, every op raises -- the HIP path is the only product path.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FPCDR_LIB_PATH") or os.path.join(_HERE, "libfpcdr.so")  # override: A/B builds
TWOCALL_LIB_PATH = os.environ.get("FPCDR_TWOCALL_LIB_PATH") or os.path.join(_HERE, "libfpcdr_twocall.so")
CSRC = os.path.join(_HERE, "csrc")

MAX_ATTR = 32
MAX_MIP = 16
LOSS_SLOTS = 256
OCC_BIN = 32         # FPCDR_OCC_BIN
ABI_VERSION = 11

FILTER = {'nearest': 0, 'linear': 1, 'linear-mipmap-nearest': 2, 'linear-mipmap-linear': 3}
BOUNDARY = {'wrap': 0, 'clamp': 1, 'zero': 2}

_p = ctypes.c_void_p
_i = ctypes.c_int32


class RasterizeFwd(ctypes.Structure):
    _fields_ = [("pos", _p), ("tri", _p), ("B", _i), ("V", _i), ("T", _i), ("H", _i), ("W", _i), ("scratch", _p),
                ("rast", _p), ("rast_db", _p), ("hint", _p), ("ranges", _p)]


class RasterizeBwd(ctypes.Structure):
    _fields_ = [("pos", _p), ("tri", _p), ("rast", _p), ("dy", _p), ("ddb", _p), ("B", _i), ("V", _i), ("T", _i),
                ("H", _i), ("W", _i), ("grad_pos", _p), ("hint", _p)]


class RenderFwd(ctypes.Structure):
    _fields_ = [("pos", _p), ("tri", _p), ("B", _i), ("V", _i), ("T", _i), ("H", _i), ("W", _i), ("scratch", _p),
                ("uv", _p), ("uv_tri", _p), ("Vt", _i), ("tex", _p), ("Ht", _i), ("Wt", _i), ("C", _i),
                ("boundary_mode", _i), ("rast", _p), ("color", _p), ("tri_uv", _p), ("occ", _p), ("empty_color", _p),
                ("mip", _i), ("n_levels", _i), ("tex_mip", _p * MAX_MIP)]


class RenderBwd(ctypes.Structure):
    _fields_ = [("pos", _p), ("tri", _p), ("uv", _p), ("uv_tri", _p), ("tex", _p), ("rast", _p), ("dy", _p), ("B", _i),
                ("V", _i), ("T", _i), ("H", _i), ("W", _i), ("Vt", _i), ("Ht", _i), ("Wt", _i), ("C", _i),
                ("boundary_mode", _i), ("grad_pos", _p), ("grad_tex", _p), ("tri_uv", _p)]


class AaLossFwd(ctypes.Structure):
    _fields_ = [("color", _p), ("rast", _p), ("pos", _p), ("tri", _p), ("adj", _p), ("ref", _p), ("B", _i), ("H", _i),
                ("W", _i), ("C", _i), ("V", _i), ("T", _i), ("bg", ctypes.c_float), ("color_scale", ctypes.c_float),
                ("grad_scale", ctypes.c_float), ("sil", _p), ("flags", _p), ("grad_aa", _p), ("occ", _p), ("empty_color", _p),
                ("loss_sum", _p), ("cap_bins", _i), ("cap_fix", _i)]


class RenderAaBwd(ctypes.Structure):
    _fields_ = [("pos", _p), ("tri", _p), ("uv", _p), ("uv_tri", _p), ("tex", _p), ("rast", _p), ("color", _p),
                ("grad_aa", _p), ("sil", _p), ("flags", _p), ("occ", _p), ("empty_color", _p), ("B", _i), ("V", _i), ("T", _i),
                ("H", _i), ("W", _i),
                ("Vt", _i), ("Ht", _i), ("Wt", _i), ("C", _i), ("boundary_mode", _i), ("grad_pos", _p), ("grad_tex", _p),
                ("tri_uv", _p), ("upstream", _p), ("queued", _i), ("cap_bwd", _i), ("binflags", _i),
                ("mip", _i), ("n_levels", _i), ("tex_mip", _p * MAX_MIP), ("grad_tex_mip", _p * MAX_MIP)]


class Objective(ctypes.Structure):
    _fields_ = [("pos", _p), ("tri", _p), ("adj", _p), ("B", _i), ("V", _i), ("T", _i), ("H", _i), ("W", _i), ("scratch", _p),
                ("uv", _p), ("uv_tri", _p), ("Vt", _i), ("tri_uv", _p), ("tex", _p), ("Ht", _i), ("Wt", _i), ("C", _i),
                ("boundary_mode", _i), ("ref", _p), ("bg", ctypes.c_float), ("color_scale", ctypes.c_float),
                ("grad_scale", ctypes.c_float), ("sil", _p), ("idp", _p), ("occ", _p), ("cmask", _p), ("rec", _p), ("color", _p),
                ("grad_aa", _p), ("empty_color", _p), ("loss_sum", _p), ("grad_pos", _p), ("grad_tex", _p), ("cap_bins", _i),
                ("cap_occ", _i), ("cap_def", _i), ("sil_ready", _i), ("flags", _p), ("mip", _i), ("n_levels", _i),
                ("tex_mip", _p * MAX_MIP), ("grad_tex_mip", _p * MAX_MIP), ("binlist", _p), ("sil_event", _p),
                ("zero_outputs", _i), ("counts_seq", _i), ("counts_out", _p), ("bg_sumsq", _p), ("bg_coeff", ctypes.c_double),
                ("n_total", ctypes.c_double), ("value_out", _p), ("zero_extra", _p), ("zero_extra_bytes", ctypes.c_int64),
                ("rec_slots", _i), ("count_only", _i), ("slot_map", _p), ("skip_out", _p)]


class InterpolateFwd(ctypes.Structure):
    _fields_ = [("attr", _p), ("rast", _p), ("tri", _p), ("rast_db", _p), ("B", _i), ("H", _i), ("W", _i), ("Ba", _i),
                ("Vt", _i), ("A", _i), ("T", _i), ("n_diff", _i), ("diff_idx", _i * MAX_ATTR), ("out", _p),
                ("out_da", _p), ("hint", _p)]


class InterpolateBwd(ctypes.Structure):
    _fields_ = [("attr", _p), ("rast", _p), ("tri", _p), ("rast_db", _p), ("dy", _p), ("dda", _p), ("B", _i),
                ("H", _i), ("W", _i), ("Ba", _i), ("Vt", _i), ("A", _i), ("T", _i), ("n_diff", _i),
                ("diff_idx", _i * MAX_ATTR), ("grad_attr", _p), ("grad_rast", _p), ("grad_rast_db", _p), ("hint", _p)]


class TextureFwd(ctypes.Structure):
    _fields_ = [("tex", _p * (MAX_MIP + 1)), ("n_levels", _i), ("uv", _p), ("uv_da", _p), ("mip_level_bias", _p),
                ("B", _i), ("H", _i), ("W", _i), ("Bt", _i), ("Ht", _i), ("Wt", _i), ("C", _i), ("filter_mode", _i),
                ("boundary_mode", _i), ("out", _p), ("hint", _p), ("empty_color", _p)]


class TextureBwd(ctypes.Structure):
    _fields_ = [("tex", _p * (MAX_MIP + 1)), ("n_levels", _i), ("uv", _p), ("uv_da", _p), ("mip_level_bias", _p),
                ("dy", _p), ("B", _i), ("H", _i), ("W", _i), ("Bt", _i), ("Ht", _i), ("Wt", _i), ("C", _i),
                ("filter_mode", _i), ("boundary_mode", _i), ("grad_tex", _p * (MAX_MIP + 1)), ("grad_uv", _p),
                ("grad_uv_da", _p), ("grad_mip_level_bias", _p), ("hint", _p)]


class AntialiasFwd(ctypes.Structure):
    _fields_ = [("color", _p), ("rast", _p), ("pos", _p), ("tri", _p), ("adj", _p), ("B", _i), ("H", _i), ("W", _i),
                ("C", _i), ("V", _i), ("T", _i), ("sil", _p), ("flags", _p), ("out", _p), ("hint", _p), ("empty_color", _p)]


class AntialiasBwd(ctypes.Structure):
    _fields_ = [("color", _p), ("rast", _p), ("pos", _p), ("tri", _p), ("adj", _p), ("dy", _p), ("B", _i), ("H", _i),
                ("W", _i), ("C", _i), ("V", _i), ("T", _i), ("sil", _p), ("flags", _p),
                ("pos_gradient_boost", ctypes.c_float), ("grad_color", _p), ("grad_pos", _p)]


class PixelLoss(ctypes.Structure):
    _fields_ = [("color", _p), ("rast", _p), ("ref", _p), ("B", _i), ("H", _i), ("W", _i), ("C", _i),
                ("bg", ctypes.c_float), ("color_scale", ctypes.c_float), ("grad_scale", ctypes.c_float),
                ("loss_sum", _p), ("grad_color", _p)]


ADAM_MAX_TENSORS = 16     # FPCDR_ADAM_MAX_TENSORS


class AdamTensor(ctypes.Structure):
    _fields_ = [("param", _p), ("grad", _p), ("exp_avg", _p), ("exp_avg_sq", _p), ("n", ctypes.c_int64),
                ("step_size", ctypes.c_float), ("bc2_sqrt", ctypes.c_float), ("renorm", _i), ("table_row", _i),
                ("step", _i), ("lr", ctypes.c_float)]


class AdamParams(ctypes.Structure):
    _fields_ = [("n_tensors", _i), ("beta1", ctypes.c_float), ("beta2", ctypes.c_float), ("eps", ctypes.c_float),
                ("one_minus_beta1", ctypes.c_float), ("one_minus_beta2", ctypes.c_float), ("t", AdamTensor * ADAM_MAX_TENSORS),
                ("step_table", _p), ("skip_flag", _p), ("skipped", _p), ("lr_skip_gain", ctypes.c_double)]


# every symbol include/fpcdr.h declares: name -> (restype, argtypes)
_sz = ctypes.c_size_t
_int = ctypes.c_int
SYMBOLS = {
    "fpcdr_abi_version": (_int, []),
    "fpcdr_last_error": (ctypes.c_char_p, []),
    "fpcdr_rasterize_scratch_bytes": (_sz, [_i, _i]),
    "fpcdr_rasterize_fwd": (_int, [ctypes.POINTER(RasterizeFwd), _p]),
    "fpcdr_rasterize_bwd": (_int, [ctypes.POINTER(RasterizeBwd), _p]),
    "fpcdr_occ_bytes": (_sz, [_i, _i, _i]),
    "fpcdr_cmask_bytes": (_sz, [_i, _i, _i]),
    "fpcdr_objective_cmask_bytes": (_sz, [_i, _i, _i]),
    "fpcdr_ref_bg_sumsq": (_int, [_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_float, _p, _p]),
    "fpcdr_objective_value": (_int, [_p, _i, _p, ctypes.c_double, ctypes.c_double, _p, _p]),
    "fpcdr_idplane_bytes": (_sz, [_i, _i, _i]),
    "fpcdr_binlist_bytes": (_sz, [_i, _i, _i]),
    "fpcdr_objective_fwd": (_int, [ctypes.POINTER(Objective), _p]),
    "fpcdr_silhouette_bits": (_int, [_p, _p, _p, _i, _i, _i, _i, _i, _p, _p]),
    "fpcdr_interpolate_fwd": (_int, [ctypes.POINTER(InterpolateFwd), _p]),
    "fpcdr_interpolate_bwd": (_int, [ctypes.POINTER(InterpolateBwd), _p]),
    "fpcdr_mip_downsample": (_int, [_p, _p, _i, _i, _i, _i, _p]),
    "fpcdr_mip_downsample_bwd": (_int, [_p, _p, _i, _i, _i, _i, _p]),
    "fpcdr_texture_fwd": (_int, [ctypes.POINTER(TextureFwd), _p]),
    "fpcdr_texture_bwd": (_int, [ctypes.POINTER(TextureBwd), _p]),
    "fpcdr_topology_scratch_bytes": (_sz, [_i]),
    "fpcdr_topology_build": (_int, [_p, _i, _p, _p, _p]),
    "fpcdr_antialias_flags_bytes": (_sz, [_i, _i, _i]),
    "fpcdr_antialias_fwd": (_int, [ctypes.POINTER(AntialiasFwd), _p]),
    "fpcdr_antialias_bwd": (_int, [ctypes.POINTER(AntialiasBwd), _p]),
    "fpcdr_transform_clip_fwd": (_int, [_p, _p, _p, _i, _i, _i, _p]),
    "fpcdr_transform_clip_bwd": (_int, [_p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "fpcdr_mvp_fwd": (_int, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _p]),
    "fpcdr_mvp_bwd": (_int, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _p]),
    "fpcdr_mvp_fwd_indexed": (_int, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _p]),
    "fpcdr_mvp_bwd_indexed": (_int, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _p]),
    "fpcdr_laplacian_gather": (_int, [_p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "fpcdr_laplacian_penalty_fwd": (_int, [_p, _p, _p, _p, _p, _p, _p, ctypes.c_float, _i, _i, _i, _p]),
    "fpcdr_laplacian_penalty_bwd": (_int, [_p, _p, _p, _p, _p, _p, ctypes.c_float, _i, _i, _i, _p]),
    "fpcdr_blend_fwd": (_int, [_p, _p, _p, _p, _i, _i, _i, _p]),
    "fpcdr_blend_bwd_w": (_int, [_p, _p, _p, _i, _i, _i, _p]),
    "fpcdr_blend_bwd_basis": (_int, [_p, _p, _p, _i, _i, _i, _p]),
    "fpcdr_rig_weights_fwd": (_int, [_p, _p, _p, _i, _i, _i, _i, _i, _p, _p]),
    "fpcdr_rig_weights_bwd": (_int, [_p, _p, _p, _i, _p, _i, _i, _i, _i, _p, _p, _p]),
    "fpcdr_pixel_loss": (_int, [ctypes.POINTER(PixelLoss), _p]),
    "fpcdr_adam_step": (_int, [ctypes.POINTER(AdamParams), _p]),
}

# the two-call form of the pixel objective + the fused render pair (include/fpcdr_twocall.h): exported by libfpcdr_twocall.so only
SYMBOLS_TWOCALL = {
    "fpcdr_render_fwd": (_int, [ctypes.POINTER(RenderFwd), _p]),
    "fpcdr_render_bwd": (_int, [ctypes.POINTER(RenderBwd), _p]),
    "fpcdr_aa_loss_fwd": (_int, [ctypes.POINTER(AaLossFwd), _p]),
    "fpcdr_render_loss_fwd": (_int, [ctypes.POINTER(RenderFwd), ctypes.POINTER(AaLossFwd), _p, _p]),
    "fpcdr_render_aa_bwd": (_int, [ctypes.POINTER(RenderAaBwd), _p]),
}

_lib = None
_lib_twocall = None


def build(force=False):
    """hipcc --offload-arch=gfx950 build of libfpcdr.so (recipe: csrc/Makefile); cross-compiles without a GPU."""
    cmd = ["make", "-s", "-C", CSRC, "-j8"] + (["-B"] if force else [])
    subprocess.check_call(cmd)
    return LIB_PATH


def load():
    """Load libfpcdr.so and bind every symbol of include/fpcdr.h.  Raises if it is missing -- no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; "
            f"g.build()'` (or `make -C {CSRC}`). There is no CPU fallback for the raster ops.")
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64 (soname libamdhip64.so.7, the
    # same soname libfpcdr.so needs).  Importing torch first makes the loader bind our library to the runtime
    # torch already uses, so streams / device pointers are shared; loaded the other way round the process
    # ends up with two runtimes and launches fail with "no ROCm-capable device is detected".
    import torch
    hip_rt = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    if os.path.exists(hip_rt):
        ctypes.CDLL(hip_rt, mode=ctypes.RTLD_GLOBAL)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    ver = lib.fpcdr_abi_version()
    if ver != ABI_VERSION:
        raise RuntimeError(f"libfpcdr.so ABI version {ver} != binding {ABI_VERSION}; rebuild the extension")
    _lib = lib
    return lib


def load_twocall():
    """libfpcdr_twocall.so: the superseded two-call form of the pixel objective and the fused render pair (include/fpcdr_twocall.h).
    A complete library of its own -- every symbol of fpcdr.h plus SYMBOLS_TWOCALL -- loaded only when one of those entry points is
    asked for (ops.pixel_objective(one_pass=False), ops.render_textured); the fit loop never does."""
    global _lib_twocall
    if _lib_twocall is not None:
        return _lib_twocall
    load()      # (binds the HIP runtime torch uses)
    if not os.path.exists(TWOCALL_LIB_PATH):
        raise RuntimeError(f"{TWOCALL_LIB_PATH} not found: run `make -C {CSRC}` (or __graft_entry__.build()). The two-call form of the "
                           "pixel objective and ops.render_textured live in this second library; there is no fallback.")
    lib = ctypes.CDLL(TWOCALL_LIB_PATH)
    for name, (res, args) in list(SYMBOLS.items()) + list(SYMBOLS_TWOCALL.items()):
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.fpcdr_abi_version() != ABI_VERSION:
        raise RuntimeError(f"libfpcdr_twocall.so ABI version {lib.fpcdr_abi_version()} != binding {ABI_VERSION}; rebuild the extension")
    _lib_twocall = lib
    return lib


class KernelTimer:
    """Per-entry-point device time, measured with HIP events recorded on the launch stream (torch's current
    stream, the one every fpcdr_* launch goes to).  Used by bench.py for the roofline figures."""

    def __init__(self, names=None):
        self.records = {}
        self.names = set(names) if names is not None else None   # None = every entry point; else only these

    def add(self, name, e0, e1):
        self.records.setdefault(name, []).append((e0, e1))

    def summary(self):
        """name -> (calls, total milliseconds); synchronises."""
        import torch
        torch.cuda.synchronize()
        return {k: (len(v), sum(a.elapsed_time(b) for a, b in v)) for k, v in self.records.items()}


TIMER = None  # set to a KernelTimer() to time every C-ABI call


def call(name, *args, twocall=False):
    """Invoke one C-ABI entry point, raise on a non-zero return code.  twocall (or a name of SYMBOLS_TWOCALL): from libfpcdr_twocall.so --
    a code path of the two-call form makes ALL its calls there, so that the scratch layouts its kernels share come from one library."""
    twocall = twocall or name in SYMBOLS_TWOCALL
    fn = getattr(load_twocall() if twocall else load(), name)
    t = TIMER
    if t is not None and (t.names is None or name in t.names):
        import torch
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = fn(*args)
        e1.record()
        t.add(name, e0, e1)
    else:
        rc = fn(*args)
    check(rc, twocall)


def check(rc, twocall=False):
    if rc != 0:
        msg = (load_twocall() if twocall else load()).fpcdr_last_error()
        raise RuntimeError(f"fpcdr error {rc}: {msg.decode() if msg else '?'}")
