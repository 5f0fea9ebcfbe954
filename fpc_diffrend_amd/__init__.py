"""
fpc_diffrend_amd -- MI355X-native differentiable-raster fitting path (gfx950 HIP kernels behind a
C ABI, include/fpcdr.h) with nvdiffrast-compatible Python signatures for the four ops the
reference's fit loop calls (reference src/torch/fit.py:151-160).

    import fpc_diffrend_amd.ops as dr      # dr.rasterize / interpolate / texture / antialias
"""
__version__ = "0.1.0"
