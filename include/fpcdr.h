/*
 * fpcdr.h -- C ABI of the MI355X (gfx950) differentiable-raster path.
 *
 * Drop-in boundary.  In the reference the four operators are Python calls
 * into the third-party package nvdiffrast (reference src/torch/fit.py:13); the reference itself
 * has no FFI.  Each entry point below states the reference call site whose work it performs; the
 * Python binding that preserves the nvdiffrast signatures is fpc_diffrend_amd/ops.py and the
 * binding a maintainer of the reference would add is shown in INTEGRATION.md.
 *
 * Conventions
 *   - plain C, no torch / HIP types: device pointers are void-free typed pointers into HBM, the
 *     stream is passed as void* (a hipStream_t); every launch is asynchronous on that stream;
 *   - the CALLER owns every buffer, including scratch; the library allocates nothing and keeps no
 *     state between calls (re-entrant, any number of devices / streams);
 *   - all tensors are contiguous, float32 / int32, NHWC images with row 0 = bottom scanline;
 *   - gradient outputs documented "accumulated" are added to with atomics and must be zeroed (or
 *     hold a running sum) on entry; all other outputs are fully overwritten;
 *   - return value 0 = launched, otherwise an FPCDR_E* code; fpcdr_last_error() gives the
 *     message (thread-local).  No exception crosses the boundary.
 */
#ifndef FPCDR_H
#define FPCDR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FPCDR_ABI_VERSION 11

enum {
    FPCDR_OK = 0,
    FPCDR_EINVAL = 1,   /* bad argument (null pointer, non-positive size, unsupported mode) */
    FPCDR_ELAUNCH = 2   /* the HIP runtime refused the launch */
};

int fpcdr_abi_version(void);
const char *fpcdr_last_error(void);

/* ------------------------------------------------------------------------------------------ */
/* rasterize -- dr.rasterize(glctx, pos, tri, resolution)        reference fit.py:151 (ctx :484) */
/* ------------------------------------------------------------------------------------------ */

/* REGION HINTS.  A face rig covers ~12 % of a 1080p frame; the other pixels of rast are zero, of the texture-coordinate image
 * zero, of the colour image one constant.  fpcdr_rasterize_fwd can leave a map of that: two planes of B x OY x OX bytes
 * (OY, OX = FPCDR_OCC_DIM(H), (W): 32 x 32-pixel bins); plane 0: 1 = some triangle's bounding box touches the bin (else every
 * pixel of the bin is EMPTY); plane 1: plane 0 OR-ed over the 3 x 3 neighbourhood.  An operator that is handed the hint of its
 * input (`hint` fields below; NULL = none) does not READ that input in empty bins -- it writes what the input implies -- which
 * halves the HBM traffic of the operator chain of fit.py:151-160.  Results are identical with and without hints.  A hint is
 * valid only for the very tensor it was produced with (the Python binding drops it when the tensor was modified). */
#define FPCDR_HINT_BYTES(B, H, W) ((size_t)2 * (B) * FPCDR_OCC_DIM(H) * FPCDR_OCC_DIM(W))

/* bytes of scratch fpcdr_rasterize_fwd needs for B images of T triangles: per image two record slots per triangle (one for the
 * triangle, one for the second piece of a triangle clipped against the near plane), their pixel boxes, chunk and image boxes */
size_t fpcdr_rasterize_scratch_bytes(int32_t B, int32_t T);

typedef struct {
    const float *pos;     /* [B,V,4] clip-space positions */
    const int32_t *tri;   /* [T,3] */
    int32_t B, V, T, H, W;
    void *scratch;        /* fpcdr_rasterize_scratch_bytes(B,T) bytes, 16-byte aligned */
    float *rast;          /* out [B,H,W,4] = (u, v, z/w, triangle index + 1; 0 = empty) */
    float *rast_db;       /* out [B,H,W,4] = (du/dx, du/dy, dv/dx, dv/dy) per pixel, or NULL */
    uint8_t *hint;        /* optional out, FPCDR_HINT_BYTES(B,H,W): REGION HINT of rast (see below), or NULL */
    const int32_t *ranges; /* optional [B,2] on the device: image b renders triangles [ranges[b][0], ranges[b][0] + ranges[b][1]) only
                              (nvdiffrast's range mode; triangle ids stay indices into tri), or NULL: every image renders all T */
} fpcdr_rasterize_fwd_params;
int fpcdr_rasterize_fwd(const fpcdr_rasterize_fwd_params *p, void *stream);

typedef struct {
    const float *pos;     /* [B,V,4] */
    const int32_t *tri;   /* [T,3] */
    const float *rast;    /* [B,H,W,4] forward output */
    const float *dy;      /* [B,H,W,4] dL/d rast (only .x .y are used: z/w and id carry no gradient) */
    const float *ddb;     /* [B,H,W,4] dL/d rast_db, or NULL */
    int32_t B, V, T, H, W;
    float *grad_pos;      /* [B,V,4] accumulated (x, y, w components; z receives nothing) */
    const uint8_t *hint;  /* optional: region hint of rast (empty bins are skipped without reading dy / rast) */
} fpcdr_rasterize_bwd_params;
int fpcdr_rasterize_bwd(const fpcdr_rasterize_bwd_params *p, void *stream);

/* ------------------------------------------------------------------------------------------ */
/* shared by the fused entry points                                                              */
/* ------------------------------------------------------------------------------------------ */

#define FPCDR_MAX_MIP 16

#define FPCDR_LOSS_SLOTS 256
#define FPCDR_OCC_BIN 32
#define FPCDR_OCC_DIM(n) (((n) + FPCDR_OCC_BIN - 1) / FPCDR_OCC_BIN)
/* bytes of the occupancy buffer (occ) and of the scratch buffer (cmask) of the fused entry points: fpcdr_objective_fwd below, and the
 * two-call form of include/fpcdr_twocall.h */
/* After a call of the two-call form (fpcdr_twocall.h: fpcdr_render_loss_fwd), int32 counts[4] at byte offset FPCDR_OCC_COUNTS_OFFSET(B,H,W) of occ hold: [0] bins the backward
 * call visits, [2] live bins of the rasteriser, [3] bins of the antialias pass -- what a caller feeds back (with a margin) as
 * cap_bwd / cap_bins / cap_fix of its NEXT calls. */
#define FPCDR_OCC_COUNTS_OFFSET(B, H, W) ((((size_t)(B) * FPCDR_OCC_DIM(H) * FPCDR_OCC_DIM(W) * 4) + 3) / 4 * 4)
size_t fpcdr_occ_bytes(int32_t B, int32_t H, int32_t W);
size_t fpcdr_cmask_bytes(int32_t B, int32_t H, int32_t W);
/* the scratch buffer of fpcdr_objective_fwd alone (its cmask): a quarter of the above (ABI v10) */
size_t fpcdr_objective_cmask_bytes(int32_t B, int32_t H, int32_t W);

/* out[i] += sum over the px_per_image pixels of image i of (ref - bg_scaled)^2, i < n_images; ref [n_images, px_per_image]
 * uint8, out f64 (zero-filled by the caller).  The part of the pixel loss (fit.py:579) that a sparse fpcdr_aa_loss_fwd
 * leaves to the caller: it depends on the reference images only, so a fit loop computes it once. */
int fpcdr_ref_bg_sumsq(const uint8_t *ref, int64_t n_images, int64_t px_per_image, float bg_scaled, double *out, void *stream);

/* Value of the pixel objective: out[0] = float((sum of the n_slots loss slots + bg_coeff * bg_sumsq[0]) / n_total), one launch.
 * bg_sumsq: device scalar (the all-background share of the sparse mode, fpcdr_ref_bg_sumsq summed over the call's images), or NULL. */
int fpcdr_objective_value(const double *loss_slots, int32_t n_slots, const double *bg_sumsq, double bg_coeff, double n_total,
                          float *out, void *stream);

/* ------------------------------------------------------------------------------------------ */
/* ONE-PASS pixel objective (ABI v8, v9) -- reference fit.py:151-161 + the pixel term of :579, VALUE AND GRADIENT in one call.
 *
 * The pixel objective is a scalar, so its gradient with respect to pos and tex does not depend on anything the caller does
 * afterwards: it is d(objective)/d(input) times one upstream scalar.  This entry point therefore computes value and gradient
 * together, while every covered pixel's barycentrics, taps, texels and vertices are still in registers, instead of writing
 * rast / colour / d loss/d colour (24 B/px) for a second call to read back and re-derive (fpcdr_render_loss_fwd +
 * fpcdr_render_aa_bwd; profiles/r04_backward_ablation.txt: 45 % of that backward's instructions were re-derivation):
 *   1. the rasteriser (same kernels and rules as fpcdr_rasterize_fwd) leaves only a 4-byte id per pixel of the occupied
 *      32 x 32 bins: (triangle + 1) | silhouette bits << 24;
 *   2. one shading kernel per occupied bin: barycentrics, texture lookup, background + squared error, and -- with the loss
 *      gradient of the UN-antialiased colour -- texture / interpolate / rasterize backward into grad_tex / grad_pos (per-bin LDS
 *      accumulators, one flush).  Pixels one of whose four pixel pairs has different ids and a triangle with a silhouette edge
 *      (read from the id planes, neighbours' border lines included) are DEFERRED: their (u, v, z/w), colour and gradient are also
 *      written, sparsely, to rec / color / grad_aa;
 *   3. two small kernels over the deferred pixels only (a few per cent of the covered ones): the exact antialias blend, the
 *      correction of their loss terms, and the scatter of the DIFFERENCE of their gradient (everything is linear in it).
 * grad_pos / grad_tex hold d(objective)/d(pos), d(objective)/d(tex) with the objective = grad_scale * sum of squares; the caller
 * multiplies by its upstream scalar.  Same value and gradients as the operator chain (tests/test_gpu_objective.py).
 * 'linear' lookup, or -- mip = 1 -- the reference's mip-mapped branch; C in {1, 3, 4}; instanced mode. */
size_t fpcdr_idplane_bytes(int32_t B, int32_t H, int32_t W);   /* 4 KB per 32 x 32 bin of the batch */
size_t fpcdr_binlist_bytes(int32_t B, int32_t H, int32_t W);   /* per-bin triangle lists: a count + 256 slots per bin (1 KB per bin) */

typedef struct {
    const float *pos;       /* [B,V,4] */
    const int32_t *tri;     /* [T,3] */
    const int32_t *adj;     /* [T,3] fpcdr_topology_build */
    int32_t B, V, T, H, W;
    void *scratch;          /* fpcdr_rasterize_scratch_bytes(B,T) */
    const float *uv;        /* [Vt,2] */
    const int32_t *uv_tri;  /* [T,3] */
    int32_t Vt;
    const float *tri_uv;    /* optional [T,3,2] = uv[uv_tri] */
    const float *tex;       /* [Ht,Wt,C] */
    int32_t Ht, Wt, C, boundary_mode;
    const uint8_t *ref;     /* [B,H,W] reference images, 8 bit */
    float bg, color_scale, grad_scale;
    uint8_t *sil;           /* scratch [B,T] */
    uint32_t *idp;          /* scratch, fpcdr_idplane_bytes(B,H,W), 16-byte aligned */
    uint16_t *occ;          /* scratch+out, fpcdr_occ_bytes(B,H,W): window masks; counts at FPCDR_OCC_COUNTS_OFFSET: [0] bins with a deferred
                               pixel, [2] live bins of the rasteriser, [3] occupied bins (cap_def / cap_bins / cap_occ of the next call) */
    uint32_t *cmask;        /* scratch, fpcdr_objective_cmask_bytes(B,H,W), 8-byte aligned */
    float *rec;             /* scratch [B,H,W,4]: (u, v, z/w, -) of DEFERRED pixels only (dense addressing, sparse writes; or rec_slots, below) */
    float *color;           /* scratch [B,H,W,C]: colour of deferred pixels only */
    float *grad_aa;         /* scratch [B,H,W,C]: d(objective)/d(antialiased colour) of deferred pixels only */
    float *empty_color;     /* out [4]: the colour of an empty pixel (the texture at uv = (0,0)) */
    double *loss_sum;       /* [FPCDR_LOSS_SLOTS] f64 accumulated: difference to an all-background image, as the sparse mode of fpcdr_aa_loss_fwd does */
    float *grad_pos;        /* [B,V,4] accumulated, or NULL */
    float *grad_tex;        /* [Ht,Wt,C] accumulated, or NULL (both NULL: value only) */
    int32_t cap_bins, cap_occ; /* launch-size hints (0 = none), as in fpcdr_aa_loss_fwd_params */
    int32_t cap_def;        /* launch-size hint for the kernels over the bins that hold a deferred pixel (count at [0] of the occ header) */
    int32_t sil_ready;      /* 1: `sil` already holds fpcdr_silhouette_bits() of this pos / tri / adj (a caller computes it on a second
                               stream beside the rasteriser's set-up kernel: ~70 us of a 2.7 ms call at 288 x 1080p); 0: computed here */
    uint64_t *flags;        /* optional (tests / diagnostics; NULL in production): the antialias flag planes of fpcdr_antialias_fwd --
                               which pixel pairs were blended --, fpcdr_antialias_flags_bytes(B,H,W), zero-filled by the caller */
    /* the reference's enable_mip branch (fit.py:153-155): interpolate with the rasteriser's screen-space derivatives and texture
     * 'linear-mipmap-linear', inside the same kernels (the derivatives never exist in HBM) */
    int32_t mip;            /* 1 = mip-mapped lookup */
    int32_t n_levels;       /* levels below tex, 0 .. FPCDR_MAX_MIP */
    const float *tex_mip[FPCDR_MAX_MIP];   /* tex_mip[l - 1] = level l, [Ht >> l, Wt >> l, C] (fpcdr_mip_downsample) */
    float *grad_tex_mip[FPCDR_MAX_MIP];    /* per level, accumulated (the caller folds them into grad_tex: fpcdr_mip_downsample_bwd) */
    void *binlist;          /* optional scratch, fpcdr_binlist_bytes(B,H,W), 4-byte aligned: per-bin triangle lists written by the set-up kernel
                               and read by the rasteriser (one count + one gather per bin instead of the scan of the chunks' boxes: 0.2 ms of
                               the call at 288 x 1080p); NULL: every bin scans.  Same result */
    void *sil_event;        /* with sil_ready = 1: optional hipEvent_t recorded behind the caller's fpcdr_silhouette_bits on ITS stream; the
                               call makes `stream` wait for it right before the first kernel that reads sil (behind the set-up kernels, which
                               is the point: they overlap).  NULL: sil is complete in stream order */
    /* ABI v9: the launches around the call folded into its first and its last kernel (each was 5-20 us of a 2.6 ms call's serial tail) */
    int32_t zero_outputs;   /* 1: the call's first kernel zero-fills loss_sum, grad_pos, grad_tex and grad_tex_mip (they need no initialisation);
                               0: they are accumulated into, as above */
    int32_t counts_seq;     /* any number that differs from call to call (see counts_out) */
    int32_t *counts_out;    /* optional [8]: a sequence lock.  The last kernel writes counts_seq to [6], then the four counters at
                               FPCDR_OCC_COUNTS_OFFSET of occ to [0..3], [5] += 1 if this call ran out of record slots (CUMULATIVE: only the
                               reader ever resets it), [7] = 1 if [1] is meaningful (compact records or count_only; a dense call counts no
                               slots), then counts_seq to [4], with system-scope fences in between.  May be HOST memory the device can write
                               (hipHostMalloc / pinned): the launch hints of the next call then need no device-to-host copy in the stream.
                               A reader takes [4], the counters, then [6]: the counters belong to ONE call iff [4] == [6] */
    const double *bg_sumsq; /* optional [1] (with value_out): the sum over the call's images of fpcdr_ref_bg_sumsq */
    double bg_coeff;        /* its coefficient (the number of colour channels) */
    double n_total;         /* the mean's denominator */
    float *value_out;       /* optional [1]: (sum of loss_sum + bg_coeff * bg_sumsq[0]) / n_total, what fpcdr_objective_value computes, from the
                               last kernel of the call */
    void *zero_extra;       /* optional: zero_extra_bytes (a multiple of 4) of the caller's own that the first kernel zero-fills as well -- a fit
                               step's small accumulators (the gradients of the camera matrices, poses and blend weights that its backward
                               kernels add into) instead of one fill launch each */
    int64_t zero_extra_bytes;
    /* ABI v10: COMPACT records.  rec / color / grad_aa are addressed by pixel of the whole batch above -- 24 B per pixel allocated (14 GB
     * at 288 x 1080p) for the 0.2 % of the pixels that are deferred.  With rec_slots > 0 they hold rec_slots SLOTS of 1 024 pixels instead
     * ([rec_slots * 1024, 4] / [.., C] / [.., C]); every occupied bin that shows a triangle with a silhouette edge -- the only bins that can
     * hold a deferred pixel -- takes the next slot.  The number of such bins is counted by every call (counts_out[1]; it keeps counting
     * beyond rec_slots), so a caller sizes rec_slots from the previous call on the batch, with a margin; a call that runs out of slots
     * bumps counts_out[5]: ITS RESULTS ARE THEN INVALID (the bins without a slot were shaded without their antialias pairs) and it SAYS SO
     * IN THE SAME CALL -- value_out receives NaN and skip_out 1 -- so that no optimiser step consumes them (fpcdr_adam_params.skip_flag).
     * The next call on the batch is sized from the demand this one counted.  A caller without a previous call asks first: count_only = 1
     * runs the rasteriser alone and reports the exact number for this very batch through counts_out (no record buffers, no gradients
     * needed). */
    int32_t rec_slots;      /* 0: dense addressing */
    int32_t count_only;
    int32_t *slot_map;      /* scratch with rec_slots > 0: one int32 per 32 x 32 bin of the batch (B * FPCDR_OCC_DIM(H) * FPCDR_OCC_DIM(W)) */
    /* ABI v11 */
    float *skip_out;        /* optional DEVICE [1]: the last kernel WRITES 1.0f when the call ran out of record slots (its value and gradients
                               are invalid), 0.0f otherwise.  Handed to fpcdr_adam_step as skip_flag the update of that step does not
                               happen; under data parallelism the caller sums it over the ranks first (it may be an element of the gradient
                               bucket), so that every rank skips the same step */
} fpcdr_objective_params;
int fpcdr_objective_fwd(const fpcdr_objective_params *p, void *stream);

/* sil[b][t] = 3 bits, "edge e of triangle t is a silhouette edge in image b" (boundary edge, or the two triangles' opposite vertices
 * project to the same side: DESIGN.md "Antialias rules"), as fpcdr_antialias_fwd and fpcdr_objective_fwd compute it themselves.
 * pos [B,V,4], tri [T,3], adj [T,3] (fpcdr_topology_build), sil out [B,T] bytes. */
int fpcdr_silhouette_bits(const float *pos, const int32_t *tri, const int32_t *adj, int32_t B, int32_t V, int32_t T, int32_t H, int32_t W,
                          uint8_t *sil, void *stream);

/* ------------------------------------------------------------------------------------------ */
/* interpolate -- dr.interpolate(attr, rast, tri[, rast_db, diff_attrs])  reference fit.py:154,157 */
/* ------------------------------------------------------------------------------------------ */

#define FPCDR_MAX_ATTR 32

typedef struct {
    const float *attr;    /* [Ba,Vt,A], Ba = 1 (broadcast) or B */
    const float *rast;    /* [B,H,W,4] */
    const int32_t *tri;   /* [T,3] index buffer of attr (e.g. uv_idx) */
    const float *rast_db; /* [B,H,W,4] or NULL when n_diff == 0 */
    int32_t B, H, W, Ba, Vt, A, T;
    int32_t n_diff;                     /* number of attributes with pixel differentials */
    int32_t diff_idx[FPCDR_MAX_ATTR];   /* their indices into A (diff_attrs='all' -> 0..A-1) */
    float *out;           /* out [B,H,W,A] */
    float *out_da;        /* out [B,H,W,2*n_diff] = (da/dx, da/dy) per selected attribute, or NULL */
    const uint8_t *hint;  /* optional: region hint of rast -- empty bins are written as zeros without reading rast */
} fpcdr_interpolate_fwd_params;
int fpcdr_interpolate_fwd(const fpcdr_interpolate_fwd_params *p, void *stream);

typedef struct {
    const float *attr, *rast;
    const int32_t *tri;
    const float *rast_db;
    const float *dy;      /* [B,H,W,A] */
    const float *dda;     /* [B,H,W,2*n_diff] or NULL */
    int32_t B, H, W, Ba, Vt, A, T;
    int32_t n_diff;
    int32_t diff_idx[FPCDR_MAX_ATTR];
    float *grad_attr;     /* [Ba,Vt,A] accumulated, or NULL (attr needs no gradient, as in fit.py:431) */
    float *grad_rast;     /* out [B,H,W,4] (.z .w are written as 0) */
    float *grad_rast_db;  /* out [B,H,W,4], or NULL when n_diff == 0 */
    const uint8_t *hint;  /* optional: region hint of rast -- empty bins get zero gradients without reading dy / rast */
} fpcdr_interpolate_bwd_params;
int fpcdr_interpolate_bwd(const fpcdr_interpolate_bwd_params *p, void *stream);

/* ------------------------------------------------------------------------------------------ */
/* texture -- dr.texture(tex, uv[, uv_da], filter_mode, boundary_mode, max_mip_level)             */
/*                                                                   reference fit.py:155,158    */
/* ------------------------------------------------------------------------------------------ */

enum { FPCDR_FILTER_NEAREST = 0, FPCDR_FILTER_LINEAR = 1, FPCDR_FILTER_LINEAR_MIPMAP_NEAREST = 2,
       FPCDR_FILTER_LINEAR_MIPMAP_LINEAR = 3 };
enum { FPCDR_BOUNDARY_WRAP = 0, FPCDR_BOUNDARY_CLAMP = 1, FPCDR_BOUNDARY_ZERO = 2 };   /* ZERO: texture padded with zeros */

/* level l+1 [N,Ht/2,Wt/2,C] = 2x2 box filter of level l [N,Ht,Wt,C] (Ht, Wt even) */
int fpcdr_mip_downsample(const float *src, float *dst, int32_t N, int32_t Ht, int32_t Wt, int32_t C, void *stream);
/* gradient of the above: grad_src [N,Ht,Wt,C] += 0.25 * grad_dst of its 2x2 parent (plain add, not atomic) */
int fpcdr_mip_downsample_bwd(const float *grad_dst, float *grad_src, int32_t N, int32_t Ht, int32_t Wt, int32_t C,
                             void *stream);

typedef struct {
    const float *tex[FPCDR_MAX_MIP + 1];  /* tex[0] = [Bt,Ht,Wt,C]; tex[l] = level l of the chain */
    int32_t n_levels;                     /* number of mip levels beyond level 0 (0 for non-mip filters) */
    const float *uv;                      /* [B,H,W,2] */
    const float *uv_da;                   /* [B,H,W,4] = (du/dx,du/dy,dv/dx,dv/dy) or NULL */
    const float *mip_level_bias;          /* [B,H,W] or NULL */
    int32_t B, H, W, Bt, Ht, Wt, C;
    int32_t filter_mode, boundary_mode;
    float *out;                           /* out [B,H,W,C] */
    const uint8_t *hint;                  /* optional (filter nearest / linear, Bt = 1): region hint of uv -- uv is (0,0) in empty
                                             bins, which get the texture's value there without reading uv */
    float *empty_color;                   /* with hint: out [C], that value (the hint of `out` for fpcdr_antialias_fwd) */
} fpcdr_texture_fwd_params;
int fpcdr_texture_fwd(const fpcdr_texture_fwd_params *p, void *stream);

typedef struct {
    const float *tex[FPCDR_MAX_MIP + 1];
    int32_t n_levels;
    const float *uv, *uv_da, *mip_level_bias;
    const float *dy;                      /* [B,H,W,C] */
    int32_t B, H, W, Bt, Ht, Wt, C;
    int32_t filter_mode, boundary_mode;
    float *grad_tex[FPCDR_MAX_MIP + 1];   /* per level, accumulated; grad_tex[0] may be NULL (tex needs no grad) */
    float *grad_uv;                       /* out [B,H,W,2] or NULL */
    float *grad_uv_da;                    /* out [B,H,W,4] or NULL */
    float *grad_mip_level_bias;           /* out [B,H,W] or NULL */
    const uint8_t *hint;                  /* optional (filter linear, C = 1, Bt = 1): region hint of uv, which is then not read
                                             in empty bins */
} fpcdr_texture_bwd_params;
int fpcdr_texture_bwd(const fpcdr_texture_bwd_params *p, void *stream);

/* ------------------------------------------------------------------------------------------ */
/* antialias -- dr.antialias(color, rast, pos, tri)                       reference fit.py:160   */
/* ------------------------------------------------------------------------------------------ */

/* Edge adjacency of an index buffer, built once per `tri` tensor (nvdiffrast's "topology hash").
 * adj[t][e], e = edge opposite to corner e (joins corners (e+1)%3 and (e+2)%3):
 *    >= 0  vertex index opposite to the edge in the one other triangle sharing it
 *    -1    boundary edge (no other triangle)      -2  shared by more than two triangles */
size_t fpcdr_topology_scratch_bytes(int32_t T);
int fpcdr_topology_build(const int32_t *tri, int32_t T, void *scratch, int32_t *adj /* out [T,3] */, void *stream);

/* words (uint64) per flag bit-plane row */
#define FPCDR_AA_ROW_WORDS(W) (((W) + 63) / 64)
/* bytes of the flag planes written by the forward pass and read by the backward pass */
size_t fpcdr_antialias_flags_bytes(int32_t B, int32_t H, int32_t W);

typedef struct {
    const float *color;   /* [B,H,W,C] */
    const float *rast;    /* [B,H,W,4] */
    const float *pos;     /* [B,V,4] */
    const int32_t *tri;   /* [T,3] */
    const int32_t *adj;   /* [T,3] from fpcdr_topology_build */
    int32_t B, H, W, C, V, T;
    uint8_t *sil;         /* scratch+saved [B,T]: per image, bit e set = edge e of t is a silhouette edge */
    uint64_t *flags;      /* saved, fpcdr_antialias_flags_bytes(): plane 0 = pair (p, p+x) blended, plane 1 = (p, p+y) */
    float *out;           /* out [B,H,W,C] */
    const uint8_t *hint;  /* optional: region hint of rast.  Where a bin and its eight neighbours are empty no pixel pair can be
                             blended: out = color there without reading rast */
    const float *empty_color; /* optional, with hint: [C], the value of `color` in empty bins (fpcdr_texture_fwd): it is then not
                             read there either */
} fpcdr_antialias_fwd_params;
int fpcdr_antialias_fwd(const fpcdr_antialias_fwd_params *p, void *stream);

typedef struct {
    const float *color, *rast, *pos;
    const int32_t *tri, *adj;
    const float *dy;      /* [B,H,W,C] */
    int32_t B, H, W, C, V, T;
    const uint8_t *sil;   /* from the forward pass */
    const uint64_t *flags;
    float pos_gradient_boost;
    float *grad_color;    /* out [B,H,W,C] */
    float *grad_pos;      /* [B,V,4] accumulated (x, y, w) */
} fpcdr_antialias_bwd_params;
int fpcdr_antialias_bwd(const fpcdr_antialias_bwd_params *p, void *stream);

/* ------------------------------------------------------------------------------------------ */
/* transform_clip for a minibatch                                      reference camera.py:11-23 */
/* ------------------------------------------------------------------------------------------ */

/* out[b][v] = mvp[b] . (verts[b / Nc][v], 1)      mvp [F*Nc,4,4] row-major, verts [F,V,3], out [F*Nc,V,4] */
int fpcdr_transform_clip_fwd(const float *mvp, const float *verts, float *out, int32_t F, int32_t Nc, int32_t V, void *stream);
/* grad_verts [F,V,3] (overwritten, may be NULL); grad_mvp [F*Nc,4,4] (accumulated: zero it first; may be NULL) */
int fpcdr_transform_clip_bwd(const float *mvp, const float *verts, const float *grad_out, float *grad_verts, float *grad_mvp,
                             int32_t F, int32_t Nc, int32_t V, void *stream);

/* mvp[f*Nc + c] = proj[c] . Rt(q_frame[f], t_frame[f]) . Rt(q_cam[c], t_cam[c]) . t_mv[c]      reference fit.py:541-553
 * (camera.rigid_grad, camera.py:128-132; roma.unitquat_to_rotmat on XYZW quaternions, NOT normalised, like the
 * reference after fit.py:616-618).  proj, t_mv [Nc,4,4] row-major; q [.,4]; t [.,3]; mvp [Fb*Nc,4,4].
 * bwd: the four gradient buffers are accumulated (zero them first).                                               */
int fpcdr_mvp_fwd(const float *proj, const float *t_mv, const float *q_cam, const float *t_cam, const float *q_frame,
                  const float *t_frame, float *mvp, int32_t Fb, int32_t Nc, void *stream);
int fpcdr_mvp_bwd(const float *proj, const float *t_mv, const float *q_cam, const float *t_cam, const float *q_frame,
                  const float *t_frame, const float *grad_mvp, float *gq_cam, float *gt_cam, float *gq_frame, float *gt_frame,
                  int32_t Fb, int32_t Nc, void *stream);
/* The same for a step that names its frames and views by INDEX into the full parameter tables (reference fit.py:525-526 draws one random
 * camera and frame per iteration and selects their rows with one-hot products, fit.py:547-550).  frame_idx [Fb] / view_idx [Nc]: int64 device
 * arrays or NULL (= 0 .. n - 1); view_idx indexes proj / t_mv, cam_of_view [views] (or NULL = identity) maps a view to its row of q_cam /
 * t_cam.  The backward ADDS into the full-size tables gq_cam / gt_cam / gq_frame / gt_frame (the caller zero-fills them). */
int fpcdr_mvp_fwd_indexed(const float *proj, const float *t_mv, const float *q_cam, const float *t_cam, const float *q_frame,
                          const float *t_frame, const int64_t *frame_idx, const int64_t *view_idx, const int64_t *cam_of_view, float *mvp,
                          int32_t Fb, int32_t Nc, void *stream);
int fpcdr_mvp_bwd_indexed(const float *proj, const float *t_mv, const float *q_cam, const float *t_cam, const float *q_frame,
                          const float *t_frame, const int64_t *frame_idx, const int64_t *view_idx, const int64_t *cam_of_view,
                          const float *grad_mvp, float *gq_cam, float *gt_cam, float *gq_frame, float *gt_frame, int32_t Fb, int32_t Nc,
                          void *stream);


/* uniform mesh Laplacian (reference fit.py:581, pytorch3d mesh_laplacian_smoothing 'uniform'), gather form:
 *   transpose = 0: out = L x,  L = D^-1 A - I;   transpose = 1: out = L^T x (the backward of the former)
 * x, out [F,V,3]; nbr [D,V] int32 one-ring table, slot-major: nbr[d][v] = d-th neighbour of v, rings stored front to
 * back and padded with indices >= V; inv_deg [V].                                                              */
int fpcdr_laplacian_gather(const float *x, const int32_t *nbr, const float *inv_deg, float *out, int32_t F, int32_t V, int32_t D,
                           int32_t transpose, void *stream);

/* The Laplacian term of the objective in two launches forward (gather + norms, then a one-workgroup finish at the kernel boundary) and one backward (reference fit.py:581 squares pytorch3d's mesh_laplacian_smoothing of
 * the ONE mesh of its step; a batch takes the mean of the squares):
 *   per[f] = mean_v || (L x_f)_v ||,   out[0] = weight / F * sum_f per[f]^2,   L = D^-1 A - I as in fpcdr_laplacian_gather.
 * lap [F,V,3] and per [F] are saved for the backward call.  acc: (F + 1) * 8 bytes, 8-byte aligned, ZERO on entry; the call leaves
 * it zero again.  Backward: grad_x [F,V,3] = upstream[0] * d out / d x (overwritten, not accumulated). */
int fpcdr_laplacian_penalty_fwd(const float *x, const int32_t *nbr, const float *inv_deg, float *lap, void *acc, float *per, float *out,
                                float weight, int32_t F, int32_t V, int32_t D, void *stream);
int fpcdr_laplacian_penalty_bwd(const float *lap, const int32_t *nbr, const float *inv_deg, const float *per, const float *upstream,
                                float *grad_x, float weight, int32_t F, int32_t V, int32_t D, void *stream);

/* ------------------------------------------------------------------------------------------ */
/* blend -- V = v_base + Bmat . w     reference fit.py:115-122 (prior), :58-62 (free)            */
/* ------------------------------------------------------------------------------------------ */

/* out[f][i] = v_base[i] + sum_k Bmat[i][k] * w[f][k]     (f32 MFMA, exact f32 accumulation)
 *   v_base [M] (may be NULL = 0), Bmat [M,K] row-major, w [F,K], out [F,M]                      */
int fpcdr_blend_fwd(const float *v_base, const float *Bmat, const float *w, float *out, int32_t M, int32_t K,
                    int32_t F, void *stream);
/* grad_w[f][k] += sum_i grad_out[f][i] * Bmat[i][k]  (accumulated: zero grad_w first)           */
int fpcdr_blend_bwd_w(const float *Bmat, const float *grad_out, float *grad_w, int32_t M, int32_t K, int32_t F,
                      void *stream);
/* grad_B[i][k] = sum_f grad_out[f][i] * w[f][k]      (free-form basis m3, reference fit.py:60; overwritten) */
int fpcdr_blend_bwd_basis(const float *w, const float *grad_out, float *grad_B, int32_t M, int32_t K, int32_t F,
                          void *stream);

/* The rig's weight algebra in front of the blend (ABI v9; reference fit.py:115-116 maps_intermediate . (maps . one-hot frame), :58-62
 * m2 . (m1 . one-hot frame)): w[fb][k] = sum_f mi[k][f] * maps[f][col(fb)], col(fb) = cols ? cols[fb] : col0 + fb -- the batch's frames
 * select columns of maps -- in the [Fb,K] layout fpcdr_blend_fwd reads.  mi [K,Fr], maps [Fr,Fc] row-major, cols [Fb] int64 or NULL.
 * bwd: grad_mi [K,Fr] and grad_maps [Fr,Fc] are OVERWRITTEN (columns no frame selects get zero); either may be NULL. */
int fpcdr_rig_weights_fwd(const float *mi, const float *maps, const int64_t *cols, int32_t col0, int32_t K, int32_t Fr, int32_t Fc,
                          int32_t Fb, float *w, void *stream);
int fpcdr_rig_weights_bwd(const float *mi, const float *maps, const int64_t *cols, int32_t col0, const float *grad_w, int32_t K,
                          int32_t Fr, int32_t Fc, int32_t Fb, float *grad_mi, float *grad_maps, void *stream);

/* ------------------------------------------------------------------------------------------ */
/* background composite + L2 pixel loss        reference fit.py:161 and the pixel term of :579   */
/* ------------------------------------------------------------------------------------------ */

/* loss_sum += sum over pixels, channels of (ref - color_scale * c)^2, c = covered ? color : bg
 * grad_color = grad_scale * d(that sum)/d color  (0 at uncovered pixels); one pass.              */
typedef struct {
    const float *color;    /* [B,H,W,C] */
    const float *rast;     /* [B,H,W,4]; covered = rast.w > 0 */
    const uint8_t *ref;    /* [B,H,W] 8-bit reference image, broadcast over C (reference: greyscale, C = 1) */
    int32_t B, H, W, C;
    float bg;              /* background colour, reference 45/255 */
    float color_scale;     /* reference 255 */
    float grad_scale;      /* 1 / (number of elements of the global mean) */
    double *loss_sum;      /* accumulated, one f64 */
    float *grad_color;     /* out [B,H,W,C], or NULL */
} fpcdr_pixel_loss_params;
int fpcdr_pixel_loss(const fpcdr_pixel_loss_params *p, void *stream);

/* ------------------------------------------------------------------------------------------ */
/* optimiser update        reference fit.py:493-505 (Adam, ten parameter groups), :610-618       */
/* ------------------------------------------------------------------------------------------ */

/* One Adam step (torch.optim.Adam arithmetic: betas, eps; no weight decay, no amsgrad) of up to FPCDR_ADAM_MAX_TENSORS
 * parameter tensors in ONE launch, each with its own learning rate and bias corrections (tensors that start training later
 * -- the learned basis of the combined mode, fit.py:603-608 -- count their own steps).  A tensor with `renorm` set is
 * afterwards divided by the Euclidean norm of the WHOLE tensor, the reference's quaternion "normalisation" (fit.py:616-618);
 * it may come without a gradient (grad = NULL), then only the division happens.  Everything is updated in place.          */
#define FPCDR_ADAM_MAX_TENSORS 16
typedef struct {
    float *param;          /* [n] */
    const float *grad;     /* [n], or NULL */
    float *exp_avg;        /* [n] first moment  (unused when grad is NULL) */
    float *exp_avg_sq;     /* [n] second moment */
    int64_t n;
    float step_size;       /* lr / (1 - beta1^step): the group's learning rate of this step (schedule applied by the caller) over
                              the first bias correction, divided in double precision as torch does */
    float bc2_sqrt;        /* sqrt(1 - beta2^step) */
    int32_t renorm;
    int32_t table_row;     /* ABI v10, with fpcdr_adam_params.step_table: this tensor's row of the table (step_size / bc2_sqrt above unused) */
    /* ABI v11, with fpcdr_adam_params.skipped: what step_size / bc2_sqrt were formed from -- the tensor's step count (this step included)
     * and its learning rate --, so that the kernel can re-form them for step - *skipped once a step has been skipped */
    int32_t step;
    float lr;
} fpcdr_adam_tensor;
typedef struct {
    int32_t n_tensors;
    float beta1, beta2, eps;
    float one_minus_beta1, one_minus_beta2;   /* rounded from the double-precision differences, as torch does */
    fpcdr_adam_tensor t[FPCDR_ADAM_MAX_TENSORS];
    /* ABI v10: optional DEVICE table of (step_size, bc2_sqrt) pairs, row t[i].table_row for tensor i; when given it replaces the two fields
     * of t[] -- a step captured in a HIP graph replays fixed kernel arguments, and its learning-rate schedule and bias corrections arrive
     * through this table (one small host-to-device copy per step in front of the replay) */
    const float *step_table;
    /* ABI v11: a step whose gradients are invalid is SKIPPED ON THE DEVICE (no host read-back, no exception between two collectives):
     * skip_flag  optional DEVICE [1]; when it holds a non-zero value the launch touches neither parameters nor moments (nor the
     *            quaternion division) -- fpcdr_objective_params.skip_out, summed over the ranks
     * skipped    optional DEVICE [1] counter of the steps skipped so far, kept by this kernel (+1 per skipped launch).  The host's step
     *            counters and learning-rate schedule have moved on regardless: with *skipped = s > 0 (and no step_table) the kernel forms
     *            step_size = lr * lr_skip_gain^s / (1 - beta1^(step - s)) and bc2_sqrt = sqrt(1 - beta2^(step - s)) itself, in double,
     *            i.e. the update of the run that never drew the skipped steps
     * lr_skip_gain  lr(i - 1) / lr(i) of the caller's schedule (1 for a constant rate; the reference's lr_ramp^(i / max_iter) decays by
     *            a constant factor per step, fit.py:506-507) */
    const float *skip_flag;
    int32_t *skipped;
    double lr_skip_gain;
} fpcdr_adam_params;
int fpcdr_adam_step(const fpcdr_adam_params *p, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* FPCDR_H */
