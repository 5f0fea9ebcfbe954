/*
 * fpcdr_twocall.h -- the TWO-CALL form of the pixel objective and the fused render pair: rounds 1-3 of this build, superseded in the fit loop
 * by fpcdr_objective_fwd (fpcdr.h) and kept as its parity partner, for the dense (non-sparse) mode and for ops.render_textured.
 *
 * These entry points are NOT exported by libfpcdr.so.  They live in libfpcdr_twocall.so (csrc/Makefile) -- a complete library: every
 * symbol of fpcdr.h plus the ones below -- which fpc_diffrend_amd/_lib.py loads only when one of them is asked for.
 */
#ifndef FPCDR_TWOCALL_H
#define FPCDR_TWOCALL_H

#include "fpcdr.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------ */
/* render -- rasterize + interpolate(uv) + texture('linear') fused   reference fit.py:151,157,158 */
/* ------------------------------------------------------------------------------------------ */


/* The non-mip branch of the reference's render() up to the antialias call, in one pass: the wave that resolves
 * a pixel also interpolates its texture coordinate and taps the texture, so texc never exists in HBM
 * (20 B/px written instead of 52 read+written).  Results are identical to the three separate calls.
 * scratch: fpcdr_rasterize_scratch_bytes(B, T).                                                    */
typedef struct {
    const float *pos;       /* [B,V,4] */
    const int32_t *tri;     /* [T,3] */
    int32_t B, V, T, H, W;
    void *scratch;
    const float *uv;        /* [Vt,2] texture coordinates (one set, broadcast over B: reference uv[None]) */
    const int32_t *uv_tri;  /* [T,3] */
    int32_t Vt;
    const float *tex;       /* [Ht,Wt,C] (one texture, broadcast over B: reference tex[None]) */
    int32_t Ht, Wt, C, boundary_mode;
    float *rast;            /* out [B,H,W,4] */
    float *color;           /* out [B,H,W,C] */
    const float *tri_uv;    /* optional [T,3,2]: uv[uv_tri] gathered once per mesh (saves one dependent load per pixel); NULL = look up */
    uint16_t *occ;          /* NULL = dense (every pixel of rast / color is written).  Otherwise SPARSE mode: out, a buffer of
                               fpcdr_occ_bytes(B,H,W) bytes (4-byte aligned).  A 32x32-pixel bin is OCCUPIED if the bounding box of some
                               triangle touches it; unoccupied bins hold no covered pixel and are NOT written at all -- only
                               for consumers that read the map (fpcdr_aa_loss_fwd / fpcdr_render_aa_bwd) and take those
                               pixels as empty: rast = 0, colour = empty_color.  The first B*OY*OX uint16 (OY, OX =
                               FPCDR_OCC_DIM(H), (W)) are per-bin WINDOW masks: bit (dy+1)*4 + (dx+1) = bin (x+dx, y+dy) is
                               occupied, dx in -1..2, dy in -1..1; the rest of the buffer belongs to the library (raw map, and the
                               list of bins the backward call visits with its work cursor). */
    float *empty_color;     /* sparse mode: out [4], the colour of an empty pixel (the texture at uv = (0,0), fit.py:157-158) */
    /* ABI v7, fpcdr_render_loss_fwd only: the reference's enable_mip branch (fit.py:153-155) -- interpolate with the rasteriser's
     * screen-space derivatives (diff_attrs='all') and texture 'linear-mipmap-linear' -- inside the same kernels (C = 1, 3, 4). */
    int32_t mip;            /* 1 = mip-mapped lookup (0: the 'linear' lookup above) */
    int32_t n_levels;       /* levels below tex in the chain, 0 .. FPCDR_MAX_MIP (nvdiffrast's max_mip_level, already clamped) */
    const float *tex_mip[FPCDR_MAX_MIP];   /* tex_mip[l - 1] = level l, [Ht >> l, Wt >> l, C] (fpcdr_mip_downsample) */
} fpcdr_render_fwd_params;
int fpcdr_render_fwd(const fpcdr_render_fwd_params *p, void *stream);

/* Backward of the above: dy = dL/d color; reads dy and rast, writes nothing dense. */
typedef struct {
    const float *pos;
    const int32_t *tri;
    const float *uv;
    const int32_t *uv_tri;
    const float *tex;
    const float *rast;      /* forward output */
    const float *dy;        /* [B,H,W,C] */
    int32_t B, V, T, H, W, Vt, Ht, Wt, C, boundary_mode;
    float *grad_pos;        /* [B,V,4] accumulated, or NULL */
    float *grad_tex;        /* [Ht,Wt,C] accumulated, or NULL */
    const float *tri_uv;    /* optional [T,3,2], as in fpcdr_render_fwd */
} fpcdr_render_bwd_params;
int fpcdr_render_bwd(const fpcdr_render_bwd_params *p, void *stream);


/* antialias + background + pixel loss in one pass (reference fit.py:160, 161, 579): reads colour, rast and the 8-bit
 * reference image, accumulates the sum of squares and writes d(grad_scale * sum)/d(antialiased colour); the
 * antialiased image itself is never stored.  sil / flags as in fpcdr_antialias_fwd.  C in {1, 3, 4}.          */
typedef struct {
    const float *color;    /* [B,H,W,C] from fpcdr_render_fwd */
    const float *rast;     /* [B,H,W,4] */
    const float *pos;      /* [B,V,4] */
    const int32_t *tri;    /* [T,3] */
    const int32_t *adj;    /* [T,3] */
    const uint8_t *ref;    /* [B,H,W] */
    int32_t B, H, W, C, V, T;
    float bg, color_scale, grad_scale;
    uint8_t *sil;          /* scratch+saved [B,T] */
    uint64_t *flags;       /* saved, fpcdr_antialias_flags_bytes() */
    float *grad_aa;        /* out [B,H,W,C] (sparse mode: only the pixels of occupied bins) */
    const uint16_t *occ;   /* NULL = dense, else the map written by fpcdr_render_fwd (sparse mode): flags must be zero-filled
                              by the caller, and loss_sum receives only the DIFFERENCE to an all-background image,
                              sum over covered pixels of (ref - s col)^2 - (ref - s bg)^2: the caller adds
                              C * sum over all pixels of (ref - s bg)^2, which depends on the reference images alone
                              (fpcdr_ref_bg_sumsq) */
    const float *empty_color; /* sparse mode: [4] from fpcdr_render_fwd */
    double *loss_sum;      /* [FPCDR_LOSS_SLOTS] f64, accumulated: the loss is the sum of all slots (workgroups spread
                              their partial sums over the slots instead of hammering one address) */
    int32_t cap_bins, cap_fix; /* fpcdr_render_loss_fwd only: launch-size HINTS for its two list kernels (0 = none): at least
                              the number of live / antialias-fix bins an earlier call on a similar batch reported (see
                              FPCDR_OCC_COUNTS), plus a margin.  Results never depend on them: entries beyond a hint are
                              swept up by a second, strided launch; without a hint one workgroup per bin of the batch is
                              dispatched (0.15-0.2 ms per kernel at 288 x 1080p). */
} fpcdr_aa_loss_fwd_params;
int fpcdr_aa_loss_fwd(const fpcdr_aa_loss_fwd_params *p, void *stream);

/* fpcdr_render_fwd (sparse mode) + fpcdr_aa_loss_fwd in one call -- reference fit.py:151-161 (render: rasterize, interpolate,
 * texture, antialias, background) and the pixel term of fit.py:579 -- without the dense antialias pass: the rasteriser's
 * workgroup, which still holds its bin's ids, gives every pixel that antialiasing cannot touch (no pixel pair with
 * different ids at a silhouette edge) its loss term and gradient straight away and leaves a bit mask of the others
 * (bin-border pixels included); a second kernel runs antialias + loss on those candidates only.  Same outputs as the
 * two calls.  r->occ, r->empty_color must be set (sparse mode only); l->color / rast / pos / tri / occ / empty_color
 * must equal r's; l->flags is zeroed by the call itself (ABI v7; fpcdr_aa_loss_fwd still wants it zero-filled); cmask: scratch of fpcdr_cmask_bytes(B,H,W) bytes, 8-byte aligned
 * (per bin 32 row masks of candidate pixels and the bin's four border lines; the work lists of the call's kernels).
 * The rasteriser and the antialias pass run over compact LISTS of the occupied bins (l->cap_bins, l->cap_fix).      */
int fpcdr_render_loss_fwd(const fpcdr_render_fwd_params *r, const fpcdr_aa_loss_fwd_params *l, uint32_t *cmask, void *stream);


/* Backward of antialias + texture + interpolate + rasterize in one pass: reads grad_aa (4C B/px), rast (16 B/px) and
 * the flag planes; scatters into grad_pos and grad_tex; writes nothing dense.                                 */
typedef struct {
    const float *pos;
    const int32_t *tri;
    const float *uv;
    const int32_t *uv_tri;
    const float *tex;
    const float *rast, *color, *grad_aa;
    const uint8_t *sil;
    const uint64_t *flags;
    uint16_t *occ;         /* NULL = dense, else the occupancy buffer of the forward call (sparse mode).  After
                              fpcdr_render_loss_fwd it also holds the list of bins to visit; the call resets its work cursor */
    const float *empty_color; /* sparse mode: [4] from fpcdr_render_fwd */
    int32_t B, V, T, H, W, Vt, Ht, Wt, C, boundary_mode;
    float *grad_pos;       /* [B,V,4] accumulated */
    float *grad_tex;       /* [Ht,Wt,C] accumulated, or NULL */
    const float *tri_uv;   /* optional [T,3,2], as in fpcdr_render_fwd */
    const float *upstream; /* optional device scalar: d(final loss)/d(objective), multiplied into both gradients (NULL = 1) */
    int32_t queued;        /* 1: occ was filled by fpcdr_render_loss_fwd -- visit only the bins on its list;
                              0: one workgroup per bin of the batch (dense mode, or occ from fpcdr_render_fwd) */
    int32_t cap_bwd;       /* queued = 1: launch-size hint for the list kernel, as cap_bins above (0 = none) */
    int32_t binflags;      /* 1: occ was filled by fpcdr_render_loss_fwd, which also left a per-bin summary of the flag planes:
                              flag words are then loaded only near bins that hold a blended pair */
    /* ABI v7: backward of the mip-mapped forward (fpcdr_render_fwd_params.mip; one workgroup per bin) */
    int32_t mip, n_levels;
    const float *tex_mip[FPCDR_MAX_MIP];        /* as in the forward call */
    float *grad_tex_mip[FPCDR_MAX_MIP];         /* per level, accumulated (the caller folds them into grad_tex: fpcdr_mip_downsample_bwd) */
} fpcdr_render_aa_bwd_params;
int fpcdr_render_aa_bwd(const fpcdr_render_aa_bwd_params *p, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* FPCDR_TWOCALL_H */
