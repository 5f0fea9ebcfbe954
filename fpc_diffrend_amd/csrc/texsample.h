// Bilinear texture sampling shared by texture.hip and the fused render kernels (rasterize.hip).
#pragma once
#include "common.h"

__device__ __forceinline__ int wrap_i(int i, int n, int mode) {
    if (mode == FPCDR_BOUNDARY_WRAP) {
        int r = i % n;
        return r < 0 ? r + n : r;
    }
    return min(max(i, 0), n - 1);
}

// wrap_i without the integer division (~35 instructions each, four per sample) for the usual case of an index in
// [-n, 2n): the taps of a coordinate in [0, 1] have x0 in [-1, n - 1] and x0 + 1 in [0, n].  Anything else (a NaN
// coordinate converts to INT_MIN) takes the general path, so the result is always a valid index.
__device__ __forceinline__ int wrap_near(int i, int n, int mode) {
    if (mode == FPCDR_BOUNDARY_WRAP) {
        if (__builtin_expect(i >= -n && i < 2 * n, 1)) return i < 0 ? i + n : (i >= n ? i - n : i);
        return wrap_i(i, n, mode);
    }
    return min(max(i, 0), n - 1);
}

__device__ __forceinline__ float prep_coord(float u, int mode) {
    if (mode == FPCDR_BOUNDARY_WRAP) return u - floorf(u);
    if (mode == FPCDR_BOUNDARY_ZERO) return u;             // the texture is padded with zeros: coordinates stay as they are
    return fminf(fmaxf(u, 0.0f), 1.0f);
}

struct Taps {
    int i00, i10, i01, i11;  // element offsets (texel index * C) within one texture image
    float fx, fy;
    unsigned int valid;      // boundary mode 'zero': bit k set = tap k (00, 10, 01, 11) lies inside the texture; else 0xF.  The
                             // offsets are always in range, so loads are safe; consumers of 'zero' mask values and skip scatters
};

__device__ __forceinline__ Taps make_taps(float u, float v, int Ht, int Wt, int C, int mode) {
    const float x = prep_coord(u, mode) * (float)Wt - 0.5f;
    const float y = prep_coord(v, mode) * (float)Ht - 0.5f;
    const float x0f = floorf(x), y0f = floorf(y);
    Taps t;
    t.fx = x - x0f;
    t.fy = y - y0f;
    const int x0 = (int)x0f, y0 = (int)y0f;
    t.valid = 0xFu;
    if (mode == FPCDR_BOUNDARY_ZERO) {      // (a NaN or huge coordinate converts to INT_MIN / INT_MAX: every tap outside)
        const bool vx0 = x0 >= 0 && x0 < Wt, vx1 = x0 >= -1 && x0 < Wt - 1, vy0 = y0 >= 0 && y0 < Ht, vy1 = y0 >= -1 && y0 < Ht - 1;
        t.valid = (vx0 && vy0 ? 1u : 0u) | (vx1 && vy0 ? 2u : 0u) | (vx0 && vy1 ? 4u : 0u) | (vx1 && vy1 ? 8u : 0u);
    }
    const int ix0 = wrap_near(x0, Wt, mode), ix1 = wrap_near(x0 == 0x7fffffff ? x0 : x0 + 1, Wt, mode);
    const int iy0 = wrap_near(y0, Ht, mode), iy1 = wrap_near(y0 == 0x7fffffff ? y0 : y0 + 1, Ht, mode);
    t.i00 = (iy0 * Wt + ix0) * C; t.i10 = (iy0 * Wt + ix1) * C;
    t.i01 = (iy1 * Wt + ix0) * C; t.i11 = (iy1 * Wt + ix1) * C;
    return t;
}

// make_taps for 'wrap' / 'clamp' without its branches (the fused kernels; ~15 vector instructions for the four indices instead of
// ~45 and four branches).  The prepared coordinate lies in [0, 1], so x0 = floor(x) lies in [-1, n - 1] and x0 + 1 in [0, n]: the
// first index wraps only below 0, the second only at n.  Same indices and fractions as make_taps for every finite coordinate; a NaN or
// infinite one (x0 = INT_MIN / INT_MAX after the conversion) ends in a valid index through the final clamp.
__device__ __forceinline__ int clamp_idx(int i, int n) { return min(max(i, 0), n - 1); }
__device__ __forceinline__ Taps make_taps_fast(float u, float v, int Ht, int Wt, int C, int mode) {
    const float x = prep_coord(u, mode) * (float)Wt - 0.5f;
    const float y = prep_coord(v, mode) * (float)Ht - 0.5f;
    const float x0f = floorf(x), y0f = floorf(y);
    Taps t;
    t.fx = x - x0f;
    t.fy = y - y0f;
    const int x0 = (int)x0f, y0 = (int)y0f;
    t.valid = 0xFu;
    int ix0, ix1, iy0, iy1;
    if (mode == FPCDR_BOUNDARY_WRAP) {
        ix0 = clamp_idx(x0 + (x0 < 0 ? Wt : 0), Wt);
        iy0 = clamp_idx(y0 + (y0 < 0 ? Ht : 0), Ht);
        ix1 = clamp_idx(x0 >= Wt - 1 ? x0 + 1 - Wt : x0 + 1, Wt);      // (no overflow: x0 = INT_MAX takes the first arm)
        iy1 = clamp_idx(y0 >= Ht - 1 ? y0 + 1 - Ht : y0 + 1, Ht);
    } else {
        ix0 = clamp_idx(x0, Wt); iy0 = clamp_idx(y0, Ht);
        ix1 = clamp_idx(x0 >= Wt - 1 ? Wt - 1 : x0 + 1, Wt);
        iy1 = clamp_idx(y0 >= Ht - 1 ? Ht - 1 : y0 + 1, Ht);
    }
    t.i00 = (iy0 * Wt + ix0) * C; t.i10 = (iy0 * Wt + ix1) * C;
    t.i01 = (iy1 * Wt + ix0) * C; t.i11 = (iy1 * Wt + ix1) * C;
    return t;
}

// two horizontally adjacent texels in one 8-byte gather (4-byte aligned): the gather rate, not HBM, bounds
// the texture kernels, so halving the number of gather instructions matters
typedef float float2_u __attribute__((ext_vector_type(2), aligned(4)));

// OFF32: the texel offsets go through 32-bit byte offsets from the (uniform) texture base (common.h ld32; the caller guarantees
// Ht * Wt * C < 2^30 values)
template <bool OFF32 = false>
__device__ __forceinline__ void load_taps(const float *tx, const Taps &t, int c, int C, float &t00, float &t10, float &t01,
                                          float &t11) {
    if (C == 1 && t.i10 == t.i00 + 1) {   // no wrap between the two columns
        float2_u va, vb;
        if (OFF32) {
            va = *reinterpret_cast<const float2_u *>(reinterpret_cast<const char *>(tx) + ((unsigned int)t.i00 << 2));
            vb = *reinterpret_cast<const float2_u *>(reinterpret_cast<const char *>(tx) + ((unsigned int)t.i01 << 2));
        } else {
            va = *reinterpret_cast<const float2_u *>(tx + t.i00);
            vb = *reinterpret_cast<const float2_u *>(tx + t.i01);
        }
        t00 = va.x; t10 = va.y; t01 = vb.x; t11 = vb.y;
    } else if (OFF32) {
        t00 = ld32(tx, (unsigned int)(t.i00 + c)); t10 = ld32(tx, (unsigned int)(t.i10 + c));
        t01 = ld32(tx, (unsigned int)(t.i01 + c)); t11 = ld32(tx, (unsigned int)(t.i11 + c));
    } else {
        t00 = tx[t.i00 + c]; t10 = tx[t.i10 + c]; t01 = tx[t.i01 + c]; t11 = tx[t.i11 + c];
    }
}

// boundary mode 'zero': taps outside the texture read as 0
__device__ __forceinline__ void mask_taps(const Taps &t, float &t00, float &t10, float &t01, float &t11) {
    if (t.valid != 0xFu) {
        if (!(t.valid & 1u)) t00 = 0.0f;
        if (!(t.valid & 2u)) t10 = 0.0f;
        if (!(t.valid & 4u)) t01 = 0.0f;
        if (!(t.valid & 8u)) t11 = 0.0f;
    }
}

template <bool OFF32 = false>
__device__ __forceinline__ float bilerp(const float *tx, const Taps &t, int c, int C) {
    float t00, t10, t01, t11;
    load_taps<OFF32>(tx, t, c, C, t00, t10, t01, t11);
    mask_taps(t, t00, t10, t01, t11);
    const float top = t00 + (t10 - t00) * t.fx;
    const float bot = t01 + (t11 - t01) * t.fx;
    return top + (bot - top) * t.fy;
}

// ---- mip-mapped sampling ('linear-mipmap-nearest' / 'linear-mipmap-linear'), shared by texture.hip and the fused objective kernels ----
struct TexLevels {
    const float *tex[FPCDR_MAX_MIP + 1];
    float *grad[FPCDR_MAX_MIP + 1];
};

// level of detail from the uv footprint; returns the unclamped level, outputs the pieces the backward needs
struct Lod { float level, l2, rt, df, bq, dudx, dudy, dvdx, dvdy; };

__device__ __forceinline__ Lod compute_lod(float4 d, int Ht, int Wt, float bias) {
    Lod L;
    L.dudx = d.x * (float)Wt; L.dudy = d.y * (float)Wt; L.dvdx = d.z * (float)Ht; L.dvdy = d.w * (float)Ht;
    const float A = L.dudx * L.dudx + L.dudy * L.dudy;
    const float Bq = L.dudx * L.dvdx + L.dudy * L.dvdy;
    const float Cc = L.dvdx * L.dvdx + L.dvdy * L.dvdy;
    const float tr = 0.5f * (A + Cc);
    L.df = 0.5f * (A - Cc);
    L.bq = Bq;
    L.rt = sqrtf(L.df * L.df + Bq * Bq + 1e-30f);
    L.l2 = tr + L.rt;
    L.level = 0.5f * log2f(fmaxf(L.l2, 1e-30f)) + bias;
    return L;
}

// the four taps of one level: hand dy * weight to scatter(tap 0..3 = (00, 10, 01, 11), element offset, channel, value) for every tap
// that lies in the texture (boundary mode 'zero': the padding receives no gradient) and return (d out / d fx, d out / d fy) summed over
// channels
template <typename Scatter>
__device__ __forceinline__ void taps_bwd_to(const float *tx, bool want_tex, const Taps &t, const float *g, float scale, int C,
                                            float &gfx, float &gfy, Scatter &&scatter) {
    const float w00 = (1.0f - t.fx) * (1.0f - t.fy), w10 = t.fx * (1.0f - t.fy), w01 = (1.0f - t.fx) * t.fy, w11 = t.fx * t.fy;
    for (int c = 0; c < C; ++c) {
        const float gc = g[c] * scale;
        float t00, t10, t01, t11;
        load_taps(tx, t, c, C, t00, t10, t01, t11);
        mask_taps(t, t00, t10, t01, t11);
        gfx += gc * ((t10 - t00) * (1.0f - t.fy) + (t11 - t01) * t.fy);
        gfy += gc * ((t01 + (t11 - t01) * t.fx) - (t00 + (t10 - t00) * t.fx));
        if (want_tex && gc != 0.0f) {
            if (t.valid & 1u) scatter(0, t.i00, c, gc * w00);
            if (t.valid & 2u) scatter(1, t.i10, c, gc * w10);
            if (t.valid & 4u) scatter(2, t.i01, c, gc * w01);
            if (t.valid & 8u) scatter(3, t.i11, c, gc * w11);
        }
    }
}
// ... with global atomics into gtx (null: no texel gradient wanted)
__device__ __forceinline__ void taps_bwd(const float *tx, float *gtx, const Taps &t, const float *g, float scale, int C,
                                         float &gfx, float &gfy) {
    taps_bwd_to(tx, gtx != nullptr, t, g, scale, C, gfx, gfy, [&](int, int off, int c, float v) { atomicAdd(gtx + off + c, v); });
}

// One pixel of a mip-mapped lookup at texture coordinate q: level from the footprint da (has_da) plus bias, clamped to
// [0, n_levels]; trilinear: the two neighbouring levels blended by the level's fraction, else the nearest level.  b = texture
// image (0 for a single texture).  emit(c, value) for every channel.
template <typename Emit>
__device__ __forceinline__ void mip_sample_fwd(const TexLevels &lv, size_t b, int n_levels, float2 q, bool has_da, float4 da, float bias,
                                               int Ht, int Wt, int C, bool trilinear, int boundary, Emit &&emit) {
    float level = bias;
    if (has_da) level = compute_lod(da, Ht, Wt, level).level;
    level = fminf(fmaxf(level, 0.0f), (float)n_levels);
    int l0;
    float fl = 0.0f;
    if (!trilinear) {
        l0 = min((int)floorf(level + 0.5f), n_levels);
    } else {
        l0 = min((int)floorf(level), n_levels);
        fl = level - (float)l0;
    }
    const int h0 = Ht >> l0, w0 = Wt >> l0;
    const Taps t0 = make_taps(q.x, q.y, h0, w0, C, boundary);
    const float *tx0 = lv.tex[l0] + b * h0 * w0 * C;
    if (!trilinear) {
        for (int c = 0; c < C; ++c) emit(c, bilerp(tx0, t0, c, C));
    } else {
        const int l1 = min(l0 + 1, n_levels);
        const int h1 = Ht >> l1, w1 = Wt >> l1;
        const Taps t1 = make_taps(q.x, q.y, h1, w1, C, boundary);
        const float *tx1 = lv.tex[l1] + b * h1 * w1 * C;
        for (int c = 0; c < C; ++c) {
            const float c0 = bilerp(tx0, t0, c, C), c1 = bilerp(tx1, t1, c, C);
            emit(c, c0 + (c1 - c0) * fl);
        }
    }
}

// Backward of mip_sample_fwd for the incoming gradient g[C]: scatters into lv.grad[level] (where non-null) and returns the gradient
// of the texture coordinate (gu, gv: the caller applies the clamp-mode mask), of the footprint (gda, has_da) and of the bias.
// scatter(level, tap, element offset inside the level's image, channel, value) receives the texel gradients.
template <typename Scatter>
__device__ __forceinline__ void mip_sample_bwd_to(const TexLevels &lv, size_t b, int n_levels, float2 q, bool has_da, float4 da, float bias,
                                                  int Ht, int Wt, int C, bool trilinear, int boundary, const float *g, float &gu, float &gv,
                                                  float4 &gda, float &gbias, Scatter &&scatter) {
    float raw = bias;
    Lod L;
    if (has_da) { L = compute_lod(da, Ht, Wt, raw); raw = L.level; }
    const float level = fminf(fmaxf(raw, 0.0f), (float)n_levels);
    int l0;
    float fl = 0.0f;
    if (!trilinear) {
        l0 = min((int)floorf(level + 0.5f), n_levels);
    } else {
        l0 = min((int)floorf(level), n_levels);
        fl = level - (float)l0;
    }
    const int h0 = Ht >> l0, w0 = Wt >> l0;
    const Taps t0 = make_taps(q.x, q.y, h0, w0, C, boundary);
    const size_t img0 = b * h0 * w0 * C;
    float gfx = 0.f, gfy = 0.f;
    taps_bwd_to(lv.tex[l0] + img0, lv.grad[l0] != nullptr, t0, g, 1.0f - fl, C, gfx, gfy,
                [&](int tap, int off, int c, float v) { scatter(l0, tap, img0 + off, c, v); });
    gu = gfx * (float)w0; gv = gfy * (float)h0;
    if (trilinear) {
        const int l1 = min(l0 + 1, n_levels);
        const int h1 = Ht >> l1, w1 = Wt >> l1;
        const Taps t1 = make_taps(q.x, q.y, h1, w1, C, boundary);
        const size_t img1 = b * h1 * w1 * C;
        float gfx1 = 0.f, gfy1 = 0.f;
        taps_bwd_to(lv.tex[l1] + img1, lv.grad[l1] != nullptr, t1, g, fl, C, gfx1, gfy1,
                    [&](int tap, int off, int c, float v) { scatter(l1, tap, img1 + off, c, v); });
        gu += gfx1 * (float)w1;
        gv += gfy1 * (float)h1;
        // d out / d fl = sum_c g_c (c1 - c0);  level clamp passes gradient inside [0, n_levels]
        float gfl = 0.f;
        for (int c = 0; c < C; ++c) gfl += g[c] * (bilerp(lv.tex[l1] + img1, t1, c, C) - bilerp(lv.tex[l0] + img0, t0, c, C));
        const float glevel = (raw >= 0.0f && raw <= (float)n_levels) ? gfl : 0.0f;
        gbias = glevel;
        if (has_da) {
            // level = 0.5 log2(max(l2, eps)) + bias ; l2 = tr + rt ; rt = sqrt(df^2 + bq^2 + eps)
            const float gl2 = (L.l2 >= 1e-30f) ? glevel * 0.5f / (L.l2 * 0.6931471805599453f) : 0.0f;
            const float gdf = gl2 * L.df / L.rt, gbq = gl2 * L.bq / L.rt;
            const float gA = 0.5f * gl2 + 0.5f * gdf, gC = 0.5f * gl2 - 0.5f * gdf;
            const float g_dudx = 2.0f * L.dudx * gA + L.dvdx * gbq;
            const float g_dudy = 2.0f * L.dudy * gA + L.dvdy * gbq;
            const float g_dvdx = 2.0f * L.dvdx * gC + L.dudx * gbq;
            const float g_dvdy = 2.0f * L.dvdy * gC + L.dudy * gbq;
            gda = make_float4(g_dudx * (float)Wt, g_dudy * (float)Wt, g_dvdx * (float)Ht, g_dvdy * (float)Ht);
        }
    }
}

// ... with global atomics into lv.grad[level]
__device__ __forceinline__ void mip_sample_bwd(const TexLevels &lv, size_t b, int n_levels, float2 q, bool has_da, float4 da, float bias,
                                               int Ht, int Wt, int C, bool trilinear, int boundary, const float *g, float &gu, float &gv,
                                               float4 &gda, float &gbias) {
    mip_sample_bwd_to(lv, b, n_levels, q, has_da, da, bias, Ht, Wt, C, trilinear, boundary, g, gu, gv, gda, gbias,
                      [&](int level, int, size_t off, int c, float v) { atomicAdd(lv.grad[level] + off + c, v); });
}

// ---- one trilinear lookup, forward AND backward (the one-pass objective's MIP instantiation, objective.hip) -------------------------
// mip_sample_fwd + mip_sample_bwd_to build four tap sets and load every texel three times (forward, tap gradients, the level
// fraction's gradient).  A pixel that is shaded and chained back by the same thread keeps the level arithmetic, the two tap sets and
// the eight texels per channel instead: same formulas in the same order, so the values are those of the two generic routines.
template <int CS>
struct MipKeep {
    Lod L;
    float raw, fl;
    int l0, l1;
    Taps t0, t1;
    float x0[CS][4], x1[CS][4];      // texels (00, 10, 01, 11) of level l0 / l1, masked for boundary mode 'zero'
    float c0[CS], c1[CS];            // bilinear values of the two levels
};

template <int CS>
__device__ __forceinline__ void mip_lookup_fwd(const TexLevels &lv, int n_levels, float2 q, float4 da, int Ht, int Wt, int boundary,
                                               MipKeep<CS> &K, float (&col)[CS]) {
    K.L = compute_lod(da, Ht, Wt, 0.0f);
    K.raw = K.L.level;
    const float level = fminf(fmaxf(K.raw, 0.0f), (float)n_levels);
    K.l0 = min((int)floorf(level), n_levels);
    K.fl = level - (float)K.l0;
    K.l1 = min(K.l0 + 1, n_levels);
    const int h0 = Ht >> K.l0, w0 = Wt >> K.l0, h1 = Ht >> K.l1, w1 = Wt >> K.l1;
    K.t0 = boundary == FPCDR_BOUNDARY_ZERO ? make_taps(q.x, q.y, h0, w0, CS, boundary) : make_taps_fast(q.x, q.y, h0, w0, CS, boundary);
    K.t1 = boundary == FPCDR_BOUNDARY_ZERO ? make_taps(q.x, q.y, h1, w1, CS, boundary) : make_taps_fast(q.x, q.y, h1, w1, CS, boundary);
    const float *tx0 = lv.tex[K.l0], *tx1 = lv.tex[K.l1];
#pragma unroll
    for (int c = 0; c < CS; ++c) {
        load_taps<true>(tx0, K.t0, c, CS, K.x0[c][0], K.x0[c][1], K.x0[c][2], K.x0[c][3]);
        load_taps<true>(tx1, K.t1, c, CS, K.x1[c][0], K.x1[c][1], K.x1[c][2], K.x1[c][3]);
    }
#pragma unroll
    for (int c = 0; c < CS; ++c) {
        mask_taps(K.t0, K.x0[c][0], K.x0[c][1], K.x0[c][2], K.x0[c][3]);
        mask_taps(K.t1, K.x1[c][0], K.x1[c][1], K.x1[c][2], K.x1[c][3]);
        const float top0 = K.x0[c][0] + (K.x0[c][1] - K.x0[c][0]) * K.t0.fx, bot0 = K.x0[c][2] + (K.x0[c][3] - K.x0[c][2]) * K.t0.fx;
        const float top1 = K.x1[c][0] + (K.x1[c][1] - K.x1[c][0]) * K.t1.fx, bot1 = K.x1[c][2] + (K.x1[c][3] - K.x1[c][2]) * K.t1.fx;
        K.c0[c] = top0 + (bot0 - top0) * K.t0.fy;
        K.c1[c] = top1 + (bot1 - top1) * K.t1.fy;
        col[c] = K.c0[c] + (K.c1[c] - K.c0[c]) * K.fl;
    }
}

// scatter(level, tap 0..3, element offset inside the level's image, channel, value); gu, gv: gradient of the texture coordinate (the
// caller applies the clamp-mode mask); gda: gradient of the footprint
template <int CS, typename Scatter>
__device__ __forceinline__ void mip_lookup_bwd(const TexLevels &lv, int n_levels, const MipKeep<CS> &K, const float (&g)[CS], int Ht, int Wt,
                                               float &gu, float &gv, float4 &gda, Scatter &&scatter) {
    const int w0 = Wt >> K.l0, h0 = Ht >> K.l0, w1 = Wt >> K.l1, h1 = Ht >> K.l1;
    auto level_bwd = [&](int l, const Taps &t, const float (&x)[CS][4], float scale, float &gfx, float &gfy) {
        const float w00 = (1.0f - t.fx) * (1.0f - t.fy), w10 = t.fx * (1.0f - t.fy), w01 = (1.0f - t.fx) * t.fy, w11 = t.fx * t.fy;
        const bool want = lv.grad[l] != nullptr;
#pragma unroll
        for (int c = 0; c < CS; ++c) {
            const float gc = g[c] * scale;
            gfx += gc * ((x[c][1] - x[c][0]) * (1.0f - t.fy) + (x[c][3] - x[c][2]) * t.fy);
            gfy += gc * ((x[c][2] + (x[c][3] - x[c][2]) * t.fx) - (x[c][0] + (x[c][1] - x[c][0]) * t.fx));
            if (want && gc != 0.0f) {
                if (t.valid & 1u) scatter(l, 0, (size_t)t.i00, c, gc * w00);
                if (t.valid & 2u) scatter(l, 1, (size_t)t.i10, c, gc * w10);
                if (t.valid & 4u) scatter(l, 2, (size_t)t.i01, c, gc * w01);
                if (t.valid & 8u) scatter(l, 3, (size_t)t.i11, c, gc * w11);
            }
        }
    };
    float gfx0 = 0.f, gfy0 = 0.f, gfx1 = 0.f, gfy1 = 0.f;
    level_bwd(K.l0, K.t0, K.x0, 1.0f - K.fl, gfx0, gfy0);
    level_bwd(K.l1, K.t1, K.x1, K.fl, gfx1, gfy1);
    gu = gfx0 * (float)w0; gv = gfy0 * (float)h0;
    gu += gfx1 * (float)w1;
    gv += gfy1 * (float)h1;
    float gfl = 0.f;
#pragma unroll
    for (int c = 0; c < CS; ++c) gfl += g[c] * (K.c1[c] - K.c0[c]);
    const float glevel = (K.raw >= 0.0f && K.raw <= (float)n_levels) ? gfl : 0.0f;
    const Lod &L = K.L;
    const float gl2 = (L.l2 >= 1e-30f) ? glevel * 0.5f / (L.l2 * 0.6931471805599453f) : 0.0f;
    const float gdf = gl2 * L.df / L.rt, gbq = gl2 * L.bq / L.rt;
    const float gA = 0.5f * gl2 + 0.5f * gdf, gC = 0.5f * gl2 - 0.5f * gdf;
    const float g_dudx = 2.0f * L.dudx * gA + L.dvdx * gbq;
    const float g_dudy = 2.0f * L.dudy * gA + L.dvdy * gbq;
    const float g_dvdx = 2.0f * L.dvdx * gC + L.dudx * gbq;
    const float g_dvdy = 2.0f * L.dvdy * gC + L.dudy * gbq;
    gda = make_float4(g_dudx * (float)Wt, g_dudy * (float)Wt, g_dvdx * (float)Ht, g_dvdy * (float)Ht);
}
