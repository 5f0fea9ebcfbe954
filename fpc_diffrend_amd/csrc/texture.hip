// texture forward / backward for gfx950 (MI355X).
//
// Performs the work of `dr.texture(tex[None], texc, filter_mode='linear')` (reference
// src/torch/fit.py:158) and `dr.texture(tex[None], texc, texd, filter_mode='linear-mipmap-linear',
// max_mip_level=n)` (fit.py:155); nvdiffrast op, absent from the reference tree.
//
// One pixel per lane.  The texture (reference: 1024 x 1024 x 1 = 4 MB, fit.py:439) and its mip chain
// stay resident in L2 / Infinity Cache, so the forward pass streams uv in and colour out at HBM rate.
// The backward pass is bounded by the f32 atomic rate into grad_tex, not by HBM: pixels whose
// incoming gradient is exactly zero (every background pixel of the fit loop, since fit.py:161
// overwrites them with a constant) issue no atomics at all.
#include "common.h"

namespace {

#include "texsample.h"

constexpr int TROWS = 8;   // rows per thread of the forward kernel (divides the 32-row hint bin)

// the texture's value at uv = (0,0) (what every empty pixel of the fit loop samples): out [C]
__global__ void k_tex_empty(TexLevels lv, int Ht, int Wt, int C, int filter, int boundary, float *__restrict__ out) {
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        if (filter == FPCDR_FILTER_NEAREST) {
            const float x = prep_coord(0.0f, boundary) * (float)Wt - 0.5f, y = prep_coord(0.0f, boundary) * (float)Ht - 0.5f;
            const int rx = (int)floorf(x + 0.5f), ry = (int)floorf(y + 0.5f);
            const int ix = wrap_i(rx, Wt, boundary), iy = wrap_i(ry, Ht, boundary);
            const bool in = boundary != FPCDR_BOUNDARY_ZERO || (rx >= 0 && rx < Wt && ry >= 0 && ry < Ht);
            out[c] = in ? lv.tex[0][((size_t)iy * Wt + ix) * C + c] : 0.0f;
        } else {
            const Taps t = make_taps(0.0f, 0.0f, Ht, Wt, C, boundary);
            out[c] = bilerp(lv.tex[0], t, c, C);
        }
    }
}

__global__ void __launch_bounds__(256) k_tex_fwd(TexLevels lv, int n_levels, const float2 *__restrict__ uv,
                                                 const float4 *__restrict__ uv_da, const float *__restrict__ bias,
                                                 int H, int W, int B, int Bt, int Ht, int Wt, int C, int filter,
                                                 int boundary, float *__restrict__ out, const uint8_t *__restrict__ hint,
                                                 const float *__restrict__ empty_color) {
    // forward: 256 x TROWS pixels per workgroup, grid (W / 256, H / TROWS, B); a thread owns TROWS vertically adjacent pixels and
    // has all their uv loads in flight at once (see k_interp_fwd); the rows lie in one 32-row hint bin
    const int px = blockIdx.x * 256 + threadIdx.x, y0 = blockIdx.y * TROWS;
    if (px >= W) return;
    const bool known_empty = hint && !fpcdr_hint_on(hint, 0, B, H, W, blockIdx.z, y0, px);
    if (known_empty) {
        // region hint: uv = (0,0) in an empty bin -- the texture's value there, without reading uv
        for (int k = 0; k < TROWS && y0 + k < H; ++k) {
            float *o = out + (((long long)blockIdx.z * H + y0 + k) * W + px) * C;
            for (int c = 0; c < C; ++c) o[c] = empty_color[c];
        }
        return;
    }
    float2 qq[TROWS];
#pragma unroll
    for (int k = 0; k < TROWS; ++k)
        qq[k] = y0 + k < H ? uv[((long long)blockIdx.z * H + y0 + k) * W + px] : make_float2(0.f, 0.f);
#pragma unroll
    for (int k = 0; k < TROWS; ++k) {
        if (y0 + k >= H) break;
        const long long i = ((long long)blockIdx.z * H + y0 + k) * W + px;
        float *o = out + i * C;
        const float2 q = qq[k];
        const int b = Bt > 1 ? (int)blockIdx.z : 0;
        if (filter == FPCDR_FILTER_NEAREST) {
            const float x = prep_coord(q.x, boundary) * (float)Wt - 0.5f, y = prep_coord(q.y, boundary) * (float)Ht - 0.5f;
            const int rx = (int)floorf(x + 0.5f), ry = (int)floorf(y + 0.5f);
            const int ix = wrap_i(rx, Wt, boundary), iy = wrap_i(ry, Ht, boundary);
            const bool in = boundary != FPCDR_BOUNDARY_ZERO || (rx >= 0 && rx < Wt && ry >= 0 && ry < Ht);
            const float *tx = lv.tex[0] + (size_t)b * Ht * Wt * C + ((size_t)iy * Wt + ix) * C;
            for (int c = 0; c < C; ++c) o[c] = in ? tx[c] : 0.0f;
        } else if (filter == FPCDR_FILTER_LINEAR) {
            const Taps t = make_taps(q.x, q.y, Ht, Wt, C, boundary);
            const float *tx = lv.tex[0] + (size_t)b * Ht * Wt * C;
            for (int c = 0; c < C; ++c) o[c] = bilerp(tx, t, c, C);
        } else {
            const float4 da = uv_da ? uv_da[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            mip_sample_fwd(lv, (size_t)b, n_levels, q, uv_da != nullptr, da, bias ? bias[i] : 0.0f, Ht, Wt, C,
                           filter == FPCDR_FILTER_LINEAR_MIPMAP_LINEAR, boundary, [&](int c, float v) { o[c] = v; });
        }
    }
}

// Bin-shaped fast path for the reference's texture call (fit.py:158: one 1-channel texture, 'linear', W % 4 == 0): one
// workgroup per 32 x 32-pixel bin, four horizontally adjacent pixels per thread, one 16-byte store per lane (see
// k_interp_fwd_bin2 in interpolate.hip); a bin the region hint calls empty gets the texture's value at uv = (0,0) unread.
__global__ void __launch_bounds__(256) k_tex_fwd_bin1(const float *__restrict__ tex, const float4 *__restrict__ uv4, int H, int W, int B,
                                                      int Ht, int Wt, int boundary, float4 *__restrict__ out4,
                                                      const uint8_t *__restrict__ hint, const float *__restrict__ empty_color) {
    const int tid = threadIdx.x, b = blockIdx.z;
    const int px = blockIdx.x * 32 + (tid & 7) * 4, py = blockIdx.y * 32 + (tid >> 3);
    if (px >= W || py >= H) return;
    const size_t i = ((size_t)b * H + py) * W + px;
    if (hint && !fpcdr_hint_on(hint, 0, B, H, W, b, py, px)) {
        const float e = empty_color[0];
        out4[i / 4] = make_float4(e, e, e, e);
        return;
    }
    const float4 a = uv4[i / 2], c = uv4[i / 2 + 1];
    float4 o;
    o.x = bilerp(tex, make_taps(a.x, a.y, Ht, Wt, 1, boundary), 0, 1);
    o.y = bilerp(tex, make_taps(a.z, a.w, Ht, Wt, 1, boundary), 0, 1);
    o.z = bilerp(tex, make_taps(c.x, c.y, Ht, Wt, 1, boundary), 0, 1);
    o.w = bilerp(tex, make_taps(c.z, c.w, Ht, Wt, 1, boundary), 0, 1);
    out4[i / 4] = o;
}

__global__ void __launch_bounds__(256) k_tex_bwd(TexLevels lv, int n_levels, const float2 *__restrict__ uv,
                                                 const float4 *__restrict__ uv_da, const float *__restrict__ bias,
                                                 const float *__restrict__ dy, int H, int W, int B, int Bt, int Ht,
                                                 int Wt, int C, int filter, int boundary, float2 *__restrict__ grad_uv,
                                                 float4 *__restrict__ grad_uv_da, float *__restrict__ grad_bias) {
    // backward: one wave = a 16 x 4 pixel tile (block 16 x 16) so that the lanes' atomics land on neighbouring
    // texels of a few texture rows
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int px = blockIdx.x * 16 + (lane & 15), py = blockIdx.y * 16 + wave * 4 + (lane >> 4);
    if (px < W && py < H) {
        const long long i = ((long long)blockIdx.z * H + py) * W + px;
        const float *g = dy + i * C;
        bool any = false;
        for (int c = 0; c < C; ++c) any |= (g[c] != 0.0f);
        float2 guv = make_float2(0.f, 0.f);
        float4 gda = make_float4(0.f, 0.f, 0.f, 0.f);
        float gbias = 0.f;
        if (any) {
            const float2 q = uv[i];
            const int b = Bt > 1 ? (int)blockIdx.z : 0;
            // clamp mode: no uv gradient outside [0,1] (torch.clamp semantics of the oracle)
            const float mu = (boundary == FPCDR_BOUNDARY_CLAMP && !(q.x >= 0.0f && q.x <= 1.0f)) ? 0.0f : 1.0f;
            const float mv = (boundary == FPCDR_BOUNDARY_CLAMP && !(q.y >= 0.0f && q.y <= 1.0f)) ? 0.0f : 1.0f;
            if (filter == FPCDR_FILTER_NEAREST) {
                if (lv.grad[0]) {
                    const float x = prep_coord(q.x, boundary) * (float)Wt - 0.5f, y = prep_coord(q.y, boundary) * (float)Ht - 0.5f;
                    const int rx = (int)floorf(x + 0.5f), ry = (int)floorf(y + 0.5f);
                    const int ix = wrap_i(rx, Wt, boundary), iy = wrap_i(ry, Ht, boundary);
                    const bool in = boundary != FPCDR_BOUNDARY_ZERO || (rx >= 0 && rx < Wt && ry >= 0 && ry < Ht);
                    float *gt = lv.grad[0] + (size_t)b * Ht * Wt * C + ((size_t)iy * Wt + ix) * C;
                    for (int c = 0; c < C; ++c)
                        if (in && g[c] != 0.0f) atomicAdd(gt + c, g[c]);
                }
            } else if (filter == FPCDR_FILTER_LINEAR) {
                const Taps t = make_taps(q.x, q.y, Ht, Wt, C, boundary);
                const size_t img = (size_t)b * Ht * Wt * C;
                float gfx = 0.f, gfy = 0.f;
                taps_bwd(lv.tex[0] + img, lv.grad[0] ? lv.grad[0] + img : nullptr, t, g, 1.0f, C, gfx, gfy);
                guv = make_float2(gfx * (float)Wt * mu, gfy * (float)Ht * mv);
            } else {
                const float4 da = uv_da ? uv_da[i] : make_float4(0.f, 0.f, 0.f, 0.f);
                float gu, gv;
                mip_sample_bwd(lv, (size_t)b, n_levels, q, uv_da != nullptr, da, bias ? bias[i] : 0.0f, Ht, Wt, C,
                               filter == FPCDR_FILTER_LINEAR_MIPMAP_LINEAR, boundary, g, gu, gv, gda, gbias);
                guv = make_float2(gu * mu, gv * mv);
            }
        }
        if (grad_uv) grad_uv[i] = guv;
        if (grad_uv_da) grad_uv_da[i] = gda;
        if (grad_bias) grad_bias[i] = gbias;
    }
}

// Bin-shaped backward for the reference's texture call (one 1-channel texture, 'linear', W % 4 == 0): one workgroup per
// 32 x 32-pixel bin, four horizontally adjacent pixels per thread (16-byte loads of dy / uv, 16-byte stores of grad_uv).  The
// texel gradient goes through an LDS window of DOUBLES shaped by the bin's footprint (below; ds_add_f64: ds_add_f32 costs
// 3 cycles per lane on gfx950) and is flushed once, row-contiguously; taps outside the window (uv seams) and the empty
// pixels' share (uv = (0,0): summed per workgroup) go to memory directly.  4 global atomics per covered pixel -- the
// generic kernel -- bound the operator at 4.3 ms for cfg3.
// r6: the window is a RECTANGLE of at most TWC cells shaped by the bounding box of the bin's taps (as k_shade's, objective.hip): a
// footprint is rarely square -- the rig's face is sampled at 1.6 texels per pixel along v and 1.0 along u, 53 x 34 texels under a bin --
// and the fixed 40 x 40 window anchored at the smallest tap left 22 % of the pixels' adds outside, four float atomics each to memory.
// A box that does not fit (a bin across the seam of a periodic coordinate) gets a window of its aspect around its centre.
constexpr int TWC = 2048;
__global__ void __launch_bounds__(256) k_tex_bwd_bin1(const float *__restrict__ tex, float *__restrict__ grad_tex,
                                                      const float4 *__restrict__ uv4, const float4 *__restrict__ dy4, int H, int W, int B,
                                                      int Ht, int Wt, int boundary, float4 *__restrict__ grad_uv4,
                                                      const uint8_t *__restrict__ hint) {
    __shared__ double s_tex[TWC];      // in double: ds_add_f64 (common.h lds_add_f64)
    __shared__ int s_org[4];           // min x0, min y0, max x0, max y0 of the bin's taps
    __shared__ float s_esum;
    const int tid = threadIdx.x, lane = tid & 63, b = blockIdx.z;
    const int px = blockIdx.x * 32 + (tid & 7) * 4, py = blockIdx.y * 32 + (tid >> 3);
    const bool inside = px < W && py < H;
    const size_t i = ((size_t)b * H + (inside ? py : 0)) * W + (inside ? px : 0);
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 g = inside ? dy4[i / 4] : z4;
    const bool any = g.x != 0.f || g.y != 0.f || g.z != 0.f || g.w != 0.f;
    if (!__builtin_amdgcn_readfirstlane(__syncthreads_or(any ? 1 : 0))) {      // no gradient arrives in this bin
        if (inside && grad_uv4) { grad_uv4[i / 2] = z4; grad_uv4[i / 2 + 1] = z4; }
        return;
    }
    // region hint: uv = (0,0) in an empty bin -- not read
    const bool known_zero = hint && !fpcdr_hint_on(hint, 0, B, H, W, b, blockIdx.y * 32, blockIdx.x * 32);
    float4 ua = z4, ub = z4;
    if (inside && any && !known_zero) { ua = uv4[i / 2]; ub = uv4[i / 2 + 1]; }
    const float qu[4] = {ua.x, ua.z, ub.x, ub.z}, qv[4] = {ua.y, ua.w, ub.y, ub.w}, gq[4] = {g.x, g.y, g.z, g.w};
    if (tid == 0) { s_org[0] = 0x7fffffff; s_org[1] = 0x7fffffff; s_org[2] = (int)0x80000000; s_org[3] = (int)0x80000000; s_esum = 0.0f; }
    for (int k = tid; k < TWC; k += 256) s_tex[k] = 0.0;
    // the bounding box of the taps of the pixels with a gradient and a texture coordinate other than (0,0)
    int ux0 = 0x7fffffff, uy0 = 0x7fffffff, ux1 = (int)0x80000000, uy1 = (int)0x80000000;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (inside && gq[k] != 0.0f && (qu[k] != 0.0f || qv[k] != 0.0f)) {
            const int tx = (int)floorf(prep_coord(qu[k], boundary) * (float)Wt - 0.5f), ty = (int)floorf(prep_coord(qv[k], boundary) * (float)Ht - 0.5f);
            ux0 = min(ux0, tx); ux1 = max(ux1, tx);
            uy0 = min(uy0, ty); uy1 = max(uy1, ty);
        }
    __syncthreads();
    {
        int mx = ux0, my = uy0, nx = ~ux1, ny = ~uy1;
        wave_min4_dpp_lane63(mx, my, nx, ny);      // (results in lane 63)
        if (lane == 63 && mx != 0x7fffffff) { atomicMin(&s_org[0], mx); atomicMin(&s_org[1], my); atomicMax(&s_org[2], ~nx); atomicMax(&s_org[3], ~ny); }
    }
    __syncthreads();
    int ox = __builtin_amdgcn_readfirstlane(s_org[0]), oy = __builtin_amdgcn_readfirstlane(s_org[1]);
    int stride = 1, rows = 1;      // (stride 1: no window -- every add outside)
    if (ox != 0x7fffffff) {      // (uniform)
        const long long nw = (long long)__builtin_amdgcn_readfirstlane(s_org[2]) - ox + 2, nh = (long long)__builtin_amdgcn_readfirstlane(s_org[3]) - oy + 2;      // (taps x0 and x0 + 1)
        long long sx0 = ox, sy0 = oy;
        if (nw * nh <= TWC) {      // it fits: the rows that are left over go half below, half above
            stride = (int)nw;
            rows = TWC / stride;
            sy0 -= (rows - (int)nh) >> 1;
        } else {                   // it does not: a window of the footprint's aspect around its centre
            const float aspect = fminf(fmaxf((float)nw / (float)nh, 1.0f / (float)TWC), (float)TWC);
            stride = min(max((int)sqrtf((float)TWC * aspect), 2), TWC / 2);
            rows = TWC / stride;
            sx0 += (nw - stride) >> 1;
            sy0 += (nh - rows) >> 1;
        }
        ox = __builtin_amdgcn_readfirstlane((int)sx0); oy = __builtin_amdgcn_readfirstlane((int)sy0);
        stride = __builtin_amdgcn_readfirstlane(stride); rows = __builtin_amdgcn_readfirstlane(rows);
    }
    float gu[4] = {0.f, 0.f, 0.f, 0.f}, gv[4] = {0.f, 0.f, 0.f, 0.f}, esum = 0.0f;
    // The eight lanes of a row own pixels 4 j .. 4 j + 3.  Visiting them in the same order would put the lanes of one LDS
    // instruction 4 texels = 8 words apart: with the eight rows of the wave 64 lanes on 4 double-wide banks.  Each lane
    // therefore starts at a different pixel of its four (neighbouring lanes end up 5 texels apart: distinct banks); the
    // thread's values are picked / put back with selects, there is no register indexing.
    const int rot = tid & 3;
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
        const int k = (s4 + rot) & 3;
        const bool k0 = k == 0, k1 = k == 1, k2 = k == 2;
        const float gc = k0 ? gq[0] : (k1 ? gq[1] : (k2 ? gq[2] : gq[3]));
        if (!inside || gc == 0.0f) continue;
        const float qu_k = k0 ? qu[0] : (k1 ? qu[1] : (k2 ? qu[2] : qu[3]));
        const float qv_k = k0 ? qv[0] : (k1 ? qv[1] : (k2 ? qv[2] : qv[3]));
        const Taps t = make_taps(qu_k, qv_k, Ht, Wt, 1, boundary);
        float t00, t10, t01, t11;
        load_taps(tex, t, 0, 1, t00, t10, t01, t11);
        const float gfx = gc * ((t10 - t00) * (1.0f - t.fy) + (t11 - t01) * t.fy);
        const float gfy = gc * ((t01 + (t11 - t01) * t.fx) - (t00 + (t10 - t00) * t.fx));
        // clamp mode: no uv gradient outside [0,1] (torch.clamp semantics of the oracle)
        const float mu = (boundary == FPCDR_BOUNDARY_CLAMP && !(qu_k >= 0.0f && qu_k <= 1.0f)) ? 0.0f : 1.0f;
        const float mv = (boundary == FPCDR_BOUNDARY_CLAMP && !(qv_k >= 0.0f && qv_k <= 1.0f)) ? 0.0f : 1.0f;
        const float gu_k = gfx * (float)Wt * mu, gv_k = gfy * (float)Ht * mv;
        gu[0] = k0 ? gu_k : gu[0]; gu[1] = k1 ? gu_k : gu[1]; gu[2] = k2 ? gu_k : gu[2]; gu[3] = (k0 | k1 | k2) ? gu[3] : gu_k;
        gv[0] = k0 ? gv_k : gv[0]; gv[1] = k1 ? gv_k : gv[1]; gv[2] = k2 ? gv_k : gv[2]; gv[3] = (k0 | k1 | k2) ? gv[3] : gv_k;
        if (!grad_tex) continue;
        if (qu_k == 0.0f && qv_k == 0.0f) { esum += gc; continue; }     // the four texels at uv = (0,0): once per workgroup
        const float w00 = (1.0f - t.fx) * (1.0f - t.fy), w10 = t.fx * (1.0f - t.fy), w01 = (1.0f - t.fx) * t.fy, w11 = t.fx * t.fy;
        const int lx = (int)floorf(prep_coord(qu_k, boundary) * (float)Wt - 0.5f) - ox;
        const int ly = (int)floorf(prep_coord(qv_k, boundary) * (float)Ht - 0.5f) - oy;
        if ((unsigned int)lx < (unsigned int)(stride - 1) && (unsigned int)ly < (unsigned int)(rows - 1)) {
            double *w = s_tex + ly * stride + lx;
            lds_add_f64(w, gc * w00);
            lds_add_f64(w + 1, gc * w10);
            lds_add_f64(w + stride, gc * w01);
            lds_add_f64(w + stride + 1, gc * w11);
        } else {
            atomicAdd(grad_tex + t.i00, gc * w00);
            atomicAdd(grad_tex + t.i10, gc * w10);
            atomicAdd(grad_tex + t.i01, gc * w01);
            atomicAdd(grad_tex + t.i11, gc * w11);
        }
    }
    if (inside && grad_uv4) {
        grad_uv4[i / 2] = make_float4(gu[0], gv[0], gu[1], gv[1]);
        grad_uv4[i / 2 + 1] = make_float4(gu[2], gv[2], gu[3], gv[3]);
    }
    if (!grad_tex) return;
    esum = wave_sum_dpp(esum);
    if (lane == 0 && esum != 0.0f) lds_add_f32(&s_esum, esum);
    __syncthreads();
    if (tid == 0 && s_esum != 0.0f) {
        const Taps t0 = make_taps(0.0f, 0.0f, Ht, Wt, 1, boundary);
        const float e = s_esum;
        atomicAdd(grad_tex + t0.i00, e * ((1.0f - t0.fx) * (1.0f - t0.fy)));
        atomicAdd(grad_tex + t0.i10, e * (t0.fx * (1.0f - t0.fy)));
        atomicAdd(grad_tex + t0.i01, e * ((1.0f - t0.fx) * t0.fy));
        atomicAdd(grad_tex + t0.i11, e * (t0.fx * t0.fy));
    }
    if (stride > 1) {
        const float inv = 1.0f / (float)stride;
        const int n = stride * rows;
        for (int k = tid; k < n; k += 256) {
            const float v = (float)s_tex[k];
            if (v != 0.0f) {
                const int ly = (int)(((float)k + 0.5f) * inv), lx = k - ly * stride;      // (exact: k < 2^12)
                const int gx = wrap_near(ox + lx, Wt, boundary), gy = wrap_near(oy + ly, Ht, boundary);
                atomicAdd(grad_tex + (size_t)gy * Wt + gx, v);
            }
        }
    }
}

__global__ void __launch_bounds__(256) k_mip_down(const float *__restrict__ src, float *__restrict__ dst, int N, int Ht,
                                                  int Wt, int C) {
    const int ho = Ht / 2, wo = Wt / 2;
    const long long total = (long long)N * ho * wo * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        long long r = i / C;
        const int x = (int)(r % wo); r /= wo;
        const int y = (int)(r % ho);
        const int n = (int)(r / ho);
        const float *s = src + (((size_t)n * Ht + 2 * y) * Wt + 2 * x) * C + c;
        dst[i] = (s[0] + s[C] + s[(size_t)Wt * C] + s[(size_t)Wt * C + C]) * 0.25f;
    }
}

__global__ void __launch_bounds__(256) k_mip_down_bwd(const float *__restrict__ gdst, float *__restrict__ gsrc, int N, int Ht,
                                                      int Wt, int C) {
    const int ho = Ht / 2, wo = Wt / 2;
    const long long total = (long long)N * Ht * Wt * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        long long r = i / C;
        const int x = (int)(r % Wt); r /= Wt;
        const int y = (int)(r % Ht);
        const int n = (int)(r / Ht);
        gsrc[i] += 0.25f * gdst[(((size_t)n * ho + y / 2) * wo + x / 2) * C + c];
    }
}

int grid_for(long long total) {
    long long g = (total + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 16384 ? 16384 : g));
}

int check_common(int B, int H, int W, int Bt, int Ht, int Wt, int C, int n_levels, int filter, int boundary) {
    if (!(B > 0 && H > 0 && W > 0 && Ht > 0 && Wt > 0 && C > 0)) return 1;
    if (!(Bt == 1 || Bt == B)) return 2;
    if (n_levels < 0 || n_levels > FPCDR_MAX_MIP) return 3;
    if (filter < FPCDR_FILTER_NEAREST || filter > FPCDR_FILTER_LINEAR_MIPMAP_LINEAR) return 4;
    if (boundary != FPCDR_BOUNDARY_WRAP && boundary != FPCDR_BOUNDARY_CLAMP && boundary != FPCDR_BOUNDARY_ZERO) return 5;
    if ((long long)Ht * Wt * C > 0x7fffffffLL) return 6;
    for (int l = 0; l < n_levels; ++l)
        if (((Ht >> l) & 1) || ((Wt >> l) & 1)) return 7;
    return 0;
}

}  // namespace

extern "C" int fpcdr_mip_downsample(const float *src, float *dst, int32_t N, int32_t Ht, int32_t Wt, int32_t C, void *stream) {
    FPCDR_REQUIRE(src && dst, "null pointer");
    FPCDR_REQUIRE(N > 0 && Ht >= 2 && Wt >= 2 && C > 0 && !(Ht & 1) && !(Wt & 1), "level must have even, positive size");
    hipLaunchKernelGGL(k_mip_down, dim3(grid_for((long long)N * (Ht / 2) * (Wt / 2) * C)), dim3(256), 0, (hipStream_t)stream, src,
                       dst, N, Ht, Wt, C);
    FPCDR_CHECK_LAUNCH();
    return FPCDR_OK;
}

extern "C" int fpcdr_mip_downsample_bwd(const float *grad_dst, float *grad_src, int32_t N, int32_t Ht, int32_t Wt, int32_t C,
                                        void *stream) {
    FPCDR_REQUIRE(grad_dst && grad_src, "null pointer");
    FPCDR_REQUIRE(N > 0 && Ht >= 2 && Wt >= 2 && C > 0 && !(Ht & 1) && !(Wt & 1), "level must have even, positive size");
    hipLaunchKernelGGL(k_mip_down_bwd, dim3(grid_for((long long)N * Ht * Wt * C)), dim3(256), 0, (hipStream_t)stream, grad_dst,
                       grad_src, N, Ht, Wt, C);
    FPCDR_CHECK_LAUNCH();
    return FPCDR_OK;
}

extern "C" int fpcdr_texture_fwd(const fpcdr_texture_fwd_params *p, void *stream) {
    FPCDR_REQUIRE(p != nullptr, "null params");
    FPCDR_REQUIRE(p->tex[0] && p->uv && p->out, "null pointer");
    int rc = check_common(p->B, p->H, p->W, p->Bt, p->Ht, p->Wt, p->C, p->n_levels, p->filter_mode, p->boundary_mode);
    FPCDR_REQUIRE(rc == 0, "bad sizes / modes");
    const bool mip = p->filter_mode >= FPCDR_FILTER_LINEAR_MIPMAP_NEAREST;
    FPCDR_REQUIRE(mip || p->n_levels == 0, "n_levels must be 0 for non-mip filters");
    TexLevels lv;
    for (int l = 0; l <= FPCDR_MAX_MIP; ++l) {
        lv.tex[l] = l <= p->n_levels ? p->tex[l] : nullptr;
        lv.grad[l] = nullptr;
        if (l <= p->n_levels) FPCDR_REQUIRE(p->tex[l] != nullptr, "missing mip level");
    }
    FPCDR_REQUIRE(p->B <= 65535 && p->H <= 65535, "image batch / height too large for one launch");
    const uint8_t *hint = nullptr;
    if (p->hint) {
        FPCDR_REQUIRE(!mip && p->Bt == 1 && p->empty_color, "a region hint needs filter nearest / linear, one texture and empty_color");
        hint = p->hint;
        hipLaunchKernelGGL(k_tex_empty, dim3(1), dim3(64), 0, (hipStream_t)stream, lv, p->Ht, p->Wt, p->C, p->filter_mode, p->boundary_mode,
                           p->empty_color);
    }
    if (p->filter_mode == FPCDR_FILTER_LINEAR && p->C == 1 && p->Bt == 1 && (p->W & 3) == 0 && (((size_t)p->uv | (size_t)p->out) & 15) == 0) {
        hipLaunchKernelGGL(k_tex_fwd_bin1, dim3(fpcdr_cdiv(p->W, 32), fpcdr_cdiv(p->H, 32), p->B), dim3(256), 0, (hipStream_t)stream,
                           p->tex[0], (const float4 *)p->uv, p->H, p->W, p->B, p->Ht, p->Wt, p->boundary_mode, (float4 *)p->out, hint,
                           p->empty_color);
        FPCDR_CHECK_LAUNCH();
        return FPCDR_OK;
    }
    hipLaunchKernelGGL(k_tex_fwd, dim3(fpcdr_cdiv(p->W, 256), fpcdr_cdiv(p->H, TROWS), p->B), dim3(256), 0, (hipStream_t)stream, lv, p->n_levels,
                       (const float2 *)p->uv, mip ? (const float4 *)p->uv_da : nullptr, mip ? p->mip_level_bias : nullptr, p->H, p->W,
                       p->B, p->Bt, p->Ht, p->Wt, p->C, p->filter_mode, p->boundary_mode, p->out, hint, p->empty_color);
    FPCDR_CHECK_LAUNCH();
    return FPCDR_OK;
}

extern "C" int fpcdr_texture_bwd(const fpcdr_texture_bwd_params *p, void *stream) {
    FPCDR_REQUIRE(p != nullptr, "null params");
    FPCDR_REQUIRE(p->tex[0] && p->uv && p->dy, "null pointer");
    int rc = check_common(p->B, p->H, p->W, p->Bt, p->Ht, p->Wt, p->C, p->n_levels, p->filter_mode, p->boundary_mode);
    FPCDR_REQUIRE(rc == 0, "bad sizes / modes");
    const bool mip = p->filter_mode >= FPCDR_FILTER_LINEAR_MIPMAP_NEAREST;
    FPCDR_REQUIRE(mip || p->n_levels == 0, "n_levels must be 0 for non-mip filters");
    TexLevels lv;
    for (int l = 0; l <= FPCDR_MAX_MIP; ++l) {
        lv.tex[l] = l <= p->n_levels ? p->tex[l] : nullptr;
        lv.grad[l] = l <= p->n_levels ? p->grad_tex[l] : nullptr;
        if (l <= p->n_levels) FPCDR_REQUIRE(p->tex[l] != nullptr, "missing mip level");
    }
    FPCDR_REQUIRE(p->B <= 65535 && fpcdr_cdiv(p->H, 16) <= 65535, "image batch / height too large for one launch");
    if (p->filter_mode == FPCDR_FILTER_LINEAR && p->C == 1 && p->Bt == 1 && (p->W & 3) == 0 && p->boundary_mode != FPCDR_BOUNDARY_ZERO &&
        (((size_t)p->uv | (size_t)p->dy | (size_t)p->grad_uv) & 15) == 0) {
        hipLaunchKernelGGL(k_tex_bwd_bin1, dim3(fpcdr_cdiv(p->W, 32), fpcdr_cdiv(p->H, 32), p->B), dim3(256), 0, (hipStream_t)stream,
                           p->tex[0], p->grad_tex[0], (const float4 *)p->uv, (const float4 *)p->dy, p->H, p->W, p->B, p->Ht, p->Wt,
                           p->boundary_mode, (float4 *)p->grad_uv, p->hint);
        FPCDR_CHECK_LAUNCH();
        return FPCDR_OK;
    }
    dim3 grid(fpcdr_cdiv(p->W, 16), fpcdr_cdiv(p->H, 16), p->B);
    hipLaunchKernelGGL(k_tex_bwd, grid, dim3(256), 0, (hipStream_t)stream, lv, p->n_levels,
                       (const float2 *)p->uv, mip ? (const float4 *)p->uv_da : nullptr, mip ? p->mip_level_bias : nullptr, p->dy,
                       p->H, p->W, p->B, p->Bt, p->Ht, p->Wt, p->C, p->filter_mode, p->boundary_mode, (float2 *)p->grad_uv,
                       mip ? (float4 *)p->grad_uv_da : nullptr, mip ? p->grad_mip_level_bias : nullptr);
    FPCDR_CHECK_LAUNCH();
    return FPCDR_OK;
}
