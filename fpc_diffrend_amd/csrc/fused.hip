// Fused pixel objective of the fit loop for gfx950 (MI355X): everything between the clip-space positions and
// the scalar pixel loss of reference src/torch/fit.py:151-161 + :579, as three kernels instead of nine:
//
//   fpcdr_render_fwd (rasterize.hip)  rasterize + interpolate + texture      -> rast, colour        20 B/px
//   fpcdr_aa_loss_fwd (this file)     antialias + background + pixel loss    -> d loss / d aa, flags 25 B/px
//   fpcdr_render_aa_bwd (this file)   antialias bwd + texture bwd + interpolate bwd + rasterize bwd  20 B/px
//
// versus 217 B/px for the separate operators (C = 1).  The antialiased colour, the texture-coordinate image and
// every intermediate gradient image never reach HBM.  Values equal the separate operators' (same device
// functions); the separate operators remain the nvdiffrast-compatible API and the parity reference.
#include "common.h"

#include <algorithm>
#include <stdlib.h>

namespace {

#include "texsample.h"
#include "raster_math.h"
#include "aa_pairs.h"
#include "sil_bits.h"

// ------------------------------------------------------------------------------------------------
// Sparse mode: validity of pixels around a workgroup, from the occupancy map of fpcdr_render_fwd.  A pixel of an
// unoccupied 32x32 bin was never written: it is EMPTY (rast = 0, colour = empty_color, no gradient).  The window holds
// the 4 x 3 bins around (bin_x0, bin_y0): bit (dy + 1) * 4 + (dx + 1) for bin (bin_x0 + dx, bin_y0 + dy).
struct OccWin {
    unsigned int mask;
    int bin_x0, bin_y0;
    __device__ __forceinline__ bool bin(int dx, int dy) const { return (mask >> ((dy + 1) * 4 + dx + 1)) & 1u; }
    __device__ __forceinline__ bool pixel(int x, int y) const {
        return bin((x >> 5) - bin_x0, (y >> 5) - bin_y0);
    }
};
__device__ __forceinline__ OccWin load_occ(const uint16_t *occ, int b, int H, int W, int bin_x0, int bin_y0) {
    const int OX = FPCDR_OCC_DIM(W), OY = FPCDR_OCC_DIM(H);
    // one load, uniform over the workgroup -- and readfirstlane makes that visible to the compiler: every early exit that
    // depends on the window is then a scalar branch (the work-queue kernels rely on it, see k_aa_fix_queue)
    OccWin w = {(unsigned int)__builtin_amdgcn_readfirstlane((int)occ[((size_t)b * OY + bin_y0) * OX + bin_x0]), bin_x0, bin_y0};
    return w;
}

// ------------------------------------------------------------------------------------------------
// One pixel of antialias (gather form, see antialias.hip) + background composite + squared error: evaluates the
// pixel's four pairs (me-R, me-U owned; L-me, D-me not), writes d(sum of squares * grad_scale)/d(antialiased colour),
// sets the two ownership flags and returns the pixel's loss term (SPARSE: the difference to a background pixel).
template <int CS, bool SPARSE>
__device__ __forceinline__ float aa_loss_pixel(const AAGeom &g, const OccWin &ow, const float *__restrict__ color, size_t img, int x,
                                               int y, float2 me, float2 nR, float2 nL, float2 nU, float2 nD, bool hasR, bool hasL,
                                               bool hasU, bool hasD, bool v_me, const float (&ecol)[CS],
                                               const uint8_t *__restrict__ ref, float bg, float color_scale, float grad_scale,
                                               float *__restrict__ g_aa, bool &fx_flag, bool &fy_flag) {
    const int W = g.W;
    const size_t off = img + (size_t)y * W + x;
    const int id = (int)me.y;
    float lsum = 0.0f;
    const bool disc = ((int)nR.y != id) | ((int)nL.y != id) | ((int)nU.y != id) | ((int)nD.y != id);
    const bool covered = id > 0;
    if (disc | covered) {
        float acc[CS], cme[CS];
#pragma unroll
        for (int c = 0; c < CS; ++c) { cme[c] = (!SPARSE || v_me) ? color[off * CS + c] : ecol[c]; acc[c] = cme[c]; }
        if (disc) {
            auto visit = [&](int x0, int y0, int d, float2 p0, float2 p1, bool own, bool &flag) {
                if ((int)p0.y == (int)p1.y) return;
                bool hit = for_active_edges(g, x0, y0, d, (int)p0.y, p0.x, (int)p1.y, p1.x,
                    [&](float t, int Px, int Py, int Qx, int Qy, int, int, const EdgeEval &, float) {
                        const bool far = t >= 0.5f;
                        const int rx = far ? Qx : Px, ry = far ? Qy : Py;
                        if (rx != x || ry != y) return;
                        const int ox = far ? Px : Qx, oy = far ? Py : Qy;
                        const float amt = far ? t - 0.5f : 0.5f - t;
                        const bool ovalid = !SPARSE || ow.pixel(ox, oy);
                        const float *co = color + (img + (size_t)oy * W + ox) * CS;
#pragma unroll
                        for (int c = 0; c < CS; ++c) acc[c] += amt * ((ovalid ? co[c] : ecol[c]) - cme[c]);
                    });
                if (own && hit) flag = true;
            };
            bool dummy = false;
            if (hasR) visit(x, y, 0, me, nR, true, fx_flag);
            if (hasU) visit(x, y, 1, me, nU, true, fy_flag);
            if (hasL) visit(x - 1, y, 0, nL, me, false, dummy);
            if (hasD) visit(x, y - 1, 1, nD, me, false, dummy);
        }
        if (covered) {   // background elsewhere (fit.py:161): no gradient, and in sparse mode no loss term either
            const float rf = (float)ref[off];
            const float d0 = rf - bg * color_scale;
#pragma unroll
            for (int c = 0; c < CS; ++c) {
                const float d = rf - acc[c] * color_scale;
                lsum += SPARSE ? (d * d - d0 * d0) : d * d;
                g_aa[off * CS + c] = (-2.0f * color_scale * grad_scale) * d;
            }
        }
    }
    if (!covered) {
        if (!SPARSE) {
            const float d0 = (float)ref[off] - bg * color_scale;
            lsum += (float)CS * d0 * d0;
        }
        if (!SPARSE || v_me) {
#pragma unroll
            for (int c = 0; c < CS; ++c) g_aa[off * CS + c] = 0.0f;
        }
    }
    return lsum;
}

// ------------------------------------------------------------------------------------------------
// antialias forward (gather form, see antialias.hip) + background composite + squared error.  One workgroup per
// 64 x 32 pixel block (two bins), one wave per 8 rows, one pixel per lane and row; the rows above / below travel in
// registers from one row to the next.  Writes d(sum of squares * grad_scale)/d(antialiased colour) and the two flag bit
// planes; reduces the loss.  SPARSE: blocks without an occupied bin (or an occupied right / upper neighbour, whose
// pairs a block owns) leave at once, and the loss is accumulated as the difference to an all-background image.
// latency-bound: 8 waves per SIMD (1.54 -> 1.34 ms)
#ifndef FPCDR_AAL_WPE
#define FPCDR_AAL_WPE __attribute__((amdgpu_waves_per_eu(8, 8)))
#endif
template <int CS, bool SPARSE>
__global__ void __launch_bounds__(256) FPCDR_AAL_WPE k_aa_loss(const float *__restrict__ color, const float4 *__restrict__ rast,
                                                 const float4 *__restrict__ pos, const int32_t *__restrict__ tri,
                                                 const uint8_t *__restrict__ sil, const uint8_t *__restrict__ ref, int B, int H,
                                                 int W, int V, int T, float bg, float color_scale, float grad_scale,
                                                 unsigned long long *__restrict__ flags, float *__restrict__ g_aa,
                                                 const uint16_t *__restrict__ occ, const float *__restrict__ empty_color,
                                                 double *__restrict__ loss_sum) {
    __shared__ float s_part[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = blockIdx.x * 64 + lane, b = blockIdx.z;
    const int band = blockIdx.y;            // 32-row band = bin row
    const int Wq = FPCDR_AA_ROW_WORDS(W);
    OccWin ow = {0xfffu, 2 * (int)blockIdx.x, band};
    float ecol[CS];
#pragma unroll
    for (int c = 0; c < CS; ++c) ecol[c] = 0.0f;
    if (SPARSE) {
        ow = load_occ(occ, b, H, W, 2 * blockIdx.x, band);
        // the block's two bins, their right neighbours (x pairs are owned by the left pixel) and upper neighbours (y pairs)
        if (!(ow.bin(0, 0) || ow.bin(1, 0) || ow.bin(2, 0) || ow.bin(0, 1) || ow.bin(1, 1))) return;
#pragma unroll
        for (int c = 0; c < CS; ++c) ecol[c] = empty_color[c];
    }
    const size_t img = (size_t)b * H * W;
    const bool v_me = ow.bin(lane >> 5, 0);
    auto load_row = [&](int yy) -> float2 {   // (z, id) of pixel (x, yy); empty outside the image / in an unwritten bin
        if (x >= W || yy < 0 || yy >= H) return make_float2(0.f, 0.f);
        if (SPARSE && !ow.pixel(x, yy)) return make_float2(0.f, 0.f);
        return load_zid(rast, img + (size_t)yy * W + x);
    };
    const int ybase = band * 32 + wave * 8;
    float lsum = 0.0f;
    float2 dn = load_row(ybase - 1), cur = load_row(ybase);
    for (int r = 0; r < 8; ++r) {
        const int y = ybase + r;
        if (y >= H) break;
        const float2 up = load_row(y + 1);
        const float2 me = cur;
        bool fx_flag = false, fy_flag = false;
        // every lane of the wave (also beyond W) takes part in the neighbour exchange
        const float zr = __shfl_down(me.x, 1, 64), ir = __shfl_down(me.y, 1, 64);
        const float zl = __shfl_up(me.x, 1, 64), il = __shfl_up(me.y, 1, 64);
        if (x < W) {
            const bool hasR = x + 1 < W, hasL = x > 0, hasU = y + 1 < H, hasD = y > 0;
            // left / right neighbours come from the neighbouring lanes' registers; only the two ends of the wave's
            // 64-pixel span are loaded
            const size_t off = img + (size_t)y * W + x;
            float2 nR = me, nL = me;
            if (hasR) nR = lane < 63 ? make_float2(zr, ir) : ((!SPARSE || ow.bin(2, 0)) ? load_zid(rast, off + 1) : make_float2(0.f, 0.f));
            if (hasL) nL = lane > 0 ? make_float2(zl, il) : ((!SPARSE || ow.bin(-1, 0)) ? load_zid(rast, off - 1) : make_float2(0.f, 0.f));
            const float2 nU = hasU ? up : me;
            const float2 nD = hasD ? dn : me;
            const AAGeom g = {pos + (size_t)b * V, tri, sil + (size_t)b * T, T, W, H, 0.5f * (float)W, 0.5f * (float)H};
            lsum += aa_loss_pixel<CS, SPARSE>(g, ow, color, img, x, y, me, nR, nL, nU, nD, hasR, hasL, hasU, hasD, v_me, ecol, ref, bg,
                                              color_scale, grad_scale, g_aa, fx_flag, fy_flag);
        }
        const unsigned long long bx = __ballot(fx_flag), by = __ballot(fy_flag);
        if (lane == 0) {
            const size_t plane = (size_t)B * H * Wq;
            const size_t wi = ((size_t)b * H + y) * Wq + blockIdx.x;
            flags[wi] = bx;
            flags[plane + wi] = by;
        }
        dn = cur;
        cur = up;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) lsum += __shfl_xor(lsum, o, 64);
    if (lane == 0) s_part[wave] = lsum;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double tot = (double)s_part[0] + (double)s_part[1] + (double)s_part[2] + (double)s_part[3];
        const unsigned int slot = (blockIdx.x + 31u * blockIdx.y + 977u * blockIdx.z) % FPCDR_LOSS_SLOTS;
        if (tot != 0.0) atomicAdd(loss_sum + slot, tot);
    }
}

// ------------------------------------------------------------------------------------------------
// Backward of the whole pixel objective, one workgroup per 32x32-pixel bin, four pixels per thread:
//   g_c = g_aa + antialias corrections (gather form, only where a flag bit says a pair was blended)
//   texture bwd -> grad_tex;  interpolate bwd;  rasterize bwd -> grad_pos;  + the antialias op's own d alpha / d pos.
//
// Scattered global f32 atomics retire at only ~15-70 G lanes/s chip-wide (r1: 51 ms for this kernel when every pixel
// scattered), so everything is summed on chip first:
//   vertices  lanes are ordered so that pixels of one triangle sit next to each other (two adjacent 32-pixel rows per
//             wave, the second one reversed); ONE segmented DPP scan sums the nine components of every run of equal
//             triangle in the wave (common.h wave_segment_reduce; the per-triangle loop it replaces took 1.1 ms of
//             3.8), the run tails add into a 256-slot LDS table keyed by vertex, one flush per workgroup;
//   texels    a TEXW x TEXH LDS window anchored at the bin's smallest tap absorbs the four taps of every covered
//             pixel and is flushed row-contiguously; taps outside it (uv seams; empty pixels, which sample
//             uv = (0,0)) go straight to global memory.
// (One lane per triangle walking its bounding box -- the scheme of the forward rasteriser -- was tried here and took
// 6.9 ms: lanes reach their pixels at different trips, so the wave pays the full shading chain on almost every trip.)
// Window capacity: measured at cfg3 (~1 texel per pixel, a 32x32 bin spans ~36 texels): 64 -> 3.18 ms, 56 -> 2.72, 48 -> 2.67,
// 40 -> 2.59, 36 -> 2.55, 32 -> 2.49, 28 -> 2.62, 24 -> 2.98, 16 -> 3.80 (taps outside go straight to memory).  40 keeps
// headroom for denser textures; override with -DFPCDR_TEXWIN=n.
#ifndef FPCDR_TEXWIN
#define FPCDR_TEXWIN 40
#endif
#ifndef FPCDR_VSLOTS
#define FPCDR_VSLOTS 256
#endif
constexpr int VSLOTS = FPCDR_VSLOTS;
constexpr int TEXW = FPCDR_TEXWIN, TEXH = FPCDR_TEXWIN;
constexpr int BBIN = 32;

// latency-bound (dependent loads per bin): 8 waves per SIMD with a few spilled registers beat 5 without (3.26 -> 2.81 ms)
#ifndef FPCDR_BWD_WPE
#define FPCDR_BWD_WPE __attribute__((amdgpu_waves_per_eu(8, 8)))
#endif
#ifndef FPCDR_BWD_NT
#define FPCDR_BWD_NT 256
#endif
constexpr int BWD_NT = FPCDR_BWD_NT;          // threads per bin workgroup (measured: 128 -> 3.08 ms, 256 -> 2.61, 512 -> 3.28)
constexpr int BWD_NPX = BBIN * BBIN / BWD_NT;  // pixels per thread

// texture coordinates of triangle t through the index buffer (callers that did not pre-gather uv[uv_tri]); out of line: merged with
// the pre-gathered branch, its loads would drag 64-bit address arithmetic into the common path
__device__ __noinline__ void uv_indirect(const float2 *__restrict__ uv, const int32_t *__restrict__ uv_tri, int t, float2 &q0, float2 &q1,
                                         float2 &q2) {
    q0 = uv[uv_tri[3 * t]]; q1 = uv[uv_tri[3 * t + 1]]; q2 = uv[uv_tri[3 * t + 2]];
}

// mip levels of the MIP instantiation (the reference's enable_mip branch): level l = tex[l - 1] / grad[l - 1], level 0 = tex / grad_tex
struct MipArgs {
    const float *tex[FPCDR_MAX_MIP];
    float *grad[FPCDR_MAX_MIP];
    int n_levels;
};

// BMODE >= 0: the texture boundary mode as a compile-time constant.  MIP: the texture lookup was 'linear-mipmap-linear' with the
// footprint from the barycentrics' screen derivatives (bins_body<MIP>, rasterize.hip): texel gradients go to every level's buffer
// with global atomics (no LDS window), and the footprint's gradient flows through the derivative outputs of the rasteriser
// (shade_pixel_bwd<true>), as in the chain interpolate(diff_attrs='all') -> texture(uv_da) of the separate operators.
template <int CS, int BMODE = -1, bool MIP = false>
__device__ __forceinline__ void render_aa_bwd_body(const int b, const int bxi, const int byi,
                                                       const float4 *__restrict__ pos, const int32_t *__restrict__ tri,
                                                       const float2 *__restrict__ uv, const int32_t *__restrict__ uv_tri,
                                                       const float *__restrict__ tex, const float4 *__restrict__ rast,
                                                       const float *__restrict__ color, const float *__restrict__ g_aa,
                                                       const uint8_t *__restrict__ sil,
                                                       const unsigned long long *__restrict__ flags,
                                                       const uint16_t *__restrict__ occ, const float *__restrict__ empty_color,
                                                       int B, int V, int T, int H,
                                                       int W, int Ht, int Wt, int boundary_arg, float *__restrict__ grad_pos,
                                                       float *__restrict__ grad_tex, const float2 *__restrict__ tri_uv,
                                                       const float *__restrict__ upstream, const uint8_t *__restrict__ binflag,
                                                       const MipArgs *ma = nullptr) {
    const int boundary = BMODE >= 0 ? BMODE : boundary_arg;
    __shared__ int s_vkey[VSLOTS];
    __shared__ double s_vacc[VSLOTS][3];           // (x, y, w) sums per vertex slot, in double: ds_add_f64 (common.h lds_add_f64)
    __shared__ double s_tex[TEXH * TEXW * CS];
    __shared__ int s_org[2];            // smallest tap x, y of the bin (unwrapped texel coordinates)
    // MIP: a second window for level 1 of the chain (a bin's footprint there is a quarter of its level-0 one); taps of coarser
    // levels, and taps outside the windows, go to memory
    constexpr int TEX1 = 24;
    __shared__ double s_tex1[MIP ? TEX1 * TEX1 * CS : 1];
    __shared__ int s_org1[2];
    __shared__ float s_esum[CS];        // gradient arriving at EMPTY pixels' colour (they all sample uv = (0,0))
    __shared__ float s_fy[BBIN];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // pixel k of this thread: row 8 wave + 2 k + (lane >> 5) of the bin; the odd row runs right to left, so that lane 31
    // and lane 32 are vertical neighbours and a triangle's run continues from one row into the next
    const int col = (lane & 32) ? 63 - lane : lane;
    const int rowk0 = 2 * BWD_NPX * wave + (lane >> 5);
    const int bx0 = bxi * BBIN, by0 = byi * BBIN;
    const int x = bx0 + col;
    float *gp = grad_pos + (size_t)b * V * 4;
    OccWin ow = {0xfffu, bxi, byi};
    float ecol[CS];
#pragma unroll
    for (int c = 0; c < CS; ++c) ecol[c] = 0.0f;
    if (occ) {
        // sparse mode: pixels of unoccupied bins were never written and are empty.  An unoccupied bin still works if a
        // neighbour is occupied: it owns the pairs across its right / top edge, and its empty pixels receive colour
        // gradient (for the texel at uv = (0,0)) from pairs across any edge.
        ow = load_occ(occ, b, H, W, bxi, byi);
        if (!(ow.bin(0, 0) || ow.bin(1, 0) || ow.bin(0, 1) || ow.bin(-1, 0) || ow.bin(0, -1))) return;
#pragma unroll
        for (int c = 0; c < CS; ++c) ecol[c] = empty_color[c];
    }
    const bool v_me = ow.bin(0, 0);
    const size_t img = (size_t)b * H * W;
    // every gather of the hot path goes through a 32-bit byte offset from a wave-uniform base (common.h ld32): the bin's own pixels
    // at bin_off + r * W + col, its flag words at flag_base + r * Wq, per-image vertex / triangle arrays
    struct I3 { int a, b, c; };
    struct UV3 { float2 q0, q1, q2; };
    const size_t bin_off = img + (size_t)by0 * W + bx0;
    const int Wq = FPCDR_AA_ROW_WORDS(W);
    const size_t flag_plane = (size_t)B * H * Wq;
    const unsigned long long *const flags_bin = flags + ((size_t)b * H + by0) * Wq + (bx0 >> 6);
    const float4 *const pos_img = pos + (size_t)b * V;
    const float up = upstream ? upstream[0] : 1.0f;   // d(final loss)/d(this objective), a device scalar
    // flag words are loaded only where this bin, its left or its lower neighbour holds a blended pair (binflag: per-bin summary
    // written by k_aa_fix; null = unknown, load everywhere)
    bool flags_here = true;
    if (binflag) {
        const int OX = FPCDR_OCC_DIM(W), OY = FPCDR_OCC_DIM(H);
        const size_t bl = ((size_t)b * OY + byi) * OX + bxi;
        unsigned int f = binflag[bl];
        if (bxi > 0) f |= binflag[bl - 1];
        if (byi > 0) f |= binflag[bl - OX];
        flags_here = __builtin_amdgcn_readfirstlane((int)f) != 0;
    }
    if (tid == 0) { s_org[0] = 0x7fffffff; s_org[1] = 0x7fffffff; s_org1[0] = 0x7fffffff; s_org1[1] = 0x7fffffff; }
    if (tid < CS) s_esum[tid] = 0.0f;
    // NDC y of the bin's 32 rows: one IEEE division per row instead of one per pixel (and the column's fx once per thread, below)
    if (tid < BBIN) s_fy[tid] = (2.0f * (float)(by0 + tid) + 1.0f) / (float)H - 1.0f;
    // (both are read only behind the barriers below; in the work-queue form the pop's barriers separate one bin's reads from
    // the next bin's initialisation)

    // ---- pixel phase A: gradient arriving at each pixel's colour (antialias backward folded in) ----
    float go[BWD_NPX][CS];
    bool any[BWD_NPX];
#pragma unroll
    for (int k = 0; k < BWD_NPX; ++k) {
        const int y = by0 + rowk0 + 2 * k;
        any[k] = false;
#pragma unroll
        for (int c = 0; c < CS; ++c) go[k][c] = 0.0f;
        if (x < W && y < H) {
            const unsigned int rk = (unsigned int)(rowk0 + 2 * k);      // the pixel's row inside the bin
            const size_t plane = flag_plane;
            const size_t wi = ((size_t)b * H + y) * Wq + (x >> 6);
            const int bit = x & 63;
            bool own_x = false, own_y = false, left_x = false, down_y = false;
            if (flags_here) {
                const unsigned int wo = rk * (unsigned int)Wq;
                const unsigned long long fxw = ld32(flags_bin, wo), fyw = ld32(flags_bin + flag_plane, wo);
                own_x = (fxw >> bit) & 1ull; own_y = (fyw >> bit) & 1ull;
                left_x = bit > 0 ? ((fxw >> (bit - 1)) & 1ull) : (x > 0 ? ((flags[wi - 1] >> 63) & 1ull) : false);
                down_y = y > 0 ? ((flags[plane + wi - Wq] >> bit) & 1ull) : false;
            }
            const unsigned int poff = rk * (unsigned int)W + (unsigned int)col;
            const size_t off = bin_off + poff;
#pragma unroll
            for (int c = 0; c < CS; ++c) { go[k][c] = v_me ? ld32(g_aa + bin_off * CS, poff * CS + c) * up : 0.0f; any[k] |= (go[k][c] != 0.0f); }
            if (own_x | own_y | left_x | down_y) {
                // antialias backward for this pixel (see k_aa_bwd_fix in antialias.hip); sparse: plain global atomics
                AAGeom geo = {pos + (size_t)b * V, tri, sil + (size_t)b * T, T, W, H, 0.5f * (float)W, 0.5f * (float)H};
                const float2 me = v_me ? load_zid(rast, off) : make_float2(0.f, 0.f);
                auto zid_at = [&](int xx, int yy) -> float2 {
                    return ow.pixel(xx, yy) ? load_zid(rast, img + (size_t)yy * W + xx) : make_float2(0.f, 0.f);
                };
                auto visit = [&](int x0, int y0, int d, float2 p0, float2 p1, bool own) {
                    for_active_edges(geo, x0, y0, d, (int)p0.y, p0.x, (int)p1.y, p1.x,
                        [&](float t, int Px, int Py, int Qx, int Qy, int va, int vb, const EdgeEval &ev, float s) {
                            const bool far = t >= 0.5f;
                            const int rx = far ? Qx : Px, ry = far ? Qy : Py;
                            const float amt = far ? t - 0.5f : 0.5f - t;
                            if (!ow.pixel(rx, ry)) return;   // the blended pixel is an unwritten, empty one: no gradient arrives
                            float gr[CS];
#pragma unroll
                            for (int c = 0; c < CS; ++c) gr[c] = g_aa[(img + (size_t)ry * W + rx) * CS + c] * up;
                            if (rx == x && ry == y) {
#pragma unroll
                                for (int c = 0; c < CS; ++c) go[k][c] -= amt * gr[c];
                            } else {
#pragma unroll
                                for (int c = 0; c < CS; ++c) go[k][c] += amt * gr[c];
                            }
                            if (!own) return;
                            const bool vP = ow.pixel(Px, Py), vQ = ow.pixel(Qx, Qy);
                            const float *cP = color + (img + (size_t)Py * W + Px) * CS;
                            const float *cQ = color + (img + (size_t)Qy * W + Qx) * CS;
                            float G = 0.f;
#pragma unroll
                            for (int c = 0; c < CS; ++c) G += gr[c] * ((vP ? cP[c] : ecol[c]) - (vQ ? cQ[c] : ecol[c]));
                            if (G == 0.0f) return;
                            const float Ld = d == 0 ? ev.Lx : ev.Ly;
                            const float gLz = -G / (s * Ld);
                            const float gLd = -G * t / Ld;
                            const float gLx = d == 0 ? gLd : 0.0f, gLy = d == 0 ? 0.0f : gLd;
                            float g_qax = 0.f, g_qay = 0.f, g_wa = 0.f, g_qbx = 0.f, g_qby = 0.f, g_wb = 0.f;
                            g_qay += gLx * ev.wb; g_wb += gLx * ev.qay; g_wa -= gLx * ev.qby; g_qby -= gLx * ev.wa;
                            g_wa += gLy * ev.qbx; g_qbx += gLy * ev.wa; g_qax -= gLy * ev.wb; g_wb -= gLy * ev.qax;
                            g_qax += gLz * ev.qby; g_qby += gLz * ev.qax; g_qay -= gLz * ev.qbx; g_qbx -= gLz * ev.qay;
                            const float fxp = (float)Px + 0.5f - geo.hw, fyp = (float)Py + 0.5f - geo.hh;
                            atomicAdd(gp + 4 * (size_t)va + 0, g_qax * geo.hw);
                            atomicAdd(gp + 4 * (size_t)va + 1, g_qay * geo.hh);
                            atomicAdd(gp + 4 * (size_t)va + 3, g_wa - fxp * g_qax - fyp * g_qay);
                            atomicAdd(gp + 4 * (size_t)vb + 0, g_qbx * geo.hw);
                            atomicAdd(gp + 4 * (size_t)vb + 1, g_qby * geo.hh);
                            atomicAdd(gp + 4 * (size_t)vb + 3, g_wb - fxp * g_qbx - fyp * g_qby);
                        });
                };
                if (own_x) visit(x, y, 0, me, zid_at(x + 1, y), true);
                if (own_y) visit(x, y, 1, me, zid_at(x, y + 1), true);
                if (left_x) visit(x - 1, y, 0, zid_at(x - 1, y), me, false);
                if (down_y) visit(x, y - 1, 1, zid_at(x, y - 1), me, false);
                any[k] = false;
#pragma unroll
                for (int c = 0; c < CS; ++c) any[k] |= (go[k][c] != 0.0f);
            }
        }
    }
    // most bins of an image see no gradient at all: leave before touching the tables
    bool any_px = false;
#pragma unroll
    for (int k = 0; k < BWD_NPX; ++k) any_px |= any[k];
    if (!__builtin_amdgcn_readfirstlane(__syncthreads_or(any_px ? 1 : 0))) return;

    // ---- tables; texture coordinate of every pixel with a gradient; origin of the texel window ----
    for (int k = tid; k < VSLOTS; k += BWD_NT) {
        s_vkey[k] = -1;
        s_vacc[k][0] = 0.0; s_vacc[k][1] = 0.0; s_vacc[k][2] = 0.0;
    }
    for (int k = tid; k < TEXH * TEXW * CS; k += BWD_NT) s_tex[k] = 0.0;
    if (MIP) for (int k = tid; k < TEX1 * TEX1 * CS; k += BWD_NT) s_tex1[k] = 0.0;
    int pt[BWD_NPX];
    float tu[BWD_NPX], tv[BWD_NPX];
    {
        int ux0 = 0x7fffffff, uy0 = 0x7fffffff, vx0 = 0x7fffffff, vy0 = 0x7fffffff;
#pragma unroll
        for (int k = 0; k < BWD_NPX; ++k) {
            pt[k] = -1; tu[k] = 0.f; tv[k] = 0.f;
            if (any[k] && v_me) {
                const float4 r = ld32(rast + bin_off, (unsigned int)(rowk0 + 2 * k) * (unsigned int)W + (unsigned int)col);
                int t = (int)r.w - 1;
                if (t >= T) t = -1;
                if (t >= 0) {
                    float2 q0, q1, q2;
                    if (tri_uv) { const UV3 tq = ld32(reinterpret_cast<const UV3 *>(tri_uv), t); q0 = tq.q0; q1 = tq.q1; q2 = tq.q2; }
                    else uv_indirect(uv, uv_tri, t, q0, q1, q2);
                    const float w = 1.0f - r.x - r.y;
                    tu[k] = r.x * q0.x + r.y * q1.x + w * q2.x;
                    tv[k] = r.x * q0.y + r.y * q1.y + w * q2.y;
                    pt[k] = t;
                    if (grad_tex) {
                        ux0 = min(ux0, (int)floorf(prep_coord(tu[k], boundary) * (float)Wt - 0.5f));
                        uy0 = min(uy0, (int)floorf(prep_coord(tv[k], boundary) * (float)Ht - 0.5f));
                        if (MIP) {
                            vx0 = min(vx0, (int)floorf(prep_coord(tu[k], boundary) * (float)(Wt >> 1) - 0.5f));
                            vy0 = min(vy0, (int)floorf(prep_coord(tv[k], boundary) * (float)(Ht >> 1) - 0.5f));
                        }
                    }
                }
            }
        }
        const int mx = wave_min_dpp(ux0), my = wave_min_dpp(uy0);
        if (lane == 0 && mx != 0x7fffffff) { atomicMin(&s_org[0], mx); atomicMin(&s_org[1], my); }
        if (MIP) {
            const int m1x = wave_min_dpp(vx0), m1y = wave_min_dpp(vy0);
            if (lane == 0 && m1x != 0x7fffffff) { atomicMin(&s_org1[0], m1x); atomicMin(&s_org1[1], m1y); }
        }
    }
    __syncthreads();
    const int ox = s_org[0], oy = s_org[1];
    const int ox1 = s_org1[0], oy1 = s_org1[1];

    // ---- pixel phase B: texture backward; (dL/du, dL/dv) of the barycentrics into LDS; triangle set ----
    const float fx_col = (2.0f * (float)x + 1.0f) / (float)W - 1.0f;
    float esum[CS];
#pragma unroll
    for (int c = 0; c < CS; ++c) esum[c] = 0.0f;
#pragma unroll
    for (int k = 0; k < BWD_NPX; ++k) {
        int tkey = -1;
        float gu = 0.f, gvv = 0.f;
        int mip_vk[3] = {0, 0, 0};
        float mip_g9[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (any[k] && pt[k] < 0) {
            // an empty pixel sampled uv = (0,0): same four texels, same weights for every one of them.  Scattering each
            // with global atomics made thousands of rim pixels per image queue on four addresses (0.6 ms); they are
            // summed per workgroup instead and scattered once, below.
#pragma unroll
            for (int c = 0; c < CS; ++c) esum[c] += go[k][c];
        } else if (MIP && any[k]) {
            // ---- mip-mapped lookup: footprint from the rasteriser's derivatives (recomputed: rast_db is never stored), both levels'
            // texels and the level's fraction; the footprint's gradient goes back through the derivatives ----
            const int t = pt[k];
            float2 q0, q1, q2;
            if (tri_uv) { const UV3 tq = ld32(reinterpret_cast<const UV3 *>(tri_uv), t); q0 = tq.q0; q1 = tq.q1; q2 = tq.q2; }
            else uv_indirect(uv, uv_tri, t, q0, q1, q2);
            const I3 ti = ld32(reinterpret_cast<const I3 *>(tri), t);
            const float4 p0 = ld32(pos_img, ti.a), p1 = ld32(pos_img, ti.b), p2 = ld32(pos_img, ti.c);
            const float fx = fx_col, fy = s_fy[rowk0 + 2 * k];
            const float sx = 2.0f / (float)W, sy = 2.0f / (float)H;
            const Shade sd = shade_pixel(p0, p1, p2, fx, fy, sx, sy);
            const float e0x = q0.x - q2.x, e0y = q0.y - q2.y, e1x = q1.x - q2.x, e1y = q1.y - q2.y;
            const float4 da = make_float4(sd.dudx * e0x + sd.dvdx * e1x, sd.dudy * e0x + sd.dvdy * e1x,
                                          sd.dudx * e0y + sd.dvdx * e1y, sd.dudy * e0y + sd.dvdy * e1y);
            TexLevels lv;
            lv.tex[0] = tex; lv.grad[0] = grad_tex;
            for (int l = 1; l <= FPCDR_MAX_MIP; ++l) { lv.tex[l] = ma->tex[l - 1]; lv.grad[l] = ma->grad[l - 1]; }
            float gch[CS];
#pragma unroll
            for (int c = 0; c < CS; ++c) gch[c] = go[k][c];
            float gtu = 0.f, gtv = 0.f, gbias = 0.f;
            float4 gda = make_float4(0.f, 0.f, 0.f, 0.f);
            // texel gradients: level 0 and level 1 through their LDS windows (flushed once per bin), anything else straight to memory
            const int lx0 = (int)floorf(prep_coord(tu[k], boundary) * (float)Wt - 0.5f) - ox;
            const int ly0 = (int)floorf(prep_coord(tv[k], boundary) * (float)Ht - 0.5f) - oy;
            const bool in0 = lx0 >= 0 && ly0 >= 0 && lx0 + 1 < TEXW && ly0 + 1 < TEXH;
            const int lx1 = (int)floorf(prep_coord(tu[k], boundary) * (float)(Wt >> 1) - 0.5f) - ox1;
            const int ly1 = (int)floorf(prep_coord(tv[k], boundary) * (float)(Ht >> 1) - 0.5f) - oy1;
            const bool in1 = lx1 >= 0 && ly1 >= 0 && lx1 + 1 < TEX1 && ly1 + 1 < TEX1;
            mip_sample_bwd_to(lv, 0, ma->n_levels, make_float2(tu[k], tv[k]), true, da, 0.0f, Ht, Wt, CS, true, boundary, gch, gtu, gtv, gda, gbias,
                              [&](int level, int tap, size_t off, int c, float v) {
                                  const int dx = tap & 1, dy = tap >> 1;
                                  if (level == 0 && in0) lds_add_f64(&s_tex[((ly0 + dy) * TEXW + lx0 + dx) * CS + c], v);
                                  else if (level == 1 && in1) lds_add_f64(&s_tex1[((ly1 + dy) * TEX1 + lx1 + dx) * CS + c], v);
                                  else atomicAdd(lv.grad[level] + off + c, v);
                              });
            const float mu = (boundary == FPCDR_BOUNDARY_CLAMP && !(tu[k] >= 0.0f && tu[k] <= 1.0f)) ? 0.0f : 1.0f;
            const float mv = (boundary == FPCDR_BOUNDARY_CLAMP && !(tv[k] >= 0.0f && tv[k] <= 1.0f)) ? 0.0f : 1.0f;
            gtu *= mu; gtv *= mv;
            gu = gtu * e0x + gtv * e0y;
            gvv = gtu * e1x + gtv * e1y;
            // interpolate backward of the derivative outputs: d (uv_da) / d (rast_db)
            const float4 gdb = make_float4(gda.x * e0x + gda.z * e0y, gda.y * e0x + gda.w * e0y, gda.x * e1x + gda.z * e1y, gda.y * e1x + gda.w * e1y);
            if (gu != 0.0f || gvv != 0.0f || gdb.x != 0.0f || gdb.y != 0.0f || gdb.z != 0.0f || gdb.w != 0.0f) {
                tkey = t;
                mip_vk[0] = ti.a; mip_vk[1] = ti.b; mip_vk[2] = ti.c;
                float g0[3], g1[3], g2[3];
                shade_pixel_bwd<true>(p0, p1, p2, fx, fy, sx, sy, make_float4(gu, gvv, 0.f, 0.f), gdb, g0, g1, g2);
                mip_g9[0] = g0[0]; mip_g9[1] = g0[1]; mip_g9[2] = g0[2];
                mip_g9[3] = g1[0]; mip_g9[4] = g1[1]; mip_g9[5] = g1[2];
                mip_g9[6] = g2[0]; mip_g9[7] = g2[1]; mip_g9[8] = g2[2];
            }
        } else if (any[k]) {
            const int t = pt[k];
            // ('zero': the general tap routine with validity bits; a tap in the padding reads 0 and receives nothing -- its weight is
            //  zeroed, so the window cell or the clamped address it maps to gets + 0)
            const Taps tp = boundary == FPCDR_BOUNDARY_ZERO ? make_taps(tu[k], tv[k], Ht, Wt, CS, boundary)
                                                            : make_taps_fast(tu[k], tv[k], Ht, Wt, CS, boundary);
            const float w00 = (tp.valid & 1u) ? (1.0f - tp.fx) * (1.0f - tp.fy) : 0.0f, w10 = (tp.valid & 2u) ? tp.fx * (1.0f - tp.fy) : 0.0f;
            const float w01 = (tp.valid & 4u) ? (1.0f - tp.fx) * tp.fy : 0.0f, w11 = (tp.valid & 8u) ? tp.fx * tp.fy : 0.0f;
            bool in_win = false;
            int lx = 0, ly = 0;
            if (t >= 0 && grad_tex) {
                lx = (int)floorf(prep_coord(tu[k], boundary) * (float)Wt - 0.5f) - ox;
                ly = (int)floorf(prep_coord(tv[k], boundary) * (float)Ht - 0.5f) - oy;
                in_win = lx >= 0 && ly >= 0 && lx + 1 < TEXW && ly + 1 < TEXH;
            }
            float gfx = 0.f, gfy = 0.f;
#pragma unroll
            for (int c = 0; c < CS; ++c) {
                const float gc = go[k][c];
                float t00, t10, t01, t11;
                load_taps<true>(tex, tp, c, CS, t00, t10, t01, t11);      // (the entry point requires < 2^30 texel values)
                mask_taps(tp, t00, t10, t01, t11);
                gfx += gc * ((t10 - t00) * (1.0f - tp.fy) + (t11 - t01) * tp.fy);
                gfy += gc * ((t01 + (t11 - t01) * tp.fx) - (t00 + (t10 - t00) * tp.fx));
                if (grad_tex && gc != 0.0f) {
                    if (in_win) {
                        double *w = s_tex + (ly * TEXW + lx) * CS + c;
                        lds_add_f64(w, gc * w00);
                        lds_add_f64(w + CS, gc * w10);
                        lds_add_f64(w + TEXW * CS, gc * w01);
                        lds_add_f64(w + TEXW * CS + CS, gc * w11);
                    } else {
                        // (the offsets go through an opaque copy: shared with the texel loads above, the compiler forms ONE 64-bit address
                        //  per tap for both and the loads lose their scalar-base form -- 8 vector instructions per pixel)
                        unsigned int o00 = (unsigned int)(tp.i00 + c), o10 = (unsigned int)(tp.i10 + c), o01 = (unsigned int)(tp.i01 + c),
                                     o11 = (unsigned int)(tp.i11 + c);
                        asm volatile("" : "+v"(o00), "+v"(o10), "+v"(o01), "+v"(o11));
                        atomicAdd(&at32(grad_tex, o00), gc * w00);
                        atomicAdd(&at32(grad_tex, o10), gc * w10);
                        atomicAdd(&at32(grad_tex, o01), gc * w01);
                        atomicAdd(&at32(grad_tex, o11), gc * w11);
                    }
                }
            }
            if (t >= 0) {
                float2 q0, q1, q2;
                if (tri_uv) { const UV3 tq = ld32(reinterpret_cast<const UV3 *>(tri_uv), t); q0 = tq.q0; q1 = tq.q1; q2 = tq.q2; }
                else uv_indirect(uv, uv_tri, t, q0, q1, q2);
                const float mu = (boundary == FPCDR_BOUNDARY_CLAMP && !(tu[k] >= 0.0f && tu[k] <= 1.0f)) ? 0.0f : 1.0f;
                const float mv = (boundary == FPCDR_BOUNDARY_CLAMP && !(tv[k] >= 0.0f && tv[k] <= 1.0f)) ? 0.0f : 1.0f;
                const float gtu = gfx * (float)Wt * mu, gtv = gfy * (float)Ht * mv;
                gu = gtu * (q0.x - q2.x) + gtv * (q0.y - q2.y);
                gvv = gtu * (q1.x - q2.x) + gtv * (q1.y - q2.y);
                if (gu != 0.0f || gvv != 0.0f) tkey = t;
            }
        }
        // ---- vertices: chain through the barycentrics, sum per run of equal triangle, tails add into the LDS table ----
        float gv9[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        int vk[3] = {0, 0, 0};   // the triangle's vertex ids: the run's last lane emits with them
        if (MIP) {
#pragma unroll
            for (int q = 0; q < 9; ++q) gv9[q] = mip_g9[q];
            vk[0] = mip_vk[0]; vk[1] = mip_vk[1]; vk[2] = mip_vk[2];
        } else if (tkey >= 0) {
            { const I3 ti = ld32(reinterpret_cast<const I3 *>(tri), tkey); vk[0] = ti.a; vk[1] = ti.b; vk[2] = ti.c; }
            {
            const float fx = fx_col;
            const float fy = s_fy[rowk0 + 2 * k];
            float g0[3], g1[3], g2[3];
            shade_pixel_bwd<false>(ld32(pos_img, vk[0]), ld32(pos_img, vk[1]), ld32(pos_img, vk[2]), fx, fy, 2.0f / (float)W, 2.0f / (float)H,
                                   make_float4(gu, gvv, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f), g0, g1, g2);
            gv9[0] = g0[0]; gv9[1] = g0[1]; gv9[2] = g0[2];
            gv9[3] = g1[0]; gv9[4] = g1[1]; gv9[5] = g1[2];
            gv9[6] = g2[0]; gv9[7] = g2[1]; gv9[8] = g2[2];
            }
        }
#ifdef FPCDR_SEG_GENERIC
        wave_segment_reduce<9>(tkey, gv9, [&](int, const float (&sm)[9]) {
#else
        wave_segment_reduce9(tkey, gv9, [&](int, const float (&sm)[9]) {
#endif
            // the three slots are claimed with three INDEPENDENT compare-and-swaps in flight (one LDS round trip instead of
            // three); only a vertex whose home slot is taken by another walks on
            unsigned int slot[3];
            int old[3];
#pragma unroll
            for (int kk = 0; kk < 3; ++kk) slot[kk] = (((unsigned int)vk[kk] * 2654435761u) >> 16) & (VSLOTS - 1);
#pragma unroll
            for (int kk = 0; kk < 3; ++kk) old[kk] = atomicCAS(&s_vkey[slot[kk]], -1, vk[kk]);
#pragma unroll
            for (int kk = 0; kk < 3; ++kk) {
                const int key = vk[kk];
                bool done = (old[kk] == -1 || old[kk] == key);
                for (int probe = 1; probe < VSLOTS && !done; ++probe) {
                    slot[kk] = (slot[kk] + 1) & (VSLOTS - 1);
                    const int o = atomicCAS(&s_vkey[slot[kk]], -1, key);
                    done = (o == -1 || o == key);
                }
                if (done) {
                    lds_add_f64(&s_vacc[slot[kk]][0], sm[3 * kk]);
                    lds_add_f64(&s_vacc[slot[kk]][1], sm[3 * kk + 1]);
                    lds_add_f64(&s_vacc[slot[kk]][2], sm[3 * kk + 2]);
                } else {   // table full: straight to memory
                    atomicAdd(&at32(gp, 4u * (unsigned int)key + 0u), sm[3 * kk]); atomicAdd(&at32(gp, 4u * (unsigned int)key + 1u), sm[3 * kk + 1]);
                    atomicAdd(&at32(gp, 4u * (unsigned int)key + 3u), sm[3 * kk + 2]);
                }
            }
        });
    }
#pragma unroll
    for (int c = 0; c < CS; ++c) {
        const float e = wave_sum_dpp(esum[c]);
        if (lane == 0 && e != 0.0f) atomicAdd(&s_esum[c], e);
    }
    __syncthreads();
    if (grad_tex && tid < CS && s_esum[tid] != 0.0f) {   // the empty pixels' share, once per workgroup
        const Taps tp0 = make_taps(0.0f, 0.0f, Ht, Wt, CS, boundary);
        const float e = s_esum[tid];
        if (tp0.valid & 1u) atomicAdd(grad_tex + tp0.i00 + tid, e * ((1.0f - tp0.fx) * (1.0f - tp0.fy)));
        if (tp0.valid & 2u) atomicAdd(grad_tex + tp0.i10 + tid, e * (tp0.fx * (1.0f - tp0.fy)));
        if (tp0.valid & 4u) atomicAdd(grad_tex + tp0.i01 + tid, e * ((1.0f - tp0.fx) * tp0.fy));
        if (tp0.valid & 8u) atomicAdd(grad_tex + tp0.i11 + tid, e * (tp0.fx * tp0.fy));
    }
    // ---- flush: lane = (slot, component), so the four dwords of a vertex are one contiguous 16-byte access ----
    for (int k = tid; k < VSLOTS * 4; k += BWD_NT) {
        const int slot = k >> 2, comp = k & 3;
        const int key = s_vkey[slot];
        if (key >= 0 && comp != 2) {      // (x, y, -, w): z receives no gradient
            const float v = (float)s_vacc[slot][comp == 3 ? 2 : comp];
            if (v != 0.0f) atomicAdd(&at32(gp, 4u * (unsigned int)key + (unsigned int)comp), v);
        }
    }
    if (MIP && grad_tex && ox1 != 0x7fffffff && ma->n_levels >= 1) {
        const int Wt1 = Wt >> 1, Ht1 = Ht >> 1;
        for (int k = tid; k < TEX1 * TEX1 * CS; k += BWD_NT) {
            const float v = (float)s_tex1[k];
            if (v != 0.0f) {
                const int c = k % CS, cell = k / CS;
                const int gx = wrap_near(ox1 + cell % TEX1, Wt1, boundary), gy = wrap_near(oy1 + cell / TEX1, Ht1, boundary);
                atomicAdd(ma->grad[0] + (size_t)(gy * Wt1 + gx) * CS + c, v);
            }
        }
    }
    if (grad_tex && ox != 0x7fffffff) {
        for (int k = tid; k < TEXH * TEXW * CS; k += BWD_NT) {
            const float v = (float)s_tex[k];
            if (v != 0.0f) {
                const int c = k % CS, cell = k / CS;
                const int colx = cell % TEXW, row = cell / TEXW;
                const int gx = wrap_near(ox + colx, Wt, boundary), gy = wrap_near(oy + row, Ht, boundary);   // no division
                atomicAdd(&at32(grad_tex, (unsigned int)((gy * Wt + gx) * CS + c)), v);
            }
        }
    }
}

// grid form (dense mode, and sparse mode after the two-call forward): one workgroup per bin, grid (OX, OY, B)
template <int CS, int BMODE = -1>
__global__ void __launch_bounds__(BWD_NT) FPCDR_BWD_WPE k_render_aa_bwd(const float4 *__restrict__ pos, const int32_t *__restrict__ tri,
                                                       const float2 *__restrict__ uv, const int32_t *__restrict__ uv_tri,
                                                       const float *__restrict__ tex, const float4 *__restrict__ rast,
                                                       const float *__restrict__ color, const float *__restrict__ g_aa,
                                                       const uint8_t *__restrict__ sil,
                                                       const unsigned long long *__restrict__ flags,
                                                       const uint16_t *__restrict__ occ, const float *__restrict__ empty_color,
                                                       int B, int V, int T, int H,
                                                       int W, int Ht, int Wt, int boundary, float *__restrict__ grad_pos,
                                                       float *__restrict__ grad_tex, const float2 *__restrict__ tri_uv,
                                                       const float *__restrict__ upstream, const uint8_t *__restrict__ binflag) {
    render_aa_bwd_body<CS, BMODE>(blockIdx.z, blockIdx.x, blockIdx.y, pos, tri, uv, uv_tri, tex, rast, color, g_aa, sil, flags, occ, empty_color,
                           B, V, T, H, W, Ht, Wt, boundary, grad_pos, grad_tex, tri_uv, upstream, binflag);
}

// grid form of the MIP instantiation (boundary mode at run time)
#ifdef FPCDR_MIPBWD_WAVES
#define FPCDR_MIPBWD_WPE __attribute__((amdgpu_waves_per_eu(FPCDR_MIPBWD_WAVES, 8)))
#else
#define FPCDR_MIPBWD_WPE
#endif
template <int CS>
__global__ void __launch_bounds__(BWD_NT) FPCDR_MIPBWD_WPE k_render_aa_bwd_mip(const float4 *__restrict__ pos, const int32_t *__restrict__ tri,
                                                       const float2 *__restrict__ uv, const int32_t *__restrict__ uv_tri,
                                                       const float *__restrict__ tex, const float4 *__restrict__ rast,
                                                       const float *__restrict__ color, const float *__restrict__ g_aa,
                                                       const uint8_t *__restrict__ sil,
                                                       const unsigned long long *__restrict__ flags,
                                                       const uint16_t *__restrict__ occ, const float *__restrict__ empty_color,
                                                       int B, int V, int T, int H,
                                                       int W, int Ht, int Wt, int boundary, float *__restrict__ grad_pos,
                                                       float *__restrict__ grad_tex, const float2 *__restrict__ tri_uv,
                                                       const float *__restrict__ upstream, const uint8_t *__restrict__ binflag, MipArgs ma) {
    render_aa_bwd_body<CS, -1, true>(blockIdx.z, blockIdx.x, blockIdx.y, pos, tri, uv, uv_tri, tex, rast, color, g_aa, sil, flags, occ, empty_color,
                                    B, V, T, H, W, Ht, Wt, boundary, grad_pos, grad_tex, tri_uv, upstream, binflag, &ma);
}

// list form (after fpcdr_render_loss_fwd): one workgroup per entry of the list k_occ_window built (own or a 4-neighbour bin
// occupied); the launch is sized by the caller's hint, the strided form sweeps up the rest (see k_bins_list, rasterize.hip)
template <int CS, int BMODE = -1>
__global__ void __launch_bounds__(BWD_NT) FPCDR_BWD_WPE k_render_aa_bwd_list(const int32_t *__restrict__ list, const int32_t *__restrict__ count,
                                                       int cap, fpcdr_bin_decode dc, const float4 *__restrict__ pos, const int32_t *__restrict__ tri,
                                                       const float2 *__restrict__ uv, const int32_t *__restrict__ uv_tri,
                                                       const float *__restrict__ tex, const float4 *__restrict__ rast,
                                                       const float *__restrict__ color, const float *__restrict__ g_aa,
                                                       const uint8_t *__restrict__ sil,
                                                       const unsigned long long *__restrict__ flags,
                                                       const uint16_t *__restrict__ occ, const float *__restrict__ empty_color,
                                                       int B, int V, int T, int H,
                                                       int W, int Ht, int Wt, int boundary, float *__restrict__ grad_pos,
                                                       float *__restrict__ grad_tex, const float2 *__restrict__ tri_uv,
                                                       const float *__restrict__ upstream, const uint8_t *__restrict__ binflag) {
    const int item = fpcdr_list_item(*count, cap);      // (XCD x takes the x-th eighth of the entries: common.h)
    if (item < 0) return;
    const int lin = __builtin_amdgcn_readfirstlane(list[item]);
    int b, byi, bxi;
    fpcdr_decode_bin(lin, dc, b, byi, bxi);
    render_aa_bwd_body<CS, BMODE>(b, bxi, byi, pos, tri, uv, uv_tri, tex, rast, color, g_aa, sil, flags, occ,
                                  empty_color, B, V, T, H, W, Ht, Wt, boundary, grad_pos, grad_tex, tri_uv, upstream, binflag);
}

#ifndef FPCDR_BWDQ_WPE
#define FPCDR_BWDQ_WPE
#endif
template <int CS>
__global__ void __launch_bounds__(BWD_NT) FPCDR_BWDQ_WPE k_render_aa_bwd_queue(const int32_t *__restrict__ list, const int32_t *__restrict__ count,
                                                       int first, fpcdr_bin_decode dc,
                                                       const float4 *__restrict__ pos, const int32_t *__restrict__ tri,
                                                       const float2 *__restrict__ uv, const int32_t *__restrict__ uv_tri,
                                                       const float *__restrict__ tex, const float4 *__restrict__ rast,
                                                       const float *__restrict__ color, const float *__restrict__ g_aa,
                                                       const uint8_t *__restrict__ sil,
                                                       const unsigned long long *__restrict__ flags,
                                                       const uint16_t *__restrict__ occ, const float *__restrict__ empty_color,
                                                       int B, int V, int T, int H,
                                                       int W, int Ht, int Wt, int boundary, float *__restrict__ grad_pos,
                                                       float *__restrict__ grad_tex, const float2 *__restrict__ tri_uv,
                                                       const float *__restrict__ upstream, const uint8_t *__restrict__ binflag) {
    const int n = *count;
    for (int item = first + blockIdx.x; item < n; item += gridDim.x) {      // scalar loop variable: uniform for the compiler
        const int lin = __builtin_amdgcn_readfirstlane(list[item]);
        int b, byi, bxi;
        fpcdr_decode_bin(lin, dc, b, byi, bxi);
        render_aa_bwd_body<CS>(b, bxi, byi, pos, tri, uv, uv_tri, tex, rast, color, g_aa, sil, flags, occ,
                               empty_color, B, V, T, H, W, Ht, Wt, boundary, grad_pos, grad_tex, tri_uv, upstream, binflag);
        __syncthreads();     // the next bin's first LDS writes must not overtake this bin's last LDS reads
    }
}

// ------------------------------------------------------------------------------------------------
// Second half of fpcdr_render_loss_fwd.  k_bins<LOSS> (rasterize.hip) gave every pixel the loss term and gradient of
// its un-antialiased colour, marked the pixels with a silhouette pair INSIDE their bin and exported each bin's four
// border lines.  One workgroup per 32x32 bin: 128 threads classify the pairs ACROSS the bin's border from its own and
// its neighbours' lines (2 KB, contiguous), the candidates are gathered into an LDS list, and each gets the full
// antialias + loss pixel, which overwrites its gradient and CORRECTS the loss; flag bits are OR-ed into the
// (zero-filled) planes.  An unoccupied bin was never rasterised (its pixels are empty) but still owns the pairs
// across its right / top border.
// Threads per bin: the kernel is a chain of dependent loads (masks -> border lines -> candidate list -> (z/w, id) -> silhouette
// byte -> triangle -> vertices -> colours) for ~30 candidate pixels per bin, i.e. latency-bound with most lanes idle; one wave per
// bin keeps four times as many bins in flight per CU as four waves do (measured at cfg3: 256 threads 0.26 ms, 64 threads see DESIGN.md).
#ifndef FPCDR_FIX_NT
#define FPCDR_FIX_NT 64
#endif
constexpr int FIX_NT = FPCDR_FIX_NT;
template <int CS>
__device__ __forceinline__ void aa_fix_body(const int b, const int bxi, const int byi, const int OX, const int OY,
                                            const float *__restrict__ color, const float4 *__restrict__ rast,
                                            const float4 *__restrict__ pos, const int32_t *__restrict__ tri,
                                            const uint8_t *__restrict__ sil, const uint8_t *__restrict__ ref, int B, int H,
                                            int W, int V, int T, float bg, float color_scale, float grad_scale,
                                            unsigned long long *__restrict__ flags, float *__restrict__ g_aa,
                                            const uint16_t *__restrict__ occ, const float *__restrict__ empty_color,
                                            const uint32_t *__restrict__ cmask, const unsigned long long *__restrict__ edges,
                                            double *__restrict__ loss_sum, uint8_t *__restrict__ binflag) {
    __shared__ unsigned int s_mask[BBIN];
    __shared__ int s_list[BBIN * BBIN];
    __shared__ int s_n;
    __shared__ float s_part[FIX_NT / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bx0 = bxi * BBIN, by0 = byi * BBIN;
    const OccWin ow = load_occ(occ, b, H, W, bxi, byi);
    const bool v_me = ow.bin(0, 0);
    if (!(v_me || ow.bin(1, 0) || ow.bin(0, 1))) return;
    const size_t bin_id = ((size_t)b * OY + byi) * OX + bxi;
    for (int r = tid; r < BBIN; r += FIX_NT) s_mask[r] = v_me ? cmask[bin_id * BBIN + r] : 0u;
    if (tid == 0) s_n = 0;
    __syncthreads();
    const uint8_t *silb = sil + (size_t)b * T;
    for (int e = tid; e < 4 * BBIN; e += FIX_NT) {
        // pair across the border: side 0 left column, 1 right column, 2 bottom row, 3 top row; the neighbour's facing line
        const int side = e >> 5, i = e & 31;
        const int dx = side == 0 ? -1 : (side == 1 ? 1 : 0), dy = side == 2 ? -1 : (side == 3 ? 1 : 0);
        const int zx = side == 0 ? 0 : (side == 1 ? BBIN - 1 : i), zy = side == 2 ? 0 : (side == 3 ? BBIN - 1 : i);
        const int x = bx0 + zx, y = by0 + zy;
        const int nx = x + dx, ny = y + dy;
        // (an empty, never-written pixel matters only as the OWNER of a pair: right / top side)
        if (x < W && y < H && nx >= 0 && nx < W && ny >= 0 && ny < H && (v_me || (side & 1))) {
            const unsigned long long me = v_me ? edges[bin_id * (4 * BBIN) + side * BBIN + i] : 0ull;
            unsigned long long nb = 0ull;
            if (ow.bin(dx, dy)) {
                const size_t nbin = ((size_t)b * OY + (byi + dy)) * OX + (bxi + dx);
                nb = edges[nbin * (4 * BBIN) + (side ^ 1) * BBIN + i];
            }
            // entries: z/w bits << 32 | silhouette bits << 24 | id + 1
            const int id = (int)((unsigned int)me & 0xffffffu), nid = (int)((unsigned int)nb & 0xffffffu);
            if (id != nid) {
                const float z = __uint_as_float((unsigned int)(me >> 32)), nz = __uint_as_float((unsigned int)(nb >> 32));
                const bool me_first = (side & 1);   // right / upper pairs are (me, n), left / lower pairs (n, me)
                const PairSel ps = me_first ? pair_select(id, z, nid, nz, T) : pair_select(nid, nz, id, z, T);
                const bool takes_n = me_first ? ps.use1 : !ps.use1;
                if (ps.tau >= 0 && (((unsigned int)(takes_n ? nb : me) >> 24) & 0xffu) != 0) atomicOr(&s_mask[zy], 1u << zx);
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < BBIN * BBIN / FIX_NT; ++k) {
        const int pix = k * FIX_NT + tid;
        const bool c = (s_mask[pix >> 5] >> (pix & 31)) & 1u;
        const unsigned long long bal = __ballot(c);
        int base = 0;
        if (lane == 0 && bal) base = atomicAdd(&s_n, __popcll(bal));
        base = __builtin_amdgcn_readfirstlane(base);
        if (c) s_list[base + __popcll(bal & ((1ull << lane) - 1ull))] = pix;
    }
    __syncthreads();
    const int n = __builtin_amdgcn_readfirstlane(s_n);
    if (n == 0) return;
    float ecol[CS];
#pragma unroll
    for (int c = 0; c < CS; ++c) ecol[c] = empty_color[c];
    const size_t img = (size_t)b * H * W;
    const int Wq = FPCDR_AA_ROW_WORDS(W);
    const size_t plane = (size_t)B * H * Wq;
    const AAGeom g = {pos + (size_t)b * V, tri, silb, T, W, H, 0.5f * (float)W, 0.5f * (float)H};
    float lsum = 0.0f;
    bool any_flag = false;
    for (int i = tid; i < n; i += FIX_NT) {
        const int pix = s_list[i];
        const int x = bx0 + (pix & 31), y = by0 + (pix >> 5);
        if (x >= W || y >= H) continue;
        auto zid_at = [&](int xx, int yy) -> float2 {
            return ow.pixel(xx, yy) ? load_zid(rast, img + (size_t)yy * W + xx) : make_float2(0.f, 0.f);
        };
        const bool hasR = x + 1 < W, hasL = x > 0, hasU = y + 1 < H, hasD = y > 0;
        const float2 me = zid_at(x, y);
        const float2 nR = hasR ? zid_at(x + 1, y) : me, nL = hasL ? zid_at(x - 1, y) : me;
        const float2 nU = hasU ? zid_at(x, y + 1) : me, nD = hasD ? zid_at(x, y - 1) : me;
        bool fx_flag = false, fy_flag = false;
        lsum += aa_loss_pixel<CS, true>(g, ow, color, img, x, y, me, nR, nL, nU, nD, hasR, hasL, hasU, hasD, v_me, ecol, ref, bg,
                                        color_scale, grad_scale, g_aa, fx_flag, fy_flag);
        if (v_me && (int)me.y > 0) {   // take back what k_bins added for the un-antialiased colour
            const size_t off = img + (size_t)y * W + x;
            const float rf = (float)ref[off];
            const float d0 = rf - bg * color_scale;
#pragma unroll
            for (int c = 0; c < CS; ++c) {
                const float dd = rf - color[off * CS + c] * color_scale;
                lsum -= dd * dd - d0 * d0;
            }
        }
        const size_t wi = ((size_t)b * H + y) * Wq + (x >> 6);
        if (fx_flag) atomicOr(flags + wi, 1ull << (x & 63));
        if (fy_flag) atomicOr(flags + plane + wi, 1ull << (x & 63));
        any_flag |= fx_flag | fy_flag;
    }
    // per-bin summary of the flag planes (zero-filled by the caller's k_init_queue): the backward kernel loads flag words only
    // where a bin or its left / lower neighbour holds a blended pair -- one bin in seven on a face rig
    if (__builtin_amdgcn_readfirstlane(__syncthreads_or(any_flag ? 1 : 0)) && tid == 0) binflag[bin_id] = 1;
    lsum = wave_sum_dpp(lsum);
    if (lane == 0) s_part[wave] = lsum;
    __syncthreads();
    if (tid == 0) {
        double tot = 0.0;
        for (int w = 0; w < FIX_NT / 64; ++w) tot += (double)s_part[w];
        const unsigned int slot = ((unsigned int)bxi + 31u * (unsigned int)byi + 977u * (unsigned int)b + 128u) % FPCDR_LOSS_SLOTS;
        if (tot != 0.0) atomicAdd(loss_sum + slot, tot);
    }
}

// list form: one workgroup per entry of the list k_occ_window built (own, right or upper bin occupied), + strided sweep
#define FPCDR_AA_FIX_ARGS                                                                                                       \
    const float *__restrict__ color, const float4 *__restrict__ rast, const float4 *__restrict__ pos, const int32_t *__restrict__ tri, \
    const uint8_t *__restrict__ sil, const uint8_t *__restrict__ ref, int B, int H, int W, int V, int T, float bg, float color_scale,  \
    float grad_scale, unsigned long long *__restrict__ flags, float *__restrict__ g_aa, const uint16_t *__restrict__ occ,              \
    const float *__restrict__ empty_color, const uint32_t *__restrict__ cmask, const unsigned long long *__restrict__ edges,           \
    double *__restrict__ loss_sum, uint8_t *__restrict__ binflag
#define FPCDR_AA_FIX_PASS                                                                                                       \
    color, rast, pos, tri, sil, ref, B, H, W, V, T, bg, color_scale, grad_scale, flags, g_aa, occ, empty_color, cmask, edges, loss_sum, binflag
template <int CS>
__global__ void __launch_bounds__(FIX_NT) k_aa_fix_list(const int32_t *__restrict__ list, const int32_t *__restrict__ count, int cap,
                                                        fpcdr_bin_decode dc, FPCDR_AA_FIX_ARGS) {
    const int item = fpcdr_list_item(*count, cap);
    if (item < 0) return;
    const int OX = FPCDR_OCC_DIM(W), OY = FPCDR_OCC_DIM(H);
    const int lin = __builtin_amdgcn_readfirstlane(list[item]);
    int b, byi, bxi;
    fpcdr_decode_bin(lin, dc, b, byi, bxi);
    aa_fix_body<CS>(b, bxi, byi, OX, OY, FPCDR_AA_FIX_PASS);
}
template <int CS>
__global__ void __launch_bounds__(FIX_NT) k_aa_fix_queue(const int32_t *__restrict__ list, const int32_t *__restrict__ count, int first,
                                                      fpcdr_bin_decode dc, FPCDR_AA_FIX_ARGS) {
    const int n = *count;
    const int OX = FPCDR_OCC_DIM(W), OY = FPCDR_OCC_DIM(H);
    for (int item = first + blockIdx.x; item < n; item += gridDim.x) {
        const int lin = __builtin_amdgcn_readfirstlane(list[item]);
        int b, byi, bxi;
        fpcdr_decode_bin(lin, dc, b, byi, bxi);
        aa_fix_body<CS>(b, bxi, byi, OX, OY, FPCDR_AA_FIX_PASS);
        __syncthreads();
    }
}

}  // namespace

// ---- a piece of fpcdr_render_loss_fwd (rasterize.hip) that lives in this file; not part of the C ABI ----
int fpcdr_launch_aa_fix(const fpcdr_aa_loss_fwd_params *p, const uint32_t *cmask, const unsigned long long *edges,
                        const int32_t *fix_list, const int32_t *fix_count, int nbins, hipStream_t st) {
    const int cap = (p->cap_fix > 0 && p->cap_fix < nbins) ? p->cap_fix : nbins;
    const fpcdr_bin_decode dc = fpcdr_make_bin_decode(FPCDR_OCC_DIM(p->W), FPCDR_OCC_DIM(p->H));
#define ARGS                                                                                                                   \
    p->color, (const float4 *)p->rast, (const float4 *)p->pos, p->tri, p->sil, p->ref, p->B, p->H, p->W, p->V, p->T, p->bg,   \
    p->color_scale, p->grad_scale, (unsigned long long *)p->flags, p->grad_aa, p->occ, p->empty_color, cmask, edges, p->loss_sum,   \
    (uint8_t *)p->occ + fpcdr_queue_layout_of(p->B, p->H, p->W).occ_binflag
#define LAUNCH(CS)                                                                                                             \
    do {                                                                                                                       \
        hipLaunchKernelGGL(k_aa_fix_list<CS>, dim3(fpcdr_list_grid(cap)), dim3(FIX_NT), 0, st, fix_list, fix_count, cap, dc, ARGS); \
        if (cap < nbins) hipLaunchKernelGGL(k_aa_fix_queue<CS>, dim3(FPCDR_SWEEP_WGS), dim3(FIX_NT), 0, st, fix_list, fix_count, cap, dc, ARGS); \
    } while (0)
    if (p->C == 1) LAUNCH(1);
    else if (p->C == 3) LAUNCH(3);
    else LAUNCH(4);
#undef LAUNCH
#undef ARGS
    FPCDR_CHECK_LAUNCH();
    return FPCDR_OK;
}

extern "C" int fpcdr_aa_loss_fwd(const fpcdr_aa_loss_fwd_params *p, void *stream) {
    FPCDR_REQUIRE(p != nullptr, "null params");
    FPCDR_REQUIRE(p->color && p->rast && p->pos && p->tri && p->adj && p->ref && p->sil && p->flags && p->grad_aa && p->loss_sum,
                  "null pointer");
    FPCDR_REQUIRE(p->B > 0 && p->H > 0 && p->W > 0 && p->C > 0 && p->V > 0 && p->T > 0, "sizes must be positive");
    FPCDR_REQUIRE(p->C == 1 || p->C == 3 || p->C == 4, "fused objective supports C = 1, 3, 4");
    FPCDR_REQUIRE(p->B <= 65535 && fpcdr_cdiv(p->H, 32) <= 65535, "image batch / height too large for one launch");
    hipStream_t st = (hipStream_t)stream;
    {
        const int rc_sil = fpcdr_launch_sil(p->pos, p->tri, p->adj, p->B, p->V, p->T, p->H, p->W, p->sil, nullptr, 0, st);      // (objective.hip)
        if (rc_sil) return rc_sil;
    }
    dim3 grid(fpcdr_cdiv(p->W, 64), fpcdr_cdiv(p->H, 32), p->B);
#define LAUNCH(CS, SP)                                                                                                         \
    hipLaunchKernelGGL((k_aa_loss<CS, SP>), grid, dim3(256), 0, st, p->color, (const float4 *)p->rast, (const float4 *)p->pos, \
                       p->tri, p->sil, p->ref, p->B, p->H, p->W, p->V, p->T, p->bg, p->color_scale, p->grad_scale,             \
                       (unsigned long long *)p->flags, p->grad_aa, p->occ, p->empty_color, p->loss_sum)
    if (p->occ) {
        FPCDR_REQUIRE(p->empty_color != nullptr, "sparse mode needs empty_color");
        if (p->C == 1) LAUNCH(1, true);
        else if (p->C == 3) LAUNCH(3, true);
        else LAUNCH(4, true);
    } else {
        if (p->C == 1) LAUNCH(1, false);
        else if (p->C == 3) LAUNCH(3, false);
        else LAUNCH(4, false);
    }
#undef LAUNCH
    FPCDR_CHECK_LAUNCH();
    return FPCDR_OK;
}

extern "C" int fpcdr_render_aa_bwd(const fpcdr_render_aa_bwd_params *p, void *stream) {
    FPCDR_REQUIRE(p != nullptr, "null params");
    FPCDR_REQUIRE(p->pos && p->tri && p->uv && p->uv_tri && p->tex && p->rast && p->color && p->grad_aa && p->sil && p->flags &&
                      p->grad_pos,
                  "null pointer");
    FPCDR_REQUIRE(p->B > 0 && p->V > 0 && p->T > 0 && p->H > 0 && p->W > 0 && p->Ht > 0 && p->Wt > 0 && p->C > 0,
                  "sizes must be positive");
    FPCDR_REQUIRE(p->C == 1 || p->C == 3 || p->C == 4, "fused objective supports C = 1, 3, 4");
    FPCDR_REQUIRE(p->B <= 65535, "more than 65535 images per call");
    FPCDR_REQUIRE((long long)p->Ht * p->Wt * p->C < (1ll << 30) && p->H <= 32767 && p->W <= 32767 && p->T < (1 << 24),
                  "texture / resolution / mesh too large for the fused backward (32-bit gather offsets)");
    FPCDR_REQUIRE(!p->occ || p->empty_color, "sparse mode needs empty_color");
    FPCDR_REQUIRE(p->boundary_mode == FPCDR_BOUNDARY_WRAP || p->boundary_mode == FPCDR_BOUNDARY_CLAMP || p->boundary_mode == FPCDR_BOUNDARY_ZERO,
                  "bad boundary mode");
    hipStream_t st = (hipStream_t)stream;
    FPCDR_REQUIRE(!(p->queued || p->binflags) || p->occ != nullptr, "queued / binflags need the occupancy buffer of fpcdr_render_loss_fwd");
    const uint8_t *binflag = p->binflags ? (const uint8_t *)p->occ + fpcdr_queue_layout_of(p->B, p->H, p->W).occ_binflag : nullptr;
    if (p->mip) {      // the reference's enable_mip branch: one workgroup per bin
        FPCDR_REQUIRE(p->n_levels >= 0 && p->n_levels <= FPCDR_MAX_MIP, "bad n_levels");
        MipArgs ma;
        for (int lvl = 0; lvl < FPCDR_MAX_MIP; ++lvl) {
            ma.tex[lvl] = lvl < p->n_levels ? p->tex_mip[lvl] : nullptr;
            ma.grad[lvl] = (lvl < p->n_levels && p->grad_tex) ? p->grad_tex_mip[lvl] : nullptr;
            FPCDR_REQUIRE(lvl >= p->n_levels || (ma.tex[lvl] && (!p->grad_tex || ma.grad[lvl])), "missing mip level");
        }
        ma.n_levels = p->n_levels;
        dim3 grid(fpcdr_cdiv(p->W, BBIN), fpcdr_cdiv(p->H, BBIN), p->B);
#define LAUNCH_MIP(CS)                                                                                                                    \
        hipLaunchKernelGGL(k_render_aa_bwd_mip<CS>, grid, dim3(BWD_NT), 0, st, (const float4 *)p->pos, p->tri, (const float2 *)p->uv, p->uv_tri, \
                           p->tex, (const float4 *)p->rast, p->color, p->grad_aa, p->sil, (const unsigned long long *)p->flags, p->occ,          \
                           p->empty_color, p->B, p->V, p->T, p->H, p->W, p->Ht, p->Wt, p->boundary_mode, p->grad_pos, p->grad_tex,               \
                           (const float2 *)p->tri_uv, p->upstream, binflag, ma)
        if (p->C == 1) LAUNCH_MIP(1);
        else if (p->C == 3) LAUNCH_MIP(3);
        else LAUNCH_MIP(4);
#undef LAUNCH_MIP
        FPCDR_CHECK_LAUNCH();
        return FPCDR_OK;
    }
    if (p->queued) {
        const fpcdr_queue_layout q = fpcdr_queue_layout_of(p->B, p->H, p->W);
        const int32_t *hdr = (const int32_t *)((const char *)p->occ + q.occ_hdr);      // [0] = number of listed bins
        const int32_t *list = (const int32_t *)((const char *)p->occ + q.occ_bwd_list);
        const long long nbins = (long long)p->B * FPCDR_OCC_DIM(p->H) * FPCDR_OCC_DIM(p->W);
        const int cap = (p->cap_bwd > 0 && p->cap_bwd < nbins) ? p->cap_bwd : (int)nbins;
        const fpcdr_bin_decode dc = fpcdr_make_bin_decode(FPCDR_OCC_DIM(p->W), FPCDR_OCC_DIM(p->H));
#define ARGSQ                                                                                                               \
        (const float4 *)p->pos, p->tri, (const float2 *)p->uv, p->uv_tri, p->tex, (const float4 *)p->rast, p->color, p->grad_aa,  \
        p->sil, (const unsigned long long *)p->flags, p->occ, p->empty_color, p->B, p->V, p->T, p->H, p->W, p->Ht, p->Wt,        \
        p->boundary_mode, p->grad_pos, p->grad_tex, (const float2 *)p->tri_uv, p->upstream, binflag
#define LAUNCHQ(CS)                                                                                                         \
        do {                                                                                                                \
            hipLaunchKernelGGL(k_render_aa_bwd_list<CS>, dim3(fpcdr_list_grid(cap)), dim3(BWD_NT), 0, st, list, hdr, cap, dc, ARGSQ);  \
            if (cap < nbins)                                                                                                \
                hipLaunchKernelGGL(k_render_aa_bwd_queue<CS>, dim3(FPCDR_SWEEP_WGS), dim3(BWD_NT), 0, st, list, hdr, cap, dc, ARGSQ); \
        } while (0)
        if (p->C == 1 && p->boundary_mode == FPCDR_BOUNDARY_WRAP) {     // the reference's case
            hipLaunchKernelGGL((k_render_aa_bwd_list<1, FPCDR_BOUNDARY_WRAP>), dim3(fpcdr_list_grid(cap)), dim3(BWD_NT), 0, st, list, hdr, cap, dc, ARGSQ);
            if (cap < nbins)
                hipLaunchKernelGGL(k_render_aa_bwd_queue<1>, dim3(FPCDR_SWEEP_WGS), dim3(BWD_NT), 0, st, list, hdr, cap, dc, ARGSQ);
        } else if (p->C == 1) LAUNCHQ(1);
        else if (p->C == 3) LAUNCHQ(3);
        else LAUNCHQ(4);
#undef LAUNCHQ
#undef ARGSQ
        FPCDR_CHECK_LAUNCH();
        return FPCDR_OK;
    }
    dim3 grid(fpcdr_cdiv(p->W, BBIN), fpcdr_cdiv(p->H, BBIN), p->B);
#define LAUNCH(CS)                                                                                                          \
    hipLaunchKernelGGL(k_render_aa_bwd<CS>, grid, dim3(BWD_NT), 0, st, (const float4 *)p->pos, p->tri,        \
                       (const float2 *)p->uv, p->uv_tri, p->tex, (const float4 *)p->rast, p->color, p->grad_aa, p->sil,     \
                       (const unsigned long long *)p->flags, p->occ, p->empty_color, p->B, p->V, p->T, p->H, p->W, p->Ht, p->Wt,   \
                       p->boundary_mode,                                                                                        \
                       p->grad_pos, p->grad_tex, (const float2 *)p->tri_uv, p->upstream, binflag)
    if (p->C == 1 && p->boundary_mode == FPCDR_BOUNDARY_WRAP) {     // the reference's case
        hipLaunchKernelGGL((k_render_aa_bwd<1, FPCDR_BOUNDARY_WRAP>), grid, dim3(BWD_NT), 0, st, (const float4 *)p->pos, p->tri,
                           (const float2 *)p->uv, p->uv_tri, p->tex, (const float4 *)p->rast, p->color, p->grad_aa, p->sil,
                           (const unsigned long long *)p->flags, p->occ, p->empty_color, p->B, p->V, p->T, p->H, p->W, p->Ht, p->Wt,
                           p->boundary_mode, p->grad_pos, p->grad_tex, (const float2 *)p->tri_uv, p->upstream, binflag);
    } else if (p->C == 1) LAUNCH(1);
    else if (p->C == 3) LAUNCH(3);
    else LAUNCH(4);
#undef LAUNCH
    FPCDR_CHECK_LAUNCH();
    return FPCDR_OK;
}
