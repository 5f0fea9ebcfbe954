// One-pass pixel objective for gfx950 (MI355X): value AND gradient of reference src/torch/fit.py:151-161 + the pixel term of :579 in
// one call (include/fpcdr.h, fpcdr_objective_fwd).
//
// The two-call form (fpcdr_render_loss_fwd + fpcdr_render_aa_bwd, fused.hip) writes rast, colour and d loss / d colour -- 24 B/px --
// so that the backward call can read them back and RE-DERIVE what the forward held in registers: 45 % of that backward's vector
// instructions (profiles/r04_backward_ablation.txt).  The objective is a scalar, so its gradient is known the moment a pixel's loss
// term is: here the thread that shades a pixel also chains its gradient back while barycentrics, taps, texels and vertices are live.
//
//   k_bins_list<IDS> (rasterize.hip)   coverage + depth only; leaves (triangle + 1) | silhouette bits << 24 per pixel, 4 KB per bin
//   k_shade                            one workgroup per occupied bin: shade, texture, background + squared error, texture /
//                                      interpolate / rasterize backward of the un-antialiased gradient into per-bin LDS accumulators
//                                      (vertex table + texel window, doubles), one flush.  Pixels with a pixel pair of different ids
//                                      at a triangle that owns a silhouette edge -- known exactly from the id planes, the neighbour
//                                      bins' border lines included -- are DEFERRED: their record, colour and gradient are also stored
//   k_fix<0>                           deferred pixels only: exact antialias blend, loss correction, final d loss / d colour
//   k_fix<1>                           deferred pixels only: the DIFFERENCE between the gradient arriving at their colour through the
//                                      antialias op and the un-antialiased one k_shade scattered (everything downstream is linear in
//                                      it), and the antialias op's own d alpha / d pos
//
// The antialias pair analysis -- ~1 % of the covered pixels, 56 spilled VGPRs in the old backward -- lives in k_fix alone.
#include "common.h"

#include <stdlib.h>

namespace {

#include "texsample.h"
#include "raster_math.h"
#include "aa_pairs.h"
#include "sil_bits.h"

constexpr int OB = 32;            // pixels per bin side
constexpr int OS = 36;            // LDS row stride of the id plane with its one-pixel apron (34 entries used)
// cells of the (non-mip) texel window, a rectangle shaped by the footprint of half a bin.  One channel: 1 216 cells of 8 bytes hold a
// half's footprint on the face rig (27 x 33 texels) and leave LDS for SEVEN workgroups per CU (22.4 KB each); for the whole bin 1 728
// cells were what six workgroups left (profiles/r05_flush_experiments.txt 10, 13).  Three / four channels: 1 600 cells of 24 / 32
// bytes, under 64 KB with the rest
template <int CS> constexpr int owin_cells() { return CS == 1 ? 1216 : 1600; }
constexpr int ONT = 256;          // threads of k_shade
constexpr int FNT = 64;           // threads of k_fix per bin (a chain of dependent loads for a few dozen pixels)

struct ObjArgs {
    const float4 *pos; const int32_t *tri; const float2 *uv; const int32_t *uv_tri; const float2 *tri_uv;
    const float *tex; const uint8_t *ref; const uint8_t *sil;
    const uint32_t *idp; const uint16_t *occ; uint8_t *binflag; uint32_t *cmask;
    float *esum;            // [ESLOTS][4]: gradient arriving at the colour of EMPTY pixels (k_fix<1> -> k_objective_finish), zeroed by the call
    int32_t *def_list, *def_count;      // bins that hold a deferred pixel, appended by k_shade (k_fix runs over these only)
    uint32_t *hitmask;      // [bins][32] row masks of the pixels that took part in a blend (k_fix<0> -> k_fix<1>); zeroed per bin by k_shade
    float4 *rec; float *color; float *g_aa; const float *empty_color;
    double *loss_sum; float *grad_pos; float *grad_tex;
    int B, V, T, H, W, Ht, Wt, boundary;
    float bg, color_scale, grad_scale;
    unsigned long long *flags;      // optional (tests / diagnostics): the antialias flag planes of fpcdr_antialias_fwd, zero-filled by the caller
    // COMPACT records (fpcdr_objective_params.rec_slots > 0): rec / color / g_aa hold pool_cap slots of 1 024 pixels instead of the whole
    // batch; a bin that shows a silhouette triangle (the only bins that can hold a deferred pixel) takes the next slot (pool_count, header
    // [1] of occ: it keeps counting beyond the capacity, so that the caller learns the need), slot_of[bin] = its slot or -1
    int32_t *slot_of; int32_t *pool_count; int32_t *overflow; int pool_cap;
};
// COLD kernel arguments.  The compiler loads every kernel argument a kernel uses at its entry and keeps it to where it is used -- in a
// scalar register or, once those run out, in a lane of a vector register (v_writelane / v_readlane: vector-issue slots, the resource
// k_shade is bound by; r5: 47 spilled scalar registers, 173 v_readlane).  What shade_body needs once or rarely -- the record arrays of
// the deferred branch, the index-buffer form of the texture coordinates, everything the bin's epilogue writes -- is read from the
// kernel-argument segment WHERE IT IS USED: cold(ka)->field is a scalar load behind an opaque barrier it cannot be hoisted across.
typedef const __attribute__((address_space(4))) ObjArgs *KArgs;
__device__ __forceinline__ KArgs cold(KArgs p) { asm volatile("" : "+s"(p)); return p; }
template <typename K> __device__ __forceinline__ KArgs kernarg_objargs() {      // K: the kernel's ONE parameter, a struct with a member `a`
    return (KArgs)((const __attribute__((address_space(4))) char *)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(K, a));
}
// element index of pixel (xx, yy) of image-bin `lin` (+ the offset to a neighbouring bin) in the record arrays
__device__ __forceinline__ size_t rec_slot_index(int slot, int xx, int yy) { return (size_t)(slot < 0 ? 0 : slot) * (OB * OB) + (yy & 31) * OB + (xx & 31); }

// MIP instantiations (the reference's enable_mip branch, fit.py:153-155): level l of the chain = tex[l - 1] / grad[l - 1]; level 0 is
// ObjArgs.tex / grad_tex
// (the whole chain, level 0 included, as ONE kernel argument: the level is a run-time index, and a TexLevels assembled in a local
//  variable is a private array -- 288 bytes of scratch memory per lane; indexed in the kernel-argument segment it is a scalar load)
struct MipO {
    TexLevels lv;      // tex[0] = ObjArgs.tex, grad[0] = ObjArgs.grad_tex; grad[] all null when no texture gradient is wanted
    int n_levels;
};
// MIP: the texel gradients of a bin go through THREE LDS windows, for the finest level any of its sampled pixels uses (lb) and the two
// above it -- a pixel's lookup blends two adjacent levels, so a bin whose level of detail spans less than two levels stays inside.  A
// window is a rectangle of at most MWIN_CELLS[j] / CS cells whose shape follows the bin's footprint in its level (the prepass of
// shade_body measures it: the extent of the texture coordinates of the pass-0 pixels), because a footprint is rarely square: the rig's
// face runs at 1.6 texels per pixel along v and 1.0 along u, 52 x 33 texels of level 0 under a 32 x 32 bin.  (r4/r5 first form: a fixed
// 40 x 40 window of level 0 and 24 x 24 of level 1.  A third of the pixels' taps fell outside and went to memory one float atomic
// each -- 2.3 ms of the kernel's 4.8 at cfg3, profiles/r05_mip_windows.txt.)
template <int CS> struct MipWinCaps {
    static constexpr int N0 = CS == 1 ? 2304 : 1088, N1 = CS == 1 ? 704 : 384, N2 = CS == 1 ? 256 : 128;      // cells
    static constexpr int TOTAL = N0 + N1 + N2;
};
// a tap's column (row) relative to a window's unwrapped origin, modulo the level's width (height): both lie within one period of zero
__device__ __forceinline__ int wrap_cell(int d, int n) { return d < 0 ? d + n : (d >= n ? d - n : d); }
// order-preserving float <-> int keys for the integer wave / LDS min and max
__device__ __forceinline__ int fkey(float x) { const int i = __float_as_int(x); return i ^ ((i >> 31) & 0x7fffffff); }
__device__ __forceinline__ float fkey_inv(int k) { return __int_as_float(k ^ ((k >> 31) & 0x7fffffff)); }

// texture coordinates of triangle t through the index buffer (callers that did not pre-gather uv[uv_tri]); out of line, see fused.hip
// (returns BY VALUE, in registers: reference arguments of an out-of-line function live in scratch memory, and a kernel with a private
//  segment is dispatched several times slower -- k_fix<1> spent 220 us launching 90 k one-wave workgroups, 47 us without scratch)
struct UV3 { float2 q0, q1, q2; };
__device__ __noinline__ UV3 uv_indirect(const float2 *__restrict__ uv, const int32_t *__restrict__ uv_tri, int t) {
    UV3 r;
    r.q0 = uv[uv_tri[3 * t]]; r.q1 = uv[uv_tri[3 * t + 1]]; r.q2 = uv[uv_tri[3 * t + 2]];
    return r;
}

// d(grad_scale * sum of squares) / d colour for one channel: ONE expression for the three kernels, so that a deferred pixel no blend
// touches gets a difference of exactly zero in k_fix<1>
__device__ __forceinline__ float loss_grad(float dd, float color_scale, float grad_scale) { return (-2.0f * color_scale * grad_scale) * dd; }

// a pixel pair can change under antialiasing only if the ids differ and one of the two triangles owns a silhouette edge
__device__ __forceinline__ bool pair_maybe(unsigned int a, unsigned int b) { return ((a ^ b) & 0xffffffu) != 0u && ((a | b) >> 24) != 0u; }

struct I3 { int a, b, c; };

// -DFPCDR_OPROF (scripts/prof_phases.py): wave-cycles of k_shade by phase, summed over all waves into a device array
#ifdef FPCDR_OPROF
__device__ unsigned long long g_oprof[16];
#define OPROF_DECL long long oprof_t0 = 0, oprof_t1 = 0, oprof_t2 = 0, oprof_t3 = 0, oprof_t4 = 0, oprof_t5 = 0, oprof_t6 = 0, oprof_t7 = 0
#define OPROF_T(i) oprof_t##i = clock64()
#define OPROF_ADD(slot, a_, b_) do { if ((threadIdx.x & 63) == 0) atomicAdd(&g_oprof[slot], (unsigned long long)(oprof_t##b_ - oprof_t##a_)); } while (0)
#else
#define OPROF_DECL
#define OPROF_T(i)
#define OPROF_ADD(slot, a_, b_)
#endif

// ------------------------------------------------------------------------------------------------
// k_shade: one 32 x 32 bin, 256 threads, four pixels per thread.  A wave pass covers two adjacent rows of the bin, the second one right
// to left, so that the pixels of a triangle are neighbours in lane order and ONE segmented scan per pass sums the nine vertex gradient
// components of every run (common.h wave_segment_reduce9).
//
// The texel gradients of the bin are summed in an LDS window and flushed once per cell ("The texel window" / "Two halves,
// seven workgroups").  The bin is shaded as two HALVES (rows 0-15, 16-31) of two wave passes each.  A half's first pass covers its row
// pairs 0, 2, 5, 7 -- its first and last rows among them: for a locally affine uv map the extreme taps lie there -- and keeps each
// pixel's taps in registers; their bounding box, reduced over the workgroup, places the half's window (a rectangle of at most 1 216
// cells shaped by the box, seam-aware); the kept taps and the second pass add into it; whatever still falls outside (3.8 % of the
// pixels at cfg3: a second surface, a pole of the uv map) goes to memory with float atomics.  The first half's window is flushed and
// zeroed before the second half's is placed.  So a pixel's four texel adds happen right where its weights are formed: apart from the
// first pass's taps nothing is kept across a barrier (the two-phase form of r3 held 20 registers per thread for it: 88 VGPRs and 5
// waves per SIMD, where this form runs 7).
#define FPCDR_SHADE_WPE __attribute__((amdgpu_waves_per_eu(CS == 1 ? 7 : 1, 8)))      // one channel: seven workgroups per CU (72 registers, 22.4 KB of LDS)
// row pair (rows 2 p, 2 p + 1) of wave w in pass k of the whole-bin order (the mip instantiation)
__device__ __forceinline__ int shade_row_pair(int k, int w) {
    // k = 0: 0, 5, 10, 15;  k = 1: 1, 6, 11, 14;  k = 2: 2, 7, 12, 13;  k = 3: 3, 4, 8, 9
    const unsigned int packed = k == 0 ? 0xFA50u : (k == 1 ? 0xEB61u : (k == 2 ? 0xDC72u : 0x9843u));
    return (int)((packed >> (4 * w)) & 15u);
}

template <int CS, int BMODE, bool MIP = false>
__device__ __forceinline__ void shade_body(const int b, const int bxi, const int byi, const int OX, const int OY, const ObjArgs &a,
                                           const KArgs ka, const MipO *ma = nullptr) {
    const int boundary = BMODE >= 0 ? BMODE : a.boundary;
    __shared__ int s_mred[18];      // MIP prepass: min level, min / max keys of the prepared u and v, and of both shifted by half a period
    __shared__ unsigned int s_id[(OB + 2) * OS];
    __shared__ int s_vkey[FPCDR_VT_SLOTS];
    __shared__ double s_vacc[FPCDR_VT_SLOTS][3];
    constexpr int OCELLS = owin_cells<CS>();
    constexpr int TEX_CELLS = MIP ? MipWinCaps<CS>::TOTAL : OCELLS;
    __shared__ double s_tex[TEX_CELLS * CS];      // texel window(s), doubles: ds_add_f64 (common.h lds_add_f64)
    __shared__ unsigned int s_cmask[OB];
    __shared__ __attribute__((aligned(16))) int s_vote[8];      // block4_any: one slot per wave and vote
    __shared__ float s_fy[OB];
    __shared__ float s_lpart[ONT / 64];
    const VTable vt = {s_vkey, s_vacc};
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = (lane & 32) ? 63 - lane : lane;
    const int bx0 = bxi * OB, by0 = byi * OB;
    const int x = bx0 + col;
    const int H = a.H, W = a.W, Ht = a.Ht, Wt = a.Wt;
    const size_t bin_lin = ((size_t)b * OY + byi) * OX + bxi;
    const unsigned int wmask = (unsigned int)__builtin_amdgcn_readfirstlane((int)a.occ[bin_lin]);
    const bool want_tex = a.grad_tex != nullptr, want_pos = a.grad_pos != nullptr, want_grad = want_tex || want_pos;   // (uniform)

    OPROF_DECL;
    OPROF_T(0);
    // ---- phase 0: the bin's ids with a one-pixel apron from the neighbours' planes; tables ----
    unsigned int sil_seen;      // silhouette bits of the entries this thread loads (own plane + apron), OR-ed
    {
        const uint4 v = reinterpret_cast<const uint4 *>(a.idp + bin_lin * (OB * OB))[tid];
        unsigned int *d = &s_id[((tid >> 3) + 1) * OS + (tid & 7) * 4 + 1];
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        sil_seen = (v.x | v.y | v.z | v.w) >> 24;
    }
    if (tid < 4 * OB) {
        const int side = tid >> 5, i = tid & 31;      // 0 left, 1 right, 2 below, 3 above
        const int dx = side == 0 ? -1 : (side == 1 ? 1 : 0), dy = side == 2 ? -1 : (side == 3 ? 1 : 0);
        unsigned int e = 0u;
        if ((wmask >> ((dy + 1) * 4 + dx + 1)) & 1u) {      // (set only for bins inside the image that were rasterised)
            const size_t nb = (size_t)((long long)bin_lin + dy * OX + dx);
            const int sx = side == 0 ? OB - 1 : (side == 1 ? 0 : i), sy = side == 2 ? OB - 1 : (side == 3 ? 0 : i);
            e = a.idp[nb * (OB * OB) + sy * OB + sx];
        }
        const int tx = side == 0 ? 0 : (side == 1 ? OB + 1 : i + 1), ty = side == 2 ? 0 : (side == 3 ? OB + 1 : i + 1);
        s_id[ty * OS + tx] = e;
        sil_seen |= e >> 24;
    }
    if (want_pos) vtable_init(vt, tid, ONT);
    if (want_tex)
        for (int k = tid; k < TEX_CELLS * CS; k += ONT) s_tex[k] = 0.0;
    if (tid == 0) {
        s_mred[0] = 0x7fffffff;
        for (int k = 1; k < 9; k += 2) { s_mred[k] = 0x7fffffff; s_mred[k + 1] = (int)0x80000000; s_mred[9 + k] = 0x7fffffff; s_mred[10 + k] = (int)0x80000000; }
    }
    if (tid < OB) {
        s_cmask[tid] = 0u;
        s_fy[tid] = (2.0f * (float)(by0 + tid) + 1.0f) / (float)H - 1.0f;      // NDC y of the bin's rows: one IEEE division per row
    }
    OPROF_T(1);
    // (barrier: plane, apron and tables are in place.)  A pixel pair can be deferred only if one of its triangles owns a silhouette edge:
    // four bins in five of a face show none at all -- the interior of the mesh -- and skip the four neighbour tests of their pixels
    const bool bin_sil = block4_any(s_vote, sil_seen != 0u);
    OPROF_T(2);
    // compact records: this bin's slot (one global fetch-add and a barrier in the one bin in five that shows a silhouette triangle)
    const bool compact = a.pool_cap > 0;      // (uniform)
    __shared__ int s_slot;
    int slot = -1;
    if (compact) {
        if (bin_sil) {
            if (tid == 0) {
                int sl = atomicAdd(a.pool_count, 1);
                if (sl >= a.pool_cap) { sl = -1; *a.overflow = 1; }      // (no room: the bin's pixels go un-deferred and the call says so)
                s_slot = sl;
                a.slot_of[bin_lin] = sl;
            }
            __syncthreads();
            slot = __builtin_amdgcn_readfirstlane(s_slot);
        } else if (tid == 0) {
            a.slot_of[bin_lin] = -1;
        }
    }

    // (a bin the pool had no slot for defers nothing -- the call reports the overflow --: one uniform flag for both facts)
    const bool bin_sil_rec = bin_sil && !(compact && slot < 0);

    const size_t img = (size_t)b * H * W;
    const size_t bin_off = img + (size_t)by0 * W + bx0;
    const float4 *const pos_img = a.pos + (size_t)b * a.V;
    float *const gp = want_pos ? a.grad_pos + (size_t)b * a.V * 4 : nullptr;
    const float fx_col = (2.0f * (float)x + 1.0f) / (float)W - 1.0f;
    const float cs = a.color_scale, gs = a.grad_scale;
    const float bgs = a.bg * cs;
    float lsum = 0.0f;
    bool any_def = false;
    bool owrap = false;      // the window reaches across the edge of a periodic coordinate
    int ox = 0, oy = 0, ows = 1, owr = 1;      // origin, row stride and rows of the texel window (set behind the barrier after pass 0; stride 1: none)
    // MIP: the three windows (set behind the prepass's barrier), as scalars -- structs selected per lane ended up in scratch memory
    int w0x = 0, w0y = 0, w0s = 1, w0r = 1, w1x = 0, w1y = 0, w1s = 1, w1r = 1, w2x = 0, w2y = 0, w2s = 1, w2r = 1;
    bool w0wrap = false, w1wrap = false, w2wrap = false;      // the window reaches across the edge of its level (then cells are found modulo its size)
    constexpr int w0b = 0, w1b = MipWinCaps<CS>::N0, w2b = MipWinCaps<CS>::N0 + MipWinCaps<CS>::N1;
    int mlb = 0;                                             // ... and the level of the first

    // the four texel adds of one pixel: into the window, or -- outside it -- to memory
    auto add_taps = [&](const float (&gc)[CS], float fx, float fy, int x0, int y0) {
        unsigned int valid = 0xFu;
        if (boundary == FPCDR_BOUNDARY_ZERO) {
            const bool vx0 = x0 >= 0 && x0 < Wt, vx1 = x0 >= -1 && x0 < Wt - 1, vy0 = y0 >= 0 && y0 < Ht, vy1 = y0 >= -1 && y0 < Ht - 1;
            valid = (vx0 && vy0 ? 1u : 0u) | (vx1 && vy0 ? 2u : 0u) | (vx0 && vy1 ? 4u : 0u) | (vx1 && vy1 ? 8u : 0u);
        }
        const float w00 = (valid & 1u) ? (1.0f - fx) * (1.0f - fy) : 0.0f, w10 = (valid & 2u) ? fx * (1.0f - fy) : 0.0f;
        const float w01 = (valid & 4u) ? (1.0f - fx) * fy : 0.0f, w11 = (valid & 8u) ? fx * fy : 0.0f;
        int lx = (int)((unsigned int)x0 - (unsigned int)ox), ly = (int)((unsigned int)y0 - (unsigned int)oy);
        // (the window's origin is unwrapped; a window that lies inside the texture -- nearly all do -- needs no wrapping: uniform branch)
        if (boundary == FPCDR_BOUNDARY_WRAP && owrap) { lx = wrap_cell(lx, Wt); ly = wrap_cell(ly, Ht); }
        // (unsigned compares: a tap below the origin wraps to a huge value; stride 1 = no sample in pass 0: everything outside)
        const bool inside = (unsigned int)lx < (unsigned int)(ows - 1) && (unsigned int)ly < (unsigned int)(owr - 1);
        if (inside) {
#pragma unroll
            for (int c = 0; c < CS; ++c) {
                double *wp = s_tex + (ly * ows + lx) * CS + c;
                lds_add_f64(wp, gc[c] * w00);
                lds_add_f64(wp + CS, gc[c] * w10);
                lds_add_f64(wp + ows * CS, gc[c] * w01);
                lds_add_f64(wp + ows * CS + CS, gc[c] * w11);
            }
        } else {
            // (a wave with ONE pixel outside its window issues all of this: with the coordinate wrapped into [0, 1) the taps lie in
            //  [-1, n - 1], so one conditional add and the clamp that make_taps_fast uses do what wrap_near's general form does)
            int ix0, ix1, iy0, iy1;
            if (boundary == FPCDR_BOUNDARY_WRAP) {
                ix0 = clamp_idx(x0 + (x0 < 0 ? Wt : 0), Wt); iy0 = clamp_idx(y0 + (y0 < 0 ? Ht : 0), Ht);
                ix1 = clamp_idx(x0 >= Wt - 1 ? x0 + 1 - Wt : x0 + 1, Wt); iy1 = clamp_idx(y0 >= Ht - 1 ? y0 + 1 - Ht : y0 + 1, Ht);
            } else {
                ix0 = wrap_near(x0, Wt, boundary); ix1 = wrap_near(x0 == 0x7fffffff ? x0 : x0 + 1, Wt, boundary);
                iy0 = wrap_near(y0, Ht, boundary); iy1 = wrap_near(y0 == 0x7fffffff ? y0 : y0 + 1, Ht, boundary);
            }
#pragma unroll
            for (int c = 0; c < CS; ++c) {
                atomicAdd(&at32(a.grad_tex, (unsigned int)((iy0 * Wt + ix0) * CS + c)), gc[c] * w00);
                atomicAdd(&at32(a.grad_tex, (unsigned int)((iy0 * Wt + ix1) * CS + c)), gc[c] * w10);
                atomicAdd(&at32(a.grad_tex, (unsigned int)((iy1 * Wt + ix0) * CS + c)), gc[c] * w01);
                atomicAdd(&at32(a.grad_tex, (unsigned int)((iy1 * Wt + ix1) * CS + c)), gc[c] * w11);
            }
        }
    };

    // pass 0 keeps its pixel's taps until the origin is known
    float k_gc[CS], k_fx = 0.f, k_fy = 0.f;
    int k_x0 = 0x7fffffff, k_y0 = 0x7fffffff;
    bool k_on = false;

    // ---- one pixel: shade, loss, chain back.  FIRST: pass 0 (texel adds deferred to behind the origin's barrier) ----
    auto pixel = [&](const int row_pair, const bool FIRST) {
        // (seven workgroups per CU leave 72 registers: two values the compiler would otherwise hold -- and, at 72, spill -- across the whole
        //  kernel are formed where they are used: lane >> 5 here, three instructions per pixel, and the column's bit in the deferred branch)
        int upper;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0\n\tv_lshrrev_b32 %0, 5, %0" : "=v"(upper));
        const int zy = 2 * row_pair + upper, y = by0 + zy;
        const unsigned int me = s_id[(zy + 1) * OS + col + 1];
        const int id = (int)(me & 0xffffffu);      // (pixels beyond the image hold 0)
        int tkey = -1;
        int vk[3] = {0, 0, 0};
        float gv9[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (id > 0) {
            bool deferred = false;
            if (bin_sil_rec) {      // (uniform)
                const unsigned int nR = s_id[(zy + 1) * OS + col + 2], nL = s_id[(zy + 1) * OS + col];
                const unsigned int nU = s_id[(zy + 2) * OS + col + 1], nD = s_id[zy * OS + col + 1];
                deferred = (x + 1 < W && pair_maybe(me, nR)) || (y + 1 < H && pair_maybe(me, nU)) || (x > 0 && pair_maybe(me, nL)) ||
                           (y > 0 && pair_maybe(me, nD));
            }
            const int t = id - 1;
            // (measured and dropped in r4: the four pixels' vertex indices fetched up front and a pixel's vertices while the one before it is
            //  worked on -- 93 VGPRs and 5 waves per SIMD against 73 and 6, the call 2.70 ms either way)
            const I3 ti = ld32(reinterpret_cast<const I3 *>(a.tri), t);
            const float4 v0 = ld32(pos_img, ti.a), v1 = ld32(pos_img, ti.b), v2 = ld32(pos_img, ti.c);
            const float fy = s_fy[zy];
            ShadeKeep K;
            float u, v, zw = 0.0f;
            float4 db = make_float4(0.f, 0.f, 0.f, 0.f);      // MIP: screen-space derivatives of the barycentrics (rast_db never exists in HBM)
            const float sx = 2.0f / (float)W, sy = 2.0f / (float)H;
            if (MIP) {
                const Shade sd = shade_pixel(v0, v1, v2, fx_col, fy, sx, sy);
                u = sd.u; v = sd.v; zw = sd.zw;
                db = make_float4(sd.dudx, sd.dudy, sd.dvdx, sd.dvdy);
            } else {
                shade_uvz<false>(v0, v1, v2, fx_col, fy, K, u, v, zw);      // (z/w: deferred pixels only, below)
            }
            // interpolate (fit.py:157) + texture 'linear' (fit.py:158): the arithmetic of the stand-alone kernels
            // (BMODE >= 0 -- the reference's case as compile-time constants -- is launched only with the gathered table: no call, and
            //  with it no calling convention, in that kernel)
            const UV3 tq = (BMODE >= 0 || a.tri_uv) ? ld32(reinterpret_cast<const UV3 *>(a.tri_uv), t) : uv_indirect(cold(ka)->uv, cold(ka)->uv_tri, t);
            const float2 q0 = tq.q0, q1 = tq.q1, q2 = tq.q2;
            const float w = 1.0f - u - v;
            const float tu = u * q0.x + v * q1.x + w * q2.x;
            const float tv = u * q0.y + v * q1.y + w * q2.y;
            const Taps tp = MIP ? Taps{} : (boundary == FPCDR_BOUNDARY_ZERO ? make_taps(tu, tv, Ht, Wt, CS, boundary) : make_taps_fast(tu, tv, Ht, Wt, CS, boundary));
            const unsigned int poff = (unsigned int)(zy * W + col);
            const size_t off = bin_off + poff;
            const float rf = (float)ld32(a.ref + bin_off, poff);
            const float d0 = rf - bgs;
            float colv[CS], gq[CS];
            float gfx = 0.f, gfy = 0.f;
            bool nz = false;
            // MIP: interpolate(..., rast_db, diff_attrs='all') + texture('linear-mipmap-linear') (fit.py:153-155), the arithmetic of the
            // stand-alone kernels: the footprint of the texture coordinate from the barycentrics' screen derivatives
            const float e0x = q0.x - q2.x, e0y = q0.y - q2.y, e1x = q1.x - q2.x, e1y = q1.y - q2.y;
            float4 da = make_float4(0.f, 0.f, 0.f, 0.f);
            MipKeep<MIP ? CS : 1> MK;      // level arithmetic, tap sets and texels of the lookup, kept for its backward
            if (MIP) {
                da = make_float4(db.x * e0x + db.z * e1x, db.y * e0x + db.w * e1x, db.x * e0y + db.z * e1y, db.y * e0y + db.w * e1y);
                mip_lookup_fwd<MIP ? CS : 1>(ma->lv, ma->n_levels, make_float2(tu, tv), da, Ht, Wt, boundary, MK, reinterpret_cast<float (&)[MIP ? CS : 1]>(colv));
            }
#pragma unroll
            for (int c = 0; c < CS; ++c) {
                float t00 = 0.f, t10 = 0.f, t01 = 0.f, t11 = 0.f;
                if (!MIP) {
                load_taps<true>(a.tex, tp, c, CS, t00, t10, t01, t11);      // (the entry point requires < 2^30 texel values)
                mask_taps(tp, t00, t10, t01, t11);
                const float top = t00 + (t10 - t00) * tp.fx;
                const float bot = t01 + (t11 - t01) * tp.fx;
                colv[c] = top + (bot - top) * tp.fy;
                }
                // background elsewhere (fit.py:161); squared error against the 8-bit reference (fit.py:579), as the difference to a
                // background pixel (the all-background share of the loss depends on the references alone: fpcdr_ref_bg_sumsq)
                const float dd = rf - colv[c] * cs;
                lsum += dd * dd - d0 * d0;
                gq[c] = loss_grad(dd, cs, gs);
                nz |= gq[c] != 0.0f;
                gfx += gq[c] * ((t10 - t00) * (1.0f - tp.fy) + (t11 - t01) * tp.fy);
                gfy += gq[c] * ((t01 + (t11 - t01) * tp.fx) - (t00 + (t10 - t00) * tp.fx));
            }
            if (deferred) {      // k_fix reads these back: a few per cent of the covered pixels
                any_def = true;
                unsigned int colbit;      // (1 << col, not hoisted: see above)
                asm volatile("v_lshlrev_b32 %0, %1, 1" : "=v"(colbit) : "v"(col));
                atomicOr(&s_cmask[zy], colbit);
                if (!MIP) zw = shade_zw(v0, v1, v2, K.a0, K.a1, K.p0x * K.p1y - K.p0y * K.p1x);
                const size_t ro = compact ? rec_slot_index(slot, col, zy) : off;
                const KArgs kc = cold(ka);
                kc->rec[ro] = make_float4(u, v, zw, (float)(t + 1));      // (= id; t stays live for the vertex table's key, id need not)
                float *const colp = kc->color, *const gaap = kc->g_aa;
#pragma unroll
                for (int c = 0; c < CS; ++c) { colp[ro * CS + c] = colv[c]; gaap[ro * CS + c] = gq[c]; }
            }
            float gtu_m = 0.f, gtv_m = 0.f;
            float4 gda = make_float4(0.f, 0.f, 0.f, 0.f);
            if (MIP && want_grad && nz) {
                // texel gradients of the pixel's two levels through the windows of those levels (the prepass below), anything else to
                // memory; the footprint's gradient goes back through the derivative outputs of the rasteriser
                const float pu = prep_coord(tu, boundary), pv = prep_coord(tv, boundary);
                // where the pixel's taps fall in each of the three windows (uniform descriptors: no per-lane table -- a select chain on
                // `level - mlb` was compiled into a lookup table in scratch memory), then which window each of its two levels takes
                int cl[3], sl[3];
                bool il[3];
                {
                    int lx = (int)floorf(pu * (float)(Wt >> mlb) - 0.5f) - w0x, ly = (int)floorf(pv * (float)(Ht >> mlb) - 0.5f) - w0y;
                    if (boundary == FPCDR_BOUNDARY_WRAP && w0wrap) { lx = wrap_cell(lx, Wt >> mlb); ly = wrap_cell(ly, Ht >> mlb); }      // (uniform)
                    cl[0] = w0b + ly * w0s + lx; sl[0] = w0s;
                    il[0] = (unsigned int)lx < (unsigned int)(w0s - 1) && (unsigned int)ly < (unsigned int)(w0r - 1);
                }
                {
                    int lx = (int)floorf(pu * (float)(Wt >> (mlb + 1)) - 0.5f) - w1x, ly = (int)floorf(pv * (float)(Ht >> (mlb + 1)) - 0.5f) - w1y;
                    if (boundary == FPCDR_BOUNDARY_WRAP && w1wrap) { lx = wrap_cell(lx, Wt >> (mlb + 1)); ly = wrap_cell(ly, Ht >> (mlb + 1)); }      // (uniform)
                    cl[1] = w1b + ly * w1s + lx; sl[1] = w1s;
                    il[1] = (unsigned int)lx < (unsigned int)(w1s - 1) && (unsigned int)ly < (unsigned int)(w1r - 1);
                }
                {
                    int lx = (int)floorf(pu * (float)(Wt >> (mlb + 2)) - 0.5f) - w2x, ly = (int)floorf(pv * (float)(Ht >> (mlb + 2)) - 0.5f) - w2y;
                    if (boundary == FPCDR_BOUNDARY_WRAP && w2wrap) { lx = wrap_cell(lx, Wt >> (mlb + 2)); ly = wrap_cell(ly, Ht >> (mlb + 2)); }      // (uniform)
                    cl[2] = w2b + ly * w2s + lx; sl[2] = w2s;
                    il[2] = (unsigned int)lx < (unsigned int)(w2s - 1) && (unsigned int)ly < (unsigned int)(w2r - 1);
                }
                const int d0 = MK.l0 - mlb, d1 = MK.l1 - mlb;
                const bool a0 = d0 == 0 && il[0], a1 = d0 == 1 && il[1], a2 = d0 == 2 && il[2];
                const bool b0 = d1 == 0 && il[0], b1 = d1 == 1 && il[1], b2 = d1 == 2 && il[2];
                const bool in0 = a0 || a1 || a2, in1 = b0 || b1 || b2;
                const int cell0 = a0 ? cl[0] : (a1 ? cl[1] : cl[2]), st0 = a0 ? sl[0] : (a1 ? sl[1] : sl[2]);
                const int cell1 = b0 ? cl[0] : (b1 ? cl[1] : cl[2]), st1 = b0 ? sl[0] : (b1 ? sl[1] : sl[2]);
                mip_lookup_bwd<MIP ? CS : 1>(ma->lv, ma->n_levels, MK, reinterpret_cast<const float (&)[MIP ? CS : 1]>(gq), Ht, Wt, gtu_m, gtv_m, gda,
                                  [&](int level, int tap, size_t offs, int c, float vv) {
                                      const int dx = tap & 1, dy = tap >> 1;
                                      const bool first = level == MK.l0;
                                      if (first ? in0 : in1) lds_add_f64(&s_tex[((first ? cell0 : cell1) + dy * (first ? st0 : st1) + dx) * CS + c], vv);
                                      else atomicAdd(ma->lv.grad[level] + offs + c, vv);
                                  });
            }
            if (!MIP && want_tex && nz) {
                const int x0 = (int)floorf(prep_coord(tu, boundary) * (float)Wt - 0.5f);
                const int y0 = (int)floorf(prep_coord(tv, boundary) * (float)Ht - 0.5f);
                if (FIRST) {
                    k_on = true;
                    k_fx = tp.fx; k_fy = tp.fy; k_x0 = x0; k_y0 = y0;
#pragma unroll
                    for (int c = 0; c < CS; ++c) k_gc[c] = gq[c];
                } else {
                    add_taps(gq, tp.fx, tp.fy, x0, y0);
                }
            }
            if (want_pos) {
                const float mu = (boundary == FPCDR_BOUNDARY_CLAMP && !(tu >= 0.0f && tu <= 1.0f)) ? 0.0f : 1.0f;
                const float mv = (boundary == FPCDR_BOUNDARY_CLAMP && !(tv >= 0.0f && tv <= 1.0f)) ? 0.0f : 1.0f;
                const float gtu = MIP ? gtu_m * mu : gfx * (float)Wt * mu, gtv = MIP ? gtv_m * mv : gfy * (float)Ht * mv;
                const float gu = gtu * (q0.x - q2.x) + gtv * (q0.y - q2.y);
                const float gvv = gtu * (q1.x - q2.x) + gtv * (q1.y - q2.y);
                // MIP: interpolate backward of the derivative outputs, d (uv_da) / d (rast_db)
                const float4 gdb = make_float4(gda.x * e0x + gda.z * e0y, gda.y * e0x + gda.w * e0y, gda.x * e1x + gda.z * e1y, gda.y * e1x + gda.w * e1y);
                if (MIP) {
                    if (gu != 0.0f || gvv != 0.0f || gdb.x != 0.0f || gdb.y != 0.0f || gdb.z != 0.0f || gdb.w != 0.0f) {
                        tkey = t;
                        vk[0] = ti.a; vk[1] = ti.b; vk[2] = ti.c;
                        float g0[3], g1[3], g2[3];
                        shade_pixel_bwd<true>(v0, v1, v2, fx_col, fy, sx, sy, make_float4(gu, gvv, 0.f, 0.f), gdb, g0, g1, g2);
                        gv9[0] = g0[0]; gv9[1] = g0[1]; gv9[2] = g0[2];
                        gv9[3] = g1[0]; gv9[4] = g1[1]; gv9[5] = g1[2];
                        gv9[6] = g2[0]; gv9[7] = g2[1]; gv9[8] = g2[2];
                    }
                } else {
                    // the chain is LINEAR in (gu, gvv): a pixel without gradient gets exact zeros from it, so every covered pixel runs it
                    // and only the table key says whether its run has anything to add (a branch around the chain made the nine values
                    // a merge of two definitions: nine copies per pixel for a case -- no gradient at a covered pixel -- that is rare)
                    tkey = (gu != 0.0f || gvv != 0.0f) ? t : -1;
                    vk[0] = ti.a; vk[1] = ti.b; vk[2] = ti.c;
                    float g0[3], g1[3], g2[3];
                    shade_uv_bwd(K, fx_col, fy, gu, gvv, g0, g1, g2);
                    gv9[0] = g0[0]; gv9[1] = g0[1]; gv9[2] = g0[2];
                    gv9[3] = g1[0]; gv9[4] = g1[1]; gv9[5] = g1[2];
                    gv9[6] = g2[0]; gv9[7] = g2[1]; gv9[8] = g2[2];
                }
            }
        }
        if (want_pos)      // (uniform)
            wave_segment_reduce9(tkey, gv9, [&](int, const float (&sm)[9]) { vtable_add(vt, gp, vk, sm); });
    };

    if (MIP) {
        // MIP: the scatter of a pixel's texel gradients is buried in the level / footprint arithmetic, so the windows are placed by a
        // PREPASS that forms the texture coordinate and the level of the pass-0 pixels only (one pixel in four shaded twice): the finest
        // level in use, and the extent of the prepared coordinates -- which every level's window scales by its own size
        if (want_tex) {
            const int zy = 2 * shade_row_pair(0, wave) + (lane >> 5);
            const int id = (int)(s_id[(zy + 1) * OS + col + 1] & 0xffffffu);
            int l0 = 0x7fffffff, ku0 = 0x7fffffff, ku1 = (int)0x80000000, kv0 = 0x7fffffff, kv1 = (int)0x80000000;
            int su0 = 0x7fffffff, su1 = (int)0x80000000, sv0 = 0x7fffffff, sv1 = (int)0x80000000;      // (the same, half a period on)
            if (id > 0) {
                const int t = id - 1;
                const I3 ti = ld32(reinterpret_cast<const I3 *>(a.tri), t);
                const Shade sd = shade_pixel(ld32(pos_img, ti.a), ld32(pos_img, ti.b), ld32(pos_img, ti.c), fx_col, s_fy[zy], 2.0f / (float)W, 2.0f / (float)H);
                const UV3 tq = a.tri_uv ? ld32(reinterpret_cast<const UV3 *>(a.tri_uv), t) : uv_indirect(cold(ka)->uv, cold(ka)->uv_tri, t);
                const float w = 1.0f - sd.u - sd.v;
                const float tu = sd.u * tq.q0.x + sd.v * tq.q1.x + w * tq.q2.x, tv = sd.u * tq.q0.y + sd.v * tq.q1.y + w * tq.q2.y;
                const float e0x = tq.q0.x - tq.q2.x, e0y = tq.q0.y - tq.q2.y, e1x = tq.q1.x - tq.q2.x, e1y = tq.q1.y - tq.q2.y;
                const float4 da = make_float4(sd.dudx * e0x + sd.dvdx * e1x, sd.dudy * e0x + sd.dvdy * e1x, sd.dudx * e0y + sd.dvdx * e1y,
                                              sd.dudy * e0y + sd.dvdy * e1y);
                const float lev = fminf(fmaxf(compute_lod(da, Ht, Wt, 0.0f).level, 0.0f), (float)ma->n_levels);
                l0 = min((int)floorf(lev), ma->n_levels);
                const float pu = prep_coord(tu, boundary), pv = prep_coord(tv, boundary);
                ku0 = ku1 = fkey(pu);
                kv0 = kv1 = fkey(pv);
                if (boundary == FPCDR_BOUNDARY_WRAP) {      // a bin across the seam of a periodic coordinate is compact half a period on
                    su0 = su1 = fkey(pu + 0.5f - floorf(pu + 0.5f));
                    sv0 = sv1 = fkey(pv + 0.5f - floorf(pv + 0.5f));
                }
            }
            const int m0 = wave_min_dpp(l0), m1 = wave_min_dpp(ku0), m2 = ~wave_min_dpp(~ku1), m3 = wave_min_dpp(kv0), m4 = ~wave_min_dpp(~kv1);
            int m5 = 0, m6 = 0, m7 = 0, m8 = 0;
            if (boundary == FPCDR_BOUNDARY_WRAP) { m5 = wave_min_dpp(su0); m6 = ~wave_min_dpp(~su1); m7 = wave_min_dpp(sv0); m8 = ~wave_min_dpp(~sv1); }
            if (lane == 0 && m0 != 0x7fffffff) {
                atomicMin(&s_mred[0], m0); atomicMin(&s_mred[1], m1); atomicMax(&s_mred[2], m2); atomicMin(&s_mred[3], m3); atomicMax(&s_mred[4], m4);
                if (boundary == FPCDR_BOUNDARY_WRAP) { atomicMin(&s_mred[5], m5); atomicMax(&s_mred[6], m6); atomicMin(&s_mred[7], m7); atomicMax(&s_mred[8], m8); }
            }
            __syncthreads();
            mlb = __builtin_amdgcn_readfirstlane(s_mred[0]);
            if (mlb != 0x7fffffff) {      // (uniform)
                float ua = fkey_inv(__builtin_amdgcn_readfirstlane(s_mred[1])), ub = fkey_inv(__builtin_amdgcn_readfirstlane(s_mred[2]));
                float va = fkey_inv(__builtin_amdgcn_readfirstlane(s_mred[3])), vb = fkey_inv(__builtin_amdgcn_readfirstlane(s_mred[4]));
                if (boundary == FPCDR_BOUNDARY_WRAP) {
                    // the narrower of the two views of each coordinate; the shifted one, shifted back, runs from below zero to above it
                    // (window coordinates are unwrapped: the pixels find their cell modulo the level's size, the flush wraps)
                    const float ua2 = fkey_inv(__builtin_amdgcn_readfirstlane(s_mred[5])), ub2 = fkey_inv(__builtin_amdgcn_readfirstlane(s_mred[6]));
                    const float va2 = fkey_inv(__builtin_amdgcn_readfirstlane(s_mred[7])), vb2 = fkey_inv(__builtin_amdgcn_readfirstlane(s_mred[8]));
                    if (ub2 - ua2 < ub - ua) { ua = ua2 - 0.5f; ub = ub2 - 0.5f; }
                    if (vb2 - va2 < vb - va) { va = va2 - 0.5f; vb = vb2 - 0.5f; }
                }
                auto place = [&](int j, int cap, int &ox_, int &oy_, int &os_, int &or_) __attribute__((always_inline)) {
                    const int l = mlb + j;
                    if (l > ma->n_levels) return;
                    const float wl = (float)(Wt >> l), hl = (float)(Ht >> l), big = 1073741824.0f;
                    // the taps of the sampled pixels and one texel around them (rows in between shift by a fraction of a texel)
                    int xa = (int)fminf(fmaxf(floorf(ua * wl - 0.5f), -big), big) - 1, ya = (int)fminf(fmaxf(floorf(va * hl - 0.5f), -big), big) - 1;
                    const long long nw = (long long)(int)fminf(fmaxf(floorf(ub * wl - 0.5f), -big), big) + 2 - xa + 1;
                    const long long nh = (long long)(int)fminf(fmaxf(floorf(vb * hl - 0.5f), -big), big) + 2 - ya + 1;
                    int stride, rows;
                    if (nw * nh <= cap) {      // it fits: the rows that are left over go half below, half above
                        stride = (int)nw;
                        rows = cap / stride;
                        ya -= (rows - (int)nh) >> 1;
                    } else {                   // it does not (a uv seam, a second surface): a window of the footprint's aspect around its centre
                        const float aspect = fminf(fmaxf((float)nw / (float)nh, 1.0f / (float)cap), (float)cap);
                        stride = min(max((int)sqrtf((float)cap * aspect), 2), cap / 2);
                        rows = cap / stride;
                        xa += (int)((nw - stride) >> 1);
                        ya += (int)((nh - rows) >> 1);
                    }
                    // (uniform values formed on the vector unit: back into scalar registers)
                    ox_ = __builtin_amdgcn_readfirstlane(xa); oy_ = __builtin_amdgcn_readfirstlane(ya);
                    os_ = __builtin_amdgcn_readfirstlane(stride); or_ = __builtin_amdgcn_readfirstlane(rows);
                };
                place(0, MipWinCaps<CS>::N0, w0x, w0y, w0s, w0r);
                place(1, MipWinCaps<CS>::N1, w1x, w1y, w1s, w1r);
                place(2, MipWinCaps<CS>::N2, w2x, w2y, w2s, w2r);
                w0wrap = w0x < 0 || w0y < 0 || w0x + w0s > (Wt >> mlb) || w0y + w0r > (Ht >> mlb);
                w1wrap = w1x < 0 || w1y < 0 || w1x + w1s > (Wt >> (mlb + 1)) || w1y + w1r > (Ht >> (mlb + 1));
                w2wrap = w2x < 0 || w2y < 0 || w2x + w2s > (Wt >> (mlb + 2)) || w2y + w2r > (Ht >> (mlb + 2));
            }
        }
        pixel(shade_row_pair(0, wave), false);
    }
    // (non-mip) the texel window of the pixels whose taps the FIRST pass kept: reductions in s_mred[red .. red + 8]
    auto setup_window = [&](const int red) __attribute__((always_inline)) {
        // the window: the bounding box of the taps the FIRST pass kept, as a rectangle of at most OCELLS cells.  A footprint is rarely square
        // (the rig's face: 1.6 texels per pixel along v, 1.0 along u -- 52 x 33 texels under a bin), and a bin across the seam of a periodic
        // coordinate is compact only in the CENTRED view of it (x >= n / 2 counted as x - n).  (r4 form: a fixed 40 x 40 window from the
        // smallest tap; 22 % of the pixels' adds fell outside and went to memory one by one: profiles/r05_flush_experiments.txt 10)
        const bool on = k_on && k_x0 != 0x7fffffff && k_y0 != 0x7fffffff;
        const int NONE_LO = 0x7fffffff, NONE_HI = (int)0x80000000;
        {
            // (four reductions in 24 vector instructions, results in lane 63: common.h)
            int r1 = on ? k_x0 : NONE_LO, r2 = ~(on ? k_x0 : NONE_HI), r3 = on ? k_y0 : NONE_LO, r4 = ~(on ? k_y0 : NONE_HI);
            wave_min4_dpp_lane63(r1, r2, r3, r4);
            if (lane == 63 && r1 != NONE_LO) lds_minmax4(&s_mred[red + 1], r1, &s_mred[red + 2], ~r2, &s_mred[red + 3], r3, &s_mred[red + 4], ~r4);
        }
        OPROF_T(3);
        __syncthreads();
        OPROF_T(4);
        int xa = __builtin_amdgcn_readfirstlane(s_mred[red + 1]);
        ox = 0; oy = 0; ows = 1; owr = 1; owrap = false;
        if (xa != NONE_LO) {      // (uniform)
            int xb = __builtin_amdgcn_readfirstlane(s_mred[red + 2]), ya = __builtin_amdgcn_readfirstlane(s_mred[red + 3]), yb = __builtin_amdgcn_readfirstlane(s_mred[red + 4]);
            if (boundary == FPCDR_BOUNDARY_WRAP && (xb - xa >= (Wt >> 1) || yb - ya >= (Ht >> 1))) {
                // (uniform, rare) a box across half the texture: a bin on the seam.  The centred view of the taps, reduced the same way
                const int cx = k_x0 >= (Wt >> 1) ? k_x0 - Wt : k_x0, cy = k_y0 >= (Ht >> 1) ? k_y0 - Ht : k_y0;
                int r5 = on ? cx : NONE_LO, r6 = ~(on ? cx : NONE_HI), r7 = on ? cy : NONE_LO, r8 = ~(on ? cy : NONE_HI);
                wave_min4_dpp_lane63(r5, r6, r7, r8);
                if (lane == 63 && r5 != NONE_LO) lds_minmax4(&s_mred[red + 5], r5, &s_mred[red + 6], ~r6, &s_mred[red + 7], r7, &s_mred[red + 8], ~r8);
                __syncthreads();
                const int xa2 = __builtin_amdgcn_readfirstlane(s_mred[red + 5]), xb2 = __builtin_amdgcn_readfirstlane(s_mred[red + 6]);
                const int ya2 = __builtin_amdgcn_readfirstlane(s_mred[red + 7]), yb2 = __builtin_amdgcn_readfirstlane(s_mred[red + 8]);
                if (xb2 - xa2 < xb - xa) { xa = xa2; xb = xb2; }
                if (yb2 - ya2 < yb - ya) { ya = ya2; yb = yb2; }
            }
            // (taps x0 and x0 + 1; a margin of a texel around the box gained nothing: profiles/r05_flush_experiments.txt 10)
            const long long nw = (long long)xb - xa + 2, nh = (long long)yb - ya + 2;
            int stride, rows;
            long long sx0 = (long long)xa, sy0 = (long long)ya;
            if (nw * nh <= OCELLS) {      // it fits: the rows that are left over go half below, half above
                stride = (int)nw;
                rows = OCELLS / stride;
                sy0 -= (rows - (int)nh) >> 1;
            } else {                      // it does not: a window of the footprint's aspect around its centre
                const float aspect = fminf(fmaxf((float)nw / (float)nh, 1.0f / (float)OCELLS), (float)OCELLS);
                stride = min(max((int)sqrtf((float)OCELLS * aspect), 2), OCELLS / 2);
                rows = OCELLS / stride;
                sx0 += (nw - stride) >> 1;
                sy0 += (nh - rows) >> 1;
            }
            ox = __builtin_amdgcn_readfirstlane((int)sx0); oy = __builtin_amdgcn_readfirstlane((int)sy0);
            ows = __builtin_amdgcn_readfirstlane(stride); owr = __builtin_amdgcn_readfirstlane(rows);
            // (taps run from -1 to n - 1: a window from 0 on misses only the column / row of taps at -1, which then go to memory)
            owrap = ox < 0 || oy < 0 || ox + ows > Wt || oy + owr > Ht;
        }
        if (k_on) add_taps(k_gc, k_fx, k_fy, k_x0, k_y0);
    };
    // ... and its flush: every cell once (a barrier has made all adds visible); rezero: the window is used again
    auto flush_window = [&](const bool rezero) __attribute__((always_inline)) {
        if (ows <= 1) return;      // (uniform)
        // (v_rcp_f32, 1 ulp: the quotient below is at least 0.5 / ows away from an integer, the error at most rows * 1.2e-7 -- an IEEE
        //  division is fourteen vector instructions, twice per workgroup)
        const float inv = __builtin_amdgcn_rcpf((float)ows);
        const int n = ows * owr * CS;
        // (a window inside the texture -- nearly all -- holds texel (ox + lx, oy + ly) as it is: no wrapping, no clamping; uniform branch)
        const bool plain = !owrap && ox >= 0 && oy >= 0 && ox + ows <= Wt && oy + owr <= Ht;
        for (int k = tid; k < n; k += ONT) {
            const float v = (float)s_tex[k];
            if (v != 0.0f) {
                const int c = k % CS, cell = k / CS;
                const int ly = (int)(((float)cell + 0.5f) * inv), lx = cell - ly * ows;      // (exact: cell < 2^12)
                int gx = ox + lx, gy = oy + ly;
                if (!plain) { gx = wrap_near(gx, Wt, boundary); gy = wrap_near(gy, Ht, boundary); }
                atomicAdd(&at32(a.grad_tex, (unsigned int)((gy * Wt + gx) * CS + c)), v);
                if (rezero) s_tex[k] = 0.0;
            }
        }
    };
    if (MIP) {
#pragma unroll
        for (int k = 1; k < 4; ++k) pixel(shade_row_pair(k, wave), false);
    } else {
        // THE BIN IN TWO HALVES (rows 0-15, then 16-31), each with a window of its own in the same LDS: half the rows have half the
        // footprint along one axis, and 52 x 33 texels under a whole bin do not fit 1 728 cells where 27 x 33 do.  A half is two passes:
        // the first (row pairs 0, 2, 5, 7 of the half: its first and last rows are among them) keeps its taps until the window is placed;
        // the flush of the first half's window -- its atomics drain under the second half's arithmetic -- costs one more barrier.
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            k_on = false; k_x0 = 0x7fffffff; k_y0 = 0x7fffffff;
            pixel(8 * half + (int)((0x7520u >> (4 * wave)) & 15u), true);
            if (want_tex) setup_window(9 * half);
            pixel(8 * half + (int)((0x6431u >> (4 * wave)) & 15u), false);
            if (half == 0 && want_tex) {
                __syncthreads();
                flush_window(true);
            }
        }
    }

    lsum = wave_sum_dpp(lsum);
    if (lane == 0) s_lpart[wave] = lsum;
    OPROF_T(5);
    const bool bin_def = block4_any(s_vote + 4, any_def);
    OPROF_T(6);
    // ---- the bin's deferred pixels for k_fix; loss ----
    const KArgs ke = cold(ka);      // (the epilogue's pointers: scalar loads here, not registers held across the shading code)
    if (tid < OB) { ke->cmask[bin_lin * OB + tid] = s_cmask[tid]; ke->hitmask[bin_lin * OB + tid] = 0u; }
    if (tid == 0) {
        if (bin_def) {
            ke->binflag[bin_lin] = 1;      // (zero-filled by the call's first kernel)
            // k_fix runs over the bins with a deferred pixel only -- one in six of the occupied ones on a face rig.  (One-wave workgroups
            // that leave at once are not free: k_fix<1> over ALL occupied bins was bound by workgroup dispatch, 220 us for 90 k of them.)
            ke->def_list[atomicAdd(ke->def_count, 1)] = (int32_t)bin_lin;
        }
        const double tot = (double)s_lpart[0] + (double)s_lpart[1] + (double)s_lpart[2] + (double)s_lpart[3];
        const unsigned int slot = ((unsigned int)bxi + 31u * (unsigned int)byi + 977u * (unsigned int)b) % FPCDR_LOSS_SLOTS;
        if (tot != 0.0) atomicAdd(ke->loss_sum + slot, tot);
    }
    OPROF_ADD(0, 0, 1); OPROF_ADD(1, 1, 2); OPROF_ADD(2, 2, 3); OPROF_ADD(3, 3, 4); OPROF_ADD(4, 4, 5); OPROF_ADD(5, 5, 6);
    if (!want_grad) return;
    // ---- flush: every vertex slot and window cell once (the barrier above has made all adds visible) ----
    if (want_pos) vtable_flush(vt, gp, tid, ONT);
    if (!MIP && want_tex) flush_window(false);
    if (MIP && want_tex) {
        auto flush = [&](int wx, int wy, int ws, int wr, int wb, int level) __attribute__((always_inline)) {
            if (ws <= 1) return;      // (uniform: an unused window)
            const int wl = Wt >> level, hl = Ht >> level;
            const float inv = 1.0f / (float)ws;
            float *const g = ma->lv.grad[level];
            const int n = ws * wr * CS;
            const bool plain = wx >= 0 && wy >= 0 && wx + ws <= wl && wy + wr <= hl;      // (uniform) the window lies inside its level
            for (int k = tid; k < n; k += ONT) {
                const float v = (float)s_tex[wb * CS + k];
                if (v != 0.0f) {
                    const int c = k % CS, cell = k / CS;
                    const int ly = (int)(((float)cell + 0.5f) * inv), lx = cell - ly * ws;      // (exact: cell < 2^12)
                    int gx = wx + lx, gy = wy + ly;
                    if (!plain) { gx = wrap_near(gx, wl, boundary); gy = wrap_near(gy, hl, boundary); }
                    atomicAdd(g + (size_t)(gy * wl + gx) * CS + c, v);
                }
            }
        };
        flush(w0x, w0y, w0s, w0r, w0b, mlb);
        flush(w1x, w1y, w1s, w1r, w1b, mlb + 1);
        flush(w2x, w2y, w2s, w2r, w2b, mlb + 2);
    }
    OPROF_T(7);
    OPROF_ADD(6, 6, 7);
#ifdef FPCDR_OPROF
    if ((threadIdx.x & 63) == 0) atomicAdd(&g_oprof[7], 1ull);
#endif
}

// (r5, measured and dropped: SEVERAL list entries per workgroup, one after the other, so that a bin's flush atomics -- for which s_endpgm
//  waits -- drain while the next bin is shaded: 1 597 us with one bin per workgroup, 1 630 with two in a loop (96 VGPRs: the loop's hoisted
//  scalar loads; arguments re-read from the kernel-argument segment per trip), 1 615 with two unrolled (84 VGPRs, and 80 with five spills),
//  1 639 with four.  What the flush costs is not the idle slot: profiles/r05_flush_experiments.txt.)
// ONE kernel parameter, a struct: the position of ObjArgs in the kernel-argument segment is then offsetof(K, a) (kernarg_objargs)
struct ShadeK { const int32_t *list; const int32_t *count; int cap, OX, OY; fpcdr_bin_decode dc; ObjArgs a; };      // cap: of a sweep, its first entry
struct ShadeMipK { const int32_t *list; const int32_t *count; int cap, OX, OY; fpcdr_bin_decode dc; ObjArgs a; MipO ma; };
template <int CS, int BMODE>
__global__ void __launch_bounds__(ONT) FPCDR_SHADE_WPE k_shade_list(const ShadeK k) {
    const int item = fpcdr_list_item(*k.count, k.cap);      // (XCD x takes the x-th eighth of the entries: common.h)
    if (item < 0) return;
    const int lin = __builtin_amdgcn_readfirstlane(k.list[item]);
    int b, byi, bxi;
    fpcdr_decode_bin(lin, k.dc, b, byi, bxi);
    shade_body<CS, BMODE>(b, bxi, byi, k.OX, k.OY, k.a, kernarg_objargs<ShadeK>());
}

// the MIP instantiation (boundary mode at run time): list form and strided sweep
template <int CS>
__global__ void __launch_bounds__(ONT) k_shade_mip_list(const ShadeMipK k) {
    const int item = fpcdr_list_item(*k.count, k.cap);
    if (item < 0) return;
    const int lin = __builtin_amdgcn_readfirstlane(k.list[item]);
    int b, byi, bxi;
    fpcdr_decode_bin(lin, k.dc, b, byi, bxi);
    shade_body<CS, -1, true>(b, bxi, byi, k.OX, k.OY, k.a, kernarg_objargs<ShadeMipK>(), &k.ma);
}
// (the level chain as a kernel parameter of its own: as a member of the one struct, indexed at run time inside the loop, it was copied
//  to a 48-byte private segment -- and a kernel with a private segment is dispatched several times slower, taken or not)
template <int CS>
__global__ void __launch_bounds__(ONT) k_shade_mip_queue(const ShadeK k, const MipO ma) {
    const int n = *k.count;
    const KArgs ka = kernarg_objargs<ShadeK>();
    for (int item = k.cap + blockIdx.x; item < n; item += gridDim.x) {
        const int lin = __builtin_amdgcn_readfirstlane(k.list[item]);
        int b, byi, bxi;
        fpcdr_decode_bin(lin, k.dc, b, byi, bxi);
        shade_body<CS, -1, true>(b, bxi, byi, k.OX, k.OY, k.a, ka, &ma);
        __syncthreads();
    }
}

// strided sweep of the entries beyond the hinted launch (normally none; scalar loop variable: see k_bins_queue in rasterize.hip)
template <int CS>
__global__ void __launch_bounds__(ONT) k_shade_queue(const ShadeK k) {
    const int n = *k.count;
    for (int item = k.cap + blockIdx.x; item < n; item += gridDim.x) {
        const int lin = __builtin_amdgcn_readfirstlane(k.list[item]);
        int b, byi, bxi;
        fpcdr_decode_bin(lin, k.dc, b, byi, bxi);
        shade_body<CS, -1>(b, bxi, byi, k.OX, k.OY, k.a, kernarg_objargs<ShadeK>());
        __syncthreads();     // the next bin's first LDS writes must not overtake this bin's last LDS reads
    }
}

// ------------------------------------------------------------------------------------------------
// k_fix: the deferred pixels of one bin, one wave per bin.
//   PASS 0  antialias forward (gather form, the pair analysis of aa_pairs.h) for each deferred pixel: blended colour, correction of
//           its loss term, final d loss / d colour.  Reads records / colours k_shade stored, writes grad_aa of its own pixels only.
//   PASS 1  (separate launch: it reads the neighbours' FINAL grad_aa) gradient arriving at each deferred pixel's colour through the
//           antialias op minus the un-antialiased one k_shade already scattered -> texture / interpolate / rasterize backward of the
//           difference; the antialias op's d alpha / d pos; the share of empty pixels' colour (all sample uv = (0,0)).
// Neighbour ids come from the id planes (own bin or a neighbour's), z/w and colour of a partner from the deferred records: both pixels
// of a pair that passes pair_maybe are deferred.
// Small open-addressed LDS accumulators of k_fix<1>: key -> NV doubles.  The few dozen pixels of a bin that took part in a blend share
// vertices and texels: scattered float atomics retire slowly (the pass spent its time on ~650 of them per bin), so they are summed in
// LDS first and every slot is flushed once.  A full table falls back to memory.
constexpr int ESLOTS = 32;               // slots of the empty pixels' colour gradient (see k_objective_finish)
constexpr int FVS = 64, FTS = 128;       // vertex / texel slots (LDS per wave decides how many bins a CU has in flight: 5.5 KB -> 28)
__device__ __forceinline__ int hacc_slot(int *keys, int mask, int key) {
    unsigned int slot = (((unsigned int)key * 2654435761u) >> 16) & (unsigned int)mask;
    for (int probe = 0; probe <= mask; ++probe) {
        const int o = atomicCAS(&keys[slot], -1, key);
        if (o == -1 || o == key) return (int)slot;
        slot = (slot + 1) & (unsigned int)mask;
    }
    return -1;
}

// (x, y, w) gradient of vertex `key` into the LDS table, or to memory when the table is full
__device__ __forceinline__ void hacc_vadd(float *gp, int *vkeys, double *vacc, int key, float gx, float gy, float gw) {
    const int sl = hacc_slot(vkeys, FVS - 1, key);
    if (sl >= 0) { lds_add_f64(&vacc[3 * sl], gx); lds_add_f64(&vacc[3 * sl + 1], gy); lds_add_f64(&vacc[3 * sl + 2], gw); }
    else { atomicAdd(gp + 4 * (size_t)key, gx); atomicAdd(gp + 4 * (size_t)key + 1, gy); atomicAdd(gp + 4 * (size_t)key + 3, gw); }
}

// d (blend weight) / d pos of one active edge, G = d loss / d (blend weight) (see k_aa_bwd_fix in antialias.hip).  OUT OF LINE: the
// pair analysis is inlined at twelve sites (four pairs x three edges) and this tail with its table adds made k_fix<1> 220 us of
// instruction fetch and registers (99 VGPRs) for a few dozen pixels per bin.
struct EdgeVals { float Lx, Ly, qax, qay, wa, qbx, qby, wb; };      // (by value: see uv_indirect)
__device__ __noinline__ void aa_edge_pos_grad(float *gp, int *vkeys, double *vacc, EdgeVals ev, float t, float s, int d, float G,
                                              int Px, int Py, int va, int vb, float hw, float hh) {
    const float Ld = d == 0 ? ev.Lx : ev.Ly;
    const float gLz = -G / (s * Ld);
    const float gLd = -G * t / Ld;
    const float gLx = d == 0 ? gLd : 0.0f, gLy = d == 0 ? 0.0f : gLd;
    float g_qax = 0.f, g_qay = 0.f, g_wa = 0.f, g_qbx = 0.f, g_qby = 0.f, g_wb = 0.f;
    g_qay += gLx * ev.wb; g_wb += gLx * ev.qay; g_wa -= gLx * ev.qby; g_qby -= gLx * ev.wa;
    g_wa += gLy * ev.qbx; g_qbx += gLy * ev.wa; g_qax -= gLy * ev.wb; g_wb -= gLy * ev.qax;
    g_qax += gLz * ev.qby; g_qby += gLz * ev.qax; g_qay -= gLz * ev.qbx; g_qbx -= gLz * ev.qay;
    const float fxp = (float)Px + 0.5f - hw, fyp = (float)Py + 0.5f - hh;
    hacc_vadd(gp, vkeys, vacc, va, g_qax * hw, g_qay * hh, g_wa - fxp * g_qax - fyp * g_qay);
    hacc_vadd(gp, vkeys, vacc, vb, g_qbx * hw, g_qby * hh, g_wb - fxp * g_qbx - fyp * g_qby);
}

template <int CS, int PASS, bool MIP = false>
__device__ __forceinline__ void fix_body(const int b, const int bxi, const int byi, const int OX, const int OY, const ObjArgs &a,
                                         const MipO *ma = nullptr) {
    __shared__ unsigned int s_mask[OB];
    __shared__ unsigned short s_list[OB * OB];
    __shared__ int s_n;
    __shared__ int s_vk[PASS == 1 ? FVS : 1], s_tk[PASS == 1 ? FTS : 1];
    __shared__ double s_va[PASS == 1 ? FVS * 3 : 1], s_ta[PASS == 1 ? FTS * CS : 1];
    const size_t bin_lin = ((size_t)b * OY + byi) * OX + bxi;
#ifdef FPCDR_OPROF
    const long long ft0 = clock64();
#endif
    if (!__builtin_amdgcn_readfirstlane((int)a.binflag[bin_lin])) return;      // no deferred pixel in this bin
    const int tid = threadIdx.x, lane = tid & 63;
    const int bx0 = bxi * OB, by0 = byi * OB;
    const int H = a.H, W = a.W;
    const unsigned int wmask = (unsigned int)__builtin_amdgcn_readfirstlane((int)a.occ[bin_lin]);
    // PASS 1 visits only the pixels PASS 0 found in a blend (as the blended pixel or as its partner): for every other deferred pixel the
    // gradient through the antialias op IS the un-antialiased one, a difference of exactly zero
    if (tid < OB) s_mask[tid] = PASS == 0 ? a.cmask[bin_lin * OB + tid] : a.hitmask[bin_lin * OB + tid];
    if (tid == 0) s_n = 0;
    if (PASS == 1) {
        for (int k = tid; k < FVS; k += FNT) { s_vk[k] = -1; s_va[3 * k] = 0.0; s_va[3 * k + 1] = 0.0; s_va[3 * k + 2] = 0.0; }
        for (int k = tid; k < FTS; k += FNT) {
            s_tk[k] = -1;
#pragma unroll
            for (int c = 0; c < CS; ++c) s_ta[k * CS + c] = 0.0;
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < OB * OB / FNT; ++k) {
        const int pix = k * FNT + tid;
        const bool c = (s_mask[pix >> 5] >> (pix & 31)) & 1u;
        const unsigned long long bal = __ballot(c);
        int base = 0;
        if (lane == 0 && bal) base = atomicAdd(&s_n, __popcll(bal));
        base = __builtin_amdgcn_readfirstlane(base);
        if (c) s_list[base + __popcll(bal & ((1ull << lane) - 1ull))] = (unsigned short)pix;
    }
    __syncthreads();
    const int n = __builtin_amdgcn_readfirstlane(s_n);
#ifdef FPCDR_OPROF
    if (tid == 0) { atomicAdd(&g_oprof[8 + 2 * PASS], (unsigned long long)n); atomicAdd(&g_oprof[9 + 2 * PASS], 1ull); }
#endif
    if (n == 0) return;
#ifdef FPCDR_OPROF
    const long long ft1 = clock64();
#endif
    float ecol[CS];
#pragma unroll
    for (int c = 0; c < CS; ++c) ecol[c] = a.empty_color[c];
    const size_t img = (size_t)b * H * W;
    const AAGeom g = {a.pos + (size_t)b * a.V, a.tri, a.sil + (size_t)b * a.T, a.T, W, H, 0.5f * (float)W, 0.5f * (float)H};
    const float cs = a.color_scale, gs = a.grad_scale, bgs = a.bg * cs;
    float *const gp = (PASS == 1 && a.grad_pos) ? a.grad_pos + (size_t)b * a.V * 4 : nullptr;
    // (x, y, w) gradient of vertex `key` / gradient of the texel at element offset `off` (+ channel c): into the LDS tables
    auto vadd = [&](int key, float gx, float gy, float gw) { hacc_vadd(gp, s_vk, s_va, key, gx, gy, gw); };
    auto tadd = [&](int off, const float (&v)[CS], float wgt) {      // off = texel index * CS
        const int sl = hacc_slot(s_tk, FTS - 1, off);
        if (sl >= 0) {
#pragma unroll
            for (int c = 0; c < CS; ++c) lds_add_f64(&s_ta[sl * CS + c], v[c] * wgt);
        } else {
#pragma unroll
            for (int c = 0; c < CS; ++c) atomicAdd(a.grad_tex + off + c, v[c] * wgt);
        }
    };
    // where pixel (xx, yy) of this bin or of one of its eight neighbours keeps its record / colour / gradient
    const bool compact = a.pool_cap > 0;      // (uniform)
    const int my_slot = compact ? __builtin_amdgcn_readfirstlane(a.slot_of[bin_lin]) : 0;
    auto rix = [&](int xx, int yy) -> size_t {
        if (!compact) return img + (size_t)yy * W + xx;
        const int dbx = (xx >> 5) - bxi, dby = (yy >> 5) - byi;
        const int sl = (dbx | dby) ? a.slot_of[(size_t)((long long)bin_lin + dby * OX + dbx)] : my_slot;
        return rec_slot_index(sl, xx, yy);
    };
    auto id_at = [&](int xx, int yy) -> unsigned int {      // entry of an in-image pixel of this bin or of one of its eight neighbours
        const int dbx = (xx >> 5) - bxi, dby = (yy >> 5) - byi;
        if (!((wmask >> ((dby + 1) * 4 + dbx + 1)) & 1u)) return 0u;      // a bin that was not rasterised holds empty pixels
        return a.idp[(size_t)((long long)bin_lin + dby * OX + dbx) * (OB * OB) + (yy & 31) * OB + (xx & 31)];
    };
    float lsum = 0.0f;
    float esum[CS];
#pragma unroll
    for (int c = 0; c < CS; ++c) esum[c] = 0.0f;
    // A lane takes ONE of the four pairs of a pixel -- lanes 4 i .. 4 i + 3: pixel i towards right, up, left, down -- and the quad's
    // sums are formed with two DPP steps: the pair analysis is a chain of six dependent loads, and one lane walking its pixel's four
    // pairs one after the other made k_fix<1> 210 us for three pixels per wave.  The rest of a pixel is its first lane's.
    auto quad_sum = [](float v) {
        v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false));      // quad_perm [1,0,3,2]
        v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, false));      // quad_perm [2,3,0,1]
        return v;
    };
    for (int base = 0; base < 4 * n; base += FNT) {      // (uniform trip count: the DPP steps and the segmented scan need the whole wave)
        const int i = base + tid;
        const bool act = i < 4 * n;
        const int pix = act ? s_list[i >> 2] : 0;
        const int dir = i & 3;
        const bool lead = act && dir == 0;
        const int x = bx0 + (pix & 31), y = by0 + (pix >> 5);
        int tkey = -1;
        int vk[3] = {0, 0, 0};
        float gv9[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        const size_t off = img + (size_t)y * W + x;      // (the pixel in the image: reference, flag planes)
        const size_t ro = act ? rix(x, y) : 0;           // (... and in the record arrays)
        unsigned int me = 0u;
        float4 rme = make_float4(0.f, 0.f, 0.f, 0.f);
        float cme[CS], acc[CS], go[CS];      // acc / go: THIS lane's pair's share (summed over the quad below)
#pragma unroll
        for (int c = 0; c < CS; ++c) { cme[c] = 0.0f; acc[c] = 0.0f; go[c] = 0.0f; }
        bool hit = false;
        if (act) {
            me = a.idp[bin_lin * (OB * OB) + pix];
            rme = a.rec[ro];
            const float zX = rme.z;
#pragma unroll
            for (int c = 0; c < CS; ++c) cme[c] = a.color[ro * CS + c];
            // pair (x0, y0) - (x0 + e_d): entries / depths in pair order; (ox, oy, eo) = the partner of X; first = X is the pair's first pixel
            auto visit = [&](int x0, int y0, int d, unsigned int e0, float z0, unsigned int e1, float z1, int ox, int oy, unsigned int eo,
                             bool first) {
                bool hit_o = false;      // PASS 0: X was blended with this (covered) partner
                const bool blended = for_active_edges(g, x0, y0, d, (int)(e0 & 0xffffffu), z0, (int)(e1 & 0xffffffu), z1,
                    [&](float t, int Px, int Py, int Qx, int Qy, int va, int vb, const EdgeEval &ev, float s) {
                        const bool far = t >= 0.5f;
                        const int rx = far ? Qx : Px, ry = far ? Qy : Py;      // the pixel that is blended
                        const float amt = far ? t - 0.5f : 0.5f - t;
                        const bool rX = rx == x && ry == y;
                        const bool o_cov = (eo & 0xffffffu) != 0u;
                        // (the partner's colour through an INDEX that is valid either way -- element 0 for an empty partner, whose bin may have no
                        //  slot --: a pointer selected between the array and `ecol` put `ecol` into scratch memory)
                        const size_t ci = o_cov ? rix(ox, oy) * CS : 0;
                        if (PASS == 0) {
                            if (!rX) return;
                            hit = true;
                            hit_o |= o_cov;
#pragma unroll
                            for (int c = 0; c < CS; ++c) { const float cv = a.color[ci + c]; acc[c] += amt * ((o_cov ? cv : ecol[c]) - cme[c]); }
                            return;
                        }
                        if (!rX && !o_cov) return;      // the blended pixel is an empty one: no gradient arrives
                        float gr[CS];
                        const size_t rr = rX ? ro : rix(rx, ry);
#pragma unroll
                        for (int c = 0; c < CS; ++c) gr[c] = a.g_aa[rr * CS + c];
                        // (branch-free: with `if (rX) go -= .. else go += ..` the compiler addressed go / esum through a selected pointer
                        //  and kept both in scratch memory)
                        const float sa = rX ? -amt : amt, ea = (rX && !o_cov) ? amt : 0.0f;
#pragma unroll
                        for (int c = 0; c < CS; ++c) { go[c] += sa * gr[c]; esum[c] += ea * gr[c]; }
                        // d alpha / d pos: once per pair, by its first pixel -- by the second if the first is empty (never visited)
                        if (!gp || !(first || !o_cov)) return;
                        const bool PisX = Px == x && Py == y;
                        float G = 0.f;
#pragma unroll
                        for (int c = 0; c < CS; ++c) {
                            const float cv = a.color[ci + c];
                            const float cO = o_cov ? cv : ecol[c];
                            G += gr[c] * (PisX ? cme[c] - cO : cO - cme[c]);
                        }
                        if (G == 0.0f) return;
                        const EdgeVals evv = {ev.Lx, ev.Ly, ev.qax, ev.qay, ev.wa, ev.qbx, ev.qby, ev.wb};
                        aa_edge_pos_grad(gp, s_vk, s_va, evv, t, s, d, G, Px, Py, va, vb, g.hw, g.hh);
                    });
                if (PASS == 0 && hit_o) {      // the partner's colour reaches the loss through X: it needs PASS 1 as well
                    const size_t obin = (size_t)((long long)bin_lin + ((oy >> 5) - byi) * OX + ((ox >> 5) - bxi));
                    atomicOr(a.hitmask + obin * OB + (oy & 31), 1u << (ox & 31));
                }
                // diagnostics: bit of the pair's first pixel in plane d = "this pair was blended" (set by the first pixel, or by the
                // second when the first is empty and never visited)
                if (PASS == 0 && a.flags && blended && (first || (eo & 0xffffffu) == 0u)) {
                    const int Wq = FPCDR_AA_ROW_WORDS(W);
                    atomicOr(a.flags + (size_t)d * a.B * H * Wq + ((size_t)b * H + y0) * Wq + (x0 >> 6), 1ull << (x0 & 63));
                }
            };
            {      // this lane's pair of X: right, up, left or down
                const int nx = x + (dir == 0 ? 1 : (dir == 2 ? -1 : 0)), ny = y + (dir == 1 ? 1 : (dir == 3 ? -1 : 0));
                if (nx >= 0 && nx < W && ny >= 0 && ny < H) {
                    const unsigned int e = id_at(nx, ny);
                    if (pair_maybe(me, e)) {
                        const float ze = (e & 0xffffffu) ? a.rec[rix(nx, ny)].z : 0.0f;
                        const bool first = dir < 2;      // X is the first pixel of its pairs towards +x / +y
                        visit(first ? x : nx, first ? y : ny, dir & 1, first ? me : e, first ? zX : ze, first ? e : me, first ? ze : zX, nx, ny, e, first);
                    }
                }
            }
        }
        // the pixel's sums over its four pairs, in every lane of the quad
        const bool hit_any = (__ballot(hit) >> (lane & ~3)) & 0xFull;
#pragma unroll
        for (int c = 0; c < CS; ++c) { acc[c] = quad_sum(acc[c]); go[c] = quad_sum(go[c]); }
        if (lead) {
#pragma unroll
            for (int c = 0; c < CS; ++c) { acc[c] += cme[c]; if (PASS == 1) go[c] += a.g_aa[ro * CS + c]; }
            const float rf = (float)a.ref[off];
            if (PASS == 0) {
                if (hit_any) {      // the antialiased colour replaces the plain one in this pixel's loss term and gradient
                    atomicOr(a.hitmask + bin_lin * OB + (pix >> 5), 1u << (pix & 31));
                    const float d0 = rf - bgs;
#pragma unroll
                    for (int c = 0; c < CS; ++c) {
                        const float dn = rf - acc[c] * cs, dd = rf - cme[c] * cs;
                        lsum += dn * dn - d0 * d0;
                        lsum -= dd * dd - d0 * d0;
                        a.g_aa[ro * CS + c] = loss_grad(dn, cs, gs);
                    }
                }
            } else {
                // what k_shade scattered for this pixel was the un-antialiased gradient: chain the difference
                float dl[CS];
                bool nz = false;
#pragma unroll
                for (int c = 0; c < CS; ++c) { dl[c] = go[c] - loss_grad(rf - cme[c] * cs, cs, gs); nz |= dl[c] != 0.0f; }
                if (nz && MIP) {
                    // the mip-mapped lookup (fit.py:153-155): footprint from the rasteriser's derivatives (recomputed), texel gradients of
                    // every level straight to memory (a few dozen pixels per bin), the footprint's gradient back through the derivatives
                    const int boundary = a.boundary, Ht = a.Ht, Wt = a.Wt;
                    const int t = (int)(me & 0xffffffu) - 1;
                    const UV3 tq = a.tri_uv ? reinterpret_cast<const UV3 *>(a.tri_uv)[t] : uv_indirect(a.uv, a.uv_tri, t);
                    const float2 q0 = tq.q0, q1 = tq.q1, q2 = tq.q2;
                    vk[0] = a.tri[3 * t]; vk[1] = a.tri[3 * t + 1]; vk[2] = a.tri[3 * t + 2];
                    const float4 p0 = g.pos[vk[0]], p1 = g.pos[vk[1]], p2 = g.pos[vk[2]];
                    const float fx = (2.0f * (float)x + 1.0f) / (float)W - 1.0f, fy = (2.0f * (float)y + 1.0f) / (float)H - 1.0f;
                    const float sx = 2.0f / (float)W, sy = 2.0f / (float)H;
                    const Shade sd = shade_pixel(p0, p1, p2, fx, fy, sx, sy);
                    const float w = 1.0f - rme.x - rme.y;
                    const float tu = rme.x * q0.x + rme.y * q1.x + w * q2.x;
                    const float tv = rme.x * q0.y + rme.y * q1.y + w * q2.y;
                    const float e0x = q0.x - q2.x, e0y = q0.y - q2.y, e1x = q1.x - q2.x, e1y = q1.y - q2.y;
                    const float4 da = make_float4(sd.dudx * e0x + sd.dvdx * e1x, sd.dudy * e0x + sd.dvdy * e1x,
                                                  sd.dudx * e0y + sd.dvdx * e1y, sd.dudy * e0y + sd.dvdy * e1y);
                    float gtu = 0.f, gtv = 0.f, gbias = 0.f;
                    float4 gda = make_float4(0.f, 0.f, 0.f, 0.f);
                    mip_sample_bwd(ma->lv, 0, ma->n_levels, make_float2(tu, tv), true, da, 0.0f, Ht, Wt, CS, true, boundary, dl, gtu, gtv, gda, gbias);
                    if (gp) {
                        const float mu = (boundary == FPCDR_BOUNDARY_CLAMP && !(tu >= 0.0f && tu <= 1.0f)) ? 0.0f : 1.0f;
                        const float mv = (boundary == FPCDR_BOUNDARY_CLAMP && !(tv >= 0.0f && tv <= 1.0f)) ? 0.0f : 1.0f;
                        gtu *= mu; gtv *= mv;
                        const float gu = gtu * e0x + gtv * e0y, gvv = gtu * e1x + gtv * e1y;
                        const float4 gdb = make_float4(gda.x * e0x + gda.z * e0y, gda.y * e0x + gda.w * e0y, gda.x * e1x + gda.z * e1y, gda.y * e1x + gda.w * e1y);
                        if (gu != 0.0f || gvv != 0.0f || gdb.x != 0.0f || gdb.y != 0.0f || gdb.z != 0.0f || gdb.w != 0.0f) {
                            float g0[3], g1[3], g2[3];
                            shade_pixel_bwd<true>(p0, p1, p2, fx, fy, sx, sy, make_float4(gu, gvv, 0.f, 0.f), gdb, g0, g1, g2);
                            gv9[0] = g0[0]; gv9[1] = g0[1]; gv9[2] = g0[2];
                            gv9[3] = g1[0]; gv9[4] = g1[1]; gv9[5] = g1[2];
                            gv9[6] = g2[0]; gv9[7] = g2[1]; gv9[8] = g2[2];
                            tkey = t;
                        }
                    }
                } else if (nz) {
                    const int boundary = a.boundary, Ht = a.Ht, Wt = a.Wt;
                    const int t = (int)(me & 0xffffffu) - 1;
                    const UV3 tq = a.tri_uv ? reinterpret_cast<const UV3 *>(a.tri_uv)[t] : uv_indirect(a.uv, a.uv_tri, t);
                    const float2 q0 = tq.q0, q1 = tq.q1, q2 = tq.q2;
                    const float w = 1.0f - rme.x - rme.y;
                    const float tu = rme.x * q0.x + rme.y * q1.x + w * q2.x;
                    const float tv = rme.x * q0.y + rme.y * q1.y + w * q2.y;
                    const Taps tp = boundary == FPCDR_BOUNDARY_ZERO ? make_taps(tu, tv, Ht, Wt, CS, boundary) : make_taps_fast(tu, tv, Ht, Wt, CS, boundary);
                    const float w00 = (1.0f - tp.fx) * (1.0f - tp.fy), w10 = tp.fx * (1.0f - tp.fy);
                    const float w01 = (1.0f - tp.fx) * tp.fy, w11 = tp.fx * tp.fy;
                    float gfx = 0.f, gfy = 0.f;
#pragma unroll
                    for (int c = 0; c < CS; ++c) {
                        const float gc = dl[c];
                        float t00, t10, t01, t11;
                        load_taps(a.tex, tp, c, CS, t00, t10, t01, t11);
                        mask_taps(tp, t00, t10, t01, t11);
                        gfx += gc * ((t10 - t00) * (1.0f - tp.fy) + (t11 - t01) * tp.fy);
                        gfy += gc * ((t01 + (t11 - t01) * tp.fx) - (t00 + (t10 - t00) * tp.fx));
                    }
                    if (a.grad_tex) {      // (boundary mode 'zero': the padding receives no gradient)
                        if (tp.valid & 1u) tadd(tp.i00, dl, w00);
                        if (tp.valid & 2u) tadd(tp.i10, dl, w10);
                        if (tp.valid & 4u) tadd(tp.i01, dl, w01);
                        if (tp.valid & 8u) tadd(tp.i11, dl, w11);
                    }
                    if (gp) {
                        const float mu = (boundary == FPCDR_BOUNDARY_CLAMP && !(tu >= 0.0f && tu <= 1.0f)) ? 0.0f : 1.0f;
                        const float mv = (boundary == FPCDR_BOUNDARY_CLAMP && !(tv >= 0.0f && tv <= 1.0f)) ? 0.0f : 1.0f;
                        const float gtu = gfx * (float)Wt * mu, gtv = gfy * (float)Ht * mv;
                        const float gu = gtu * (q0.x - q2.x) + gtv * (q0.y - q2.y);
                        const float gvv = gtu * (q1.x - q2.x) + gtv * (q1.y - q2.y);
                        if (gu != 0.0f || gvv != 0.0f) {
                            vk[0] = a.tri[3 * t]; vk[1] = a.tri[3 * t + 1]; vk[2] = a.tri[3 * t + 2];
                            const float fx = (2.0f * (float)x + 1.0f) / (float)W - 1.0f;
                            const float fy = (2.0f * (float)y + 1.0f) / (float)H - 1.0f;
                            float g0[3], g1[3], g2[3];
                            shade_pixel_bwd<false>(g.pos[vk[0]], g.pos[vk[1]], g.pos[vk[2]], fx, fy, 2.0f / (float)W, 2.0f / (float)H,
                                                   make_float4(gu, gvv, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f), g0, g1, g2);
                            gv9[0] = g0[0]; gv9[1] = g0[1]; gv9[2] = g0[2];
                            gv9[3] = g1[0]; gv9[4] = g1[1]; gv9[5] = g1[2];
                            gv9[6] = g2[0]; gv9[7] = g2[1]; gv9[8] = g2[2];
                            tkey = t;
                        }
                    }
                }
            }
        }
        if (PASS == 1 && gp)      // (uniform) pixels of one triangle are often neighbours in the list: sum their runs, then nine atomics per run
            wave_segment_reduce9(tkey, gv9, [&](int, const float (&sm)[9]) {
#pragma unroll
                for (int kk = 0; kk < 3; ++kk) vadd(vk[kk], sm[3 * kk], sm[3 * kk + 1], sm[3 * kk + 2]);
            });
    }
#ifdef FPCDR_OPROF
    if (PASS == 1 && tid == 0) { const long long ft2 = clock64(); atomicAdd(&g_oprof[12], (unsigned long long)(ft1 - ft0)); atomicAdd(&g_oprof[13], (unsigned long long)(ft2 - ft1)); atomicAdd(&g_oprof[14], 1ull); }
#endif
    if (PASS == 0) {
        lsum = wave_sum_dpp(lsum);
        if (lane == 0 && lsum != 0.0f) {
            const unsigned int slot = ((unsigned int)bxi + 31u * (unsigned int)byi + 977u * (unsigned int)b + 128u) % FPCDR_LOSS_SLOTS;
            atomicAdd(a.loss_sum + slot, (double)lsum);
        }
    } else {
        __syncthreads();
        // flush of the tables: every vertex slot and texel slot once
        if (gp)
            for (int k = tid; k < FVS * 4; k += FNT) {
                const int sl = k >> 2, comp = k & 3, key = s_vk[sl];
                if (key >= 0 && comp != 2) {      // (x, y, -, w)
                    const float v = (float)s_va[3 * sl + (comp == 3 ? 2 : comp)];
                    if (v != 0.0f) atomicAdd(gp + 4 * (size_t)key + comp, v);
                }
            }
        if (a.grad_tex)
            for (int k = tid; k < FTS * CS; k += FNT) {
                const int key = s_tk[k / CS];
                if (key >= 0) {
                    const float v = (float)s_ta[k];
                    if (v != 0.0f) atomicAdd(a.grad_tex + key + k % CS, v);
                }
            }
    }
    if (PASS == 1 && a.grad_tex) {
        // empty pixels blended into covered ones: they ALL sample uv = (0,0), i.e. the same four texels with the same weights.  Every
        // wave adding to those four addresses itself serialised ~55 k atomics on four cache lines: 210 us of this kernel whatever else
        // it did.  The scalar is summed in a few slots instead and scattered once by k_objective_finish.
        const unsigned int slot = ((unsigned int)bxi + 31u * (unsigned int)byi + 977u * (unsigned int)b) % ESLOTS;
#pragma unroll
        for (int c = 0; c < CS; ++c) {
            const float e = wave_sum_dpp(esum[c]);
            if (lane == 0 && e != 0.0f) atomicAdd(a.esum + 4 * slot + c, e);
        }
    }
}

// Counting call (fpcdr_objective_params.count_only): how many occupied bins show a silhouette triangle in their plane or its apron -- the
// bins a real call gives a record slot to.  One workgroup per occupied bin, the load phase of k_shade and nothing else.
__global__ void __launch_bounds__(ONT) k_count_sil(const int32_t *__restrict__ list, const int32_t *__restrict__ count, int OX, int OY,
                                                   fpcdr_bin_decode dc, ObjArgs a) {
    const int n = *count;
    for (int item = blockIdx.x; item < n; item += gridDim.x) {      // (uniform trip count)
        const int lin = __builtin_amdgcn_readfirstlane(list[item]);
        int b, byi, bxi;
        fpcdr_decode_bin(lin, dc, b, byi, bxi);
        const size_t bin_lin = ((size_t)b * OY + byi) * OX + bxi;
        const unsigned int wmask = (unsigned int)__builtin_amdgcn_readfirstlane((int)a.occ[bin_lin]);
        const int tid = threadIdx.x;
        const uint4 v = reinterpret_cast<const uint4 *>(a.idp + bin_lin * (OB * OB))[tid];
        unsigned int seen = (v.x | v.y | v.z | v.w) >> 24;
        if (tid < 4 * OB) {
            const int side = tid >> 5, i = tid & 31;      // 0 left, 1 right, 2 below, 3 above (as in shade_body)
            const int dx = side == 0 ? -1 : (side == 1 ? 1 : 0), dy = side == 2 ? -1 : (side == 3 ? 1 : 0);
            if ((wmask >> ((dy + 1) * 4 + dx + 1)) & 1u) {
                const size_t nb = (size_t)((long long)bin_lin + dy * OX + dx);
                const int sx = side == 0 ? OB - 1 : (side == 1 ? 0 : i), sy = side == 2 ? OB - 1 : (side == 3 ? 0 : i);
                seen |= a.idp[nb * (OB * OB) + sy * OB + sx] >> 24;
            }
        }
        if (__syncthreads_or(seen != 0u ? 1 : 0) && tid == 0) atomicAdd(a.pool_count, 1);
    }
}

// The last kernel of the call, one wave:
//  * the empty pixels' share of the texture gradient: sum of the slots, times the four bilinear weights of uv = (0,0);
//  * the objective's value from the loss slots (the arithmetic of k_objective_value, loss.hip);
//  * the launch-hint counters of the occupancy header copied to where the caller wants them -- host-mapped memory as a rule -- with
//    the caller's sequence number behind them.
struct FinishArgs {
    int32_t *counts_out; int32_t counts_seq;
    const double *bg_sumsq; double bg_coeff, n_total; float *value_out;
    int esum;
    float *skip_out;      // [1] device: 1 when the call ran out of record slots (fpcdr_adam_params.skip_flag), else 0
    int slots_valid;      // counts_out[7]: header [1] counts record slots (compact records or a counting call)
};
template <int CS>
__global__ void __launch_bounds__(64) k_objective_finish(ObjArgs a, FinishArgs f) {
    const int c = threadIdx.x;
    // compact records: 1 = the pool was too small -- some bins were shaded without their antialias pairs, the call's results are invalid
    const bool overflow = a.def_count[4] != 0;
    if (f.esum && c < CS) {
        float e = 0.0f;
        for (int sl = 0; sl < ESLOTS; ++sl) e += a.esum[4 * sl + c];
        if (e != 0.0f) {
            const Taps tp0 = make_taps(0.0f, 0.0f, a.Ht, a.Wt, CS, a.boundary);
            if (tp0.valid & 1u) atomicAdd(a.grad_tex + tp0.i00 + c, e * ((1.0f - tp0.fx) * (1.0f - tp0.fy)));
            if (tp0.valid & 2u) atomicAdd(a.grad_tex + tp0.i10 + c, e * (tp0.fx * (1.0f - tp0.fy)));
            if (tp0.valid & 4u) atomicAdd(a.grad_tex + tp0.i01 + c, e * ((1.0f - tp0.fx) * tp0.fy));
            if (tp0.valid & 8u) atomicAdd(a.grad_tex + tp0.i11 + c, e * (tp0.fx * tp0.fy));
        }
    }
    if (f.value_out) {
        double acc = 0.0;
        for (int i = threadIdx.x; i < FPCDR_LOSS_SLOTS; i += 64) acc += a.loss_sum[i];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o, 64);
        if (threadIdx.x == 0) {
            if (f.bg_sumsq) acc = acc + f.bg_coeff * f.bg_sumsq[0];
            // (an invalid call says so in its value: nobody mistakes it for a loss)
            f.value_out[0] = overflow ? __int_as_float(0x7fc00000) : (float)(acc / f.n_total);
        }
    }
    if (f.skip_out && threadIdx.x == 0) f.skip_out[0] = overflow ? 1.0f : 0.0f;
    if (f.counts_out && threadIdx.x == 0) {
        // a sequence lock (include/fpcdr.h): [6] = seq, counters, [4] = seq -- a reader that finds [4] == [6] around its reads of the
        // counters has the counters of one call.  [5] is CUMULATIVE (a later call on the shape must not wipe an overflow the host has
        // not polled yet): host-mapped memory this kernel reads back, one load over the bus in a call of milliseconds
        volatile int32_t *o = f.counts_out;
        o[6] = f.counts_seq;
        __threadfence_system();
        for (int i = 0; i < 4; ++i) o[i] = a.def_count[i];
        if (overflow) o[5] = o[5] + 1;
        o[7] = f.slots_valid;
        __threadfence_system();
        o[4] = f.counts_seq;
    }
}

template <int CS>
__global__ void __launch_bounds__(FNT) k_fix_mip_list(const int32_t *__restrict__ list, const int32_t *__restrict__ count, int cap, int OX, int OY,
                                                      fpcdr_bin_decode dc, ObjArgs a, MipO ma) {
    const int item = fpcdr_list_item(*count, cap);
    if (item < 0) return;
    const int lin = __builtin_amdgcn_readfirstlane(list[item]);
    int b, byi, bxi;
    fpcdr_decode_bin(lin, dc, b, byi, bxi);
    fix_body<CS, 1, true>(b, bxi, byi, OX, OY, a, &ma);
}
template <int CS>
__global__ void __launch_bounds__(FNT) k_fix_mip_queue(const int32_t *__restrict__ list, const int32_t *__restrict__ count, int first, int OX, int OY,
                                                       fpcdr_bin_decode dc, ObjArgs a, MipO ma) {
    const int n = *count;
    for (int item = first + blockIdx.x; item < n; item += gridDim.x) {
        const int lin = __builtin_amdgcn_readfirstlane(list[item]);
        int b, byi, bxi;
        fpcdr_decode_bin(lin, dc, b, byi, bxi);
        fix_body<CS, 1, true>(b, bxi, byi, OX, OY, a, &ma);
        __syncthreads();
    }
}

template <int CS, int PASS>
__global__ void __launch_bounds__(FNT) k_fix_list(const int32_t *__restrict__ list, const int32_t *__restrict__ count, int cap, int OX, int OY,
                                                  fpcdr_bin_decode dc, ObjArgs a) {
    const int item = fpcdr_list_item(*count, cap);
    if (item < 0) return;
    const int lin = __builtin_amdgcn_readfirstlane(list[item]);
    int b, byi, bxi;
    fpcdr_decode_bin(lin, dc, b, byi, bxi);
    fix_body<CS, PASS>(b, bxi, byi, OX, OY, a);
}
template <int CS, int PASS>
__global__ void __launch_bounds__(FNT) k_fix_queue(const int32_t *__restrict__ list, const int32_t *__restrict__ count, int first, int OX, int OY,
                                                   fpcdr_bin_decode dc, ObjArgs a) {
    const int n = *count;
    for (int item = first + blockIdx.x; item < n; item += gridDim.x) {
        const int lin = __builtin_amdgcn_readfirstlane(list[item]);
        int b, byi, bxi;
        fpcdr_decode_bin(lin, dc, b, byi, bxi);
        fix_body<CS, PASS>(b, bxi, byi, OX, OY, a);
        __syncthreads();
    }
}

// per-image silhouette classification (same arithmetic as k_sil in antialias.hip).  The kernel is a chain of gathers with two dozen
// instructions behind them -- latency, not issue, is its cost -- so a thread classifies its triangle in SIL_NI images: the six
// indices (own vertices, vertices across the three edges) are loaded once, and the 6 x SIL_NI position gathers are all in flight
// before the first is used (the vertex across an edge used to be fetched only after the edge's line had been computed).
__global__ void __launch_bounds__(256) k_sil2(const float4 *__restrict__ pos, const int32_t *__restrict__ tri,
                                              const int32_t *__restrict__ adj, int B, int V, int T, float hw, float hh,
                                              uint8_t *__restrict__ sil, uint4 *__restrict__ zero_dst, unsigned long long zero_n16) {
    // grid (triangle chunks, groups of SIL_NI images): a flat thread index would cost every thread a 64-bit division
    const int b0 = blockIdx.y * SIL_NI, t = blockIdx.x * blockDim.x + threadIdx.x;
    // fpcdr_render_loss_fwd: the antialias flag planes of the call (149 MB at cfg3) are zeroed HERE, by stores that cost this
    // latency-bound kernel next to nothing, instead of by a 32-50 us fill of the caller's in front of the call
    if (zero_dst) {
        const unsigned long long stride = (unsigned long long)gridDim.x * gridDim.y * blockDim.x;
        for (unsigned long long i = ((unsigned long long)blockIdx.y * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x; i < zero_n16; i += stride)
            zero_dst[i] = make_uint4(0u, 0u, 0u, 0u);
    }
    if (t >= T) return;
    sil_classify(pos, tri, adj, B, V, T, hw, hh, sil, b0, t);      // (sil_bits.h: same arithmetic as k_sil in antialias.hip)
}

}  // namespace

// (also a piece of the two-call form's fpcdr_render_loss_fwd; not part of the C ABI)
int fpcdr_launch_sil(const float *pos, const int32_t *tri, const int32_t *adj, int B, int V, int T, int H, int W, uint8_t *sil,
                     void *zero_dst, size_t zero_bytes, hipStream_t st) {
    if (zero_dst && (((size_t)zero_dst | zero_bytes) & 15)) {      // (not 16-byte shaped: a plain memset)
        FPCDR_REQUIRE(hipMemsetAsync(zero_dst, 0, zero_bytes, st) == hipSuccess, "memset of the flag planes failed");
        zero_dst = nullptr;
    }
    hipLaunchKernelGGL(k_sil2, dim3(fpcdr_cdiv(T, 256), fpcdr_cdiv(B, SIL_NI)), dim3(256), 0, st, (const float4 *)pos, tri, adj, B, V, T,
                       0.5f * (float)W, 0.5f * (float)H, sil, (uint4 *)zero_dst, (unsigned long long)(zero_bytes / 16));
    FPCDR_CHECK_LAUNCH();
    return FPCDR_OK;
}

#ifdef FPCDR_OPROF
extern "C" int fpcdr_debug_oprof(unsigned long long *out16, int reset) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_oprof), sizeof(unsigned long long) * 16) != hipSuccess) return 1;
    if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_oprof), z, sizeof(z)) != hipSuccess) return 1; }
    return 0;
}
#endif

extern "C" int fpcdr_silhouette_bits(const float *pos, const int32_t *tri, const int32_t *adj, int32_t B, int32_t V, int32_t T, int32_t H,
                                     int32_t W, uint8_t *sil, void *stream) {
    FPCDR_REQUIRE(pos && tri && adj && sil, "null pointer");
    FPCDR_REQUIRE(B > 0 && V > 0 && T > 0 && H > 0 && W > 0 && B <= 65535, "bad sizes");
    return fpcdr_launch_sil(pos, tri, adj, B, V, T, H, W, sil, nullptr, 0, (hipStream_t)stream);
}

extern "C" int fpcdr_objective_fwd(const fpcdr_objective_params *p, void *stream) {
    FPCDR_REQUIRE(p != nullptr, "null params");
    FPCDR_REQUIRE(p->pos && p->tri && p->adj && p->scratch && p->uv && p->uv_tri && p->tex && p->ref, "null pointer");
    FPCDR_REQUIRE(p->sil && p->idp && p->occ && p->cmask && p->empty_color && p->loss_sum, "null pointer");
    FPCDR_REQUIRE(p->count_only || (p->rec && p->color && p->grad_aa), "null record buffers");
    FPCDR_REQUIRE(p->rec_slots >= 0 && (p->rec_slots == 0 || p->slot_map), "compact records (rec_slots > 0) need slot_map");
    FPCDR_REQUIRE((int64_t)p->rec_slots * (OB * OB) * 4 < ((int64_t)1 << 31), "rec_slots too large");
    FPCDR_REQUIRE(p->B > 0 && p->V > 0 && p->T > 0 && p->H > 0 && p->W > 0 && p->Vt > 0 && p->Ht > 0 && p->Wt > 0 && p->C > 0,
                  "sizes must be positive");
    FPCDR_REQUIRE(p->C == 1 || p->C == 3 || p->C == 4, "fused objective supports C = 1, 3, 4");
    FPCDR_REQUIRE(p->H <= 32767 && p->W <= 32767 && p->B <= 65535, "resolution / batch too large");
    FPCDR_REQUIRE(p->T < (1 << 24), "more than 2^24 triangles");
    FPCDR_REQUIRE(p->boundary_mode == FPCDR_BOUNDARY_WRAP || p->boundary_mode == FPCDR_BOUNDARY_CLAMP || p->boundary_mode == FPCDR_BOUNDARY_ZERO,
                  "bad boundary mode");
    FPCDR_REQUIRE((long long)p->Ht * p->Wt * p->C < (1ll << 30), "texture too large (the fused paths take < 2^30 texel values)");
    FPCDR_REQUIRE(((size_t)p->occ & 3) == 0 && ((size_t)p->cmask & 7) == 0 && ((size_t)p->idp & 15) == 0 && ((size_t)p->rec & 15) == 0,
                  "occ must be 4-byte, cmask 8-byte, idp and rec 16-byte aligned");
    if (p->mip) {
        FPCDR_REQUIRE(p->n_levels >= 0 && p->n_levels <= FPCDR_MAX_MIP, "bad n_levels");
        for (int lvl = 1; lvl <= p->n_levels; ++lvl) {
            FPCDR_REQUIRE(p->tex_mip[lvl - 1] != nullptr && (!p->grad_tex || p->grad_tex_mip[lvl - 1] != nullptr), "missing mip level");
            FPCDR_REQUIRE(!((p->Ht >> (lvl - 1)) & 1) && !((p->Wt >> (lvl - 1)) & 1), "mip levels need even sizes");
        }
    }
    hipStream_t st = (hipStream_t)stream;
    int rc = FPCDR_OK;      // (the silhouette bits: the caller's (sil_ready), or the set-up kernel's)
    const int32_t *occ_list = nullptr, *n_occ = nullptr;
    const int OX = FPCDR_OCC_DIM(p->W), OY = FPCDR_OCC_DIM(p->H);
    const long long nbins = (long long)p->B * OY * OX;
    const fpcdr_queue_layout q = fpcdr_queue_layout_of(p->B, p->H, p->W, true);
    // (this form's cmask: 128 B per bin of candidate masks, 128 B per bin of hit masks, then the slots)
    float *esum_slots = (float *)((char *)p->cmask + q.cm_esum);
    static_assert(ESLOTS * 4 * sizeof(float) <= 512, "the slots' room in fpcdr_queue_layout_of");
    // what the first kernel zero-fills beside its own maps: the slots, and with zero_outputs the caller's accumulators
    FpcdrZeroList zl = {};
    if (p->grad_tex) zl.add(esum_slots, ESLOTS * 4);
    if (p->zero_outputs) {
        zl.add(p->loss_sum, (long long)FPCDR_LOSS_SLOTS * 2);
        zl.add(p->grad_pos, (long long)p->B * p->V * 4);
        zl.add(p->grad_tex, (long long)p->Ht * p->Wt * p->C);
        if (p->mip && p->grad_tex)
            for (int lvl = 1; lvl <= p->n_levels; ++lvl) zl.add(p->grad_tex_mip[lvl - 1], (long long)(p->Ht >> lvl) * (p->Wt >> lvl) * p->C);
    }
    if (p->zero_extra) {
        FPCDR_REQUIRE(p->zero_extra_bytes >= 0 && (p->zero_extra_bytes & 3) == 0 && ((size_t)p->zero_extra & 3) == 0, "zero_extra: 4-byte units");
        zl.add(p->zero_extra, p->zero_extra_bytes / 4);
    }
    FPCDR_REQUIRE(!zl.overflow, "too many buffers to zero-fill (FpcdrZeroList::MAXR)");
    rc = fpcdr_launch_raster_ids(p, st, &occ_list, &n_occ, zl, !p->sil_ready);
    if (rc) return rc;
    ObjArgs a = {(const float4 *)p->pos, p->tri, (const float2 *)p->uv, p->uv_tri, (const float2 *)p->tri_uv, p->tex, p->ref, p->sil,
                 p->idp, p->occ, (uint8_t *)p->occ + q.occ_binflag, p->cmask, esum_slots, (int32_t *)((char *)p->occ + q.occ_bwd_list),
                 (int32_t *)((char *)p->occ + q.occ_hdr), (uint32_t *)((char *)p->cmask + q.cm_edges), (float4 *)p->rec, p->color, p->grad_aa, p->empty_color,
                 p->loss_sum, p->grad_pos, p->grad_tex, p->B, p->V, p->T, p->H, p->W, p->Ht, p->Wt, p->boundary_mode,
                 p->bg, p->color_scale, p->grad_scale, (unsigned long long *)p->flags,
                 p->slot_map, (int32_t *)((char *)p->occ + q.occ_hdr) + 1, (int32_t *)((char *)p->occ + q.occ_hdr) + 4, p->rec_slots};
    const fpcdr_bin_decode dc = fpcdr_make_bin_decode(OX, OY);
    const int cap = (p->cap_occ > 0 && p->cap_occ < nbins) ? p->cap_occ : (int)nbins;
    const dim3 grid(fpcdr_list_grid(cap));
    const bool sweep = cap < nbins;
    const ShadeK shade_k = {occ_list, n_occ, cap, OX, OY, dc, a};
#define SHADE(CS, BM)                                                                                                             \
    do {                                                                                                                          \
        hipLaunchKernelGGL((k_shade_list<CS, BM>), grid, dim3(ONT), 0, st, shade_k);                                               \
        if (sweep) hipLaunchKernelGGL(k_shade_queue<CS>, dim3(FPCDR_SWEEP_WGS), dim3(ONT), 0, st, shade_k);                        \
    } while (0)
    // k_fix: over the bins k_shade found a deferred pixel in (count at occ header [0]; p->cap_def: launch hint)
    const int cap_d = (p->cap_def > 0 && p->cap_def < nbins) ? p->cap_def : (int)nbins;
    const dim3 grid_d(fpcdr_list_grid(cap_d));
    const bool sweep_d = cap_d < nbins;
#define FIX(CS, PASS)                                                                                                             \
    do {                                                                                                                          \
        hipLaunchKernelGGL((k_fix_list<CS, PASS>), grid_d, dim3(FNT), 0, st, a.def_list, a.def_count, cap_d, OX, OY, dc, a);       \
        if (sweep_d) hipLaunchKernelGGL((k_fix_queue<CS, PASS>), dim3(FPCDR_SWEEP_WGS), dim3(FNT), 0, st, a.def_list, a.def_count, cap_d, OX, OY, dc, a); \
    } while (0)
    if (p->count_only) {
        // the lists, the id planes and the number of bins that would take a record slot (header [1]), reported through counts_out: what a
        // caller without launch hints needs to size the compact record arrays of the real call (once per batch shape)
        FPCDR_REQUIRE(p->counts_out != nullptr, "count_only reports through counts_out");
        hipLaunchKernelGGL(k_count_sil, dim3(2048), dim3(ONT), 0, st, occ_list, n_occ, OX, OY, dc, a);
        const FinishArgs fin0 = {p->counts_out, p->counts_seq, nullptr, 0.0, 1.0, nullptr, 0, nullptr, 1};
        hipLaunchKernelGGL(k_objective_finish<1>, dim3(1), dim3(64), 0, st, a, fin0);
        FPCDR_CHECK_LAUNCH();
        return FPCDR_OK;
    }
    const bool grads = p->grad_pos || p->grad_tex;
    const FinishArgs fin = {p->counts_out, p->counts_seq, p->bg_sumsq, p->bg_coeff, p->n_total, p->value_out, p->grad_tex ? 1 : 0,
                            p->skip_out, p->rec_slots > 0 ? 1 : 0};
    FPCDR_REQUIRE(!p->value_out || p->n_total > 0.0, "value_out needs n_total > 0");
#define FINISH(CS)                                                                                                                \
    do {                                                                                                                          \
        if (fin.esum || fin.value_out || fin.counts_out || fin.skip_out) hipLaunchKernelGGL(k_objective_finish<CS>, dim3(1), dim3(64), 0, st, a, fin); \
    } while (0)
    if (p->mip) {      // the reference's enable_mip branch (fit.py:153-155)
        MipO ma;
        ma.lv.tex[0] = p->tex;
        ma.lv.grad[0] = p->grad_tex;
        for (int lvl = 1; lvl <= FPCDR_MAX_MIP; ++lvl) {
            ma.lv.tex[lvl] = lvl <= p->n_levels ? p->tex_mip[lvl - 1] : nullptr;
            ma.lv.grad[lvl] = (lvl <= p->n_levels && p->grad_tex) ? p->grad_tex_mip[lvl - 1] : nullptr;
        }
        ma.n_levels = p->n_levels;
        const ShadeMipK shade_mip_k = {occ_list, n_occ, cap, OX, OY, dc, a, ma};
#define SHADE_MIP(CS)                                                                                                             \
    do {                                                                                                                          \
        hipLaunchKernelGGL(k_shade_mip_list<CS>, grid, dim3(ONT), 0, st, shade_mip_k);                                             \
        if (sweep) hipLaunchKernelGGL(k_shade_mip_queue<CS>, dim3(FPCDR_SWEEP_WGS), dim3(ONT), 0, st, shade_k, ma);               \
    } while (0)
#define FIX_MIP(CS)                                                                                                               \
    do {                                                                                                                          \
        hipLaunchKernelGGL(k_fix_mip_list<CS>, grid_d, dim3(FNT), 0, st, a.def_list, a.def_count, cap_d, OX, OY, dc, a, ma);       \
        if (sweep_d) hipLaunchKernelGGL(k_fix_mip_queue<CS>, dim3(FPCDR_SWEEP_WGS), dim3(FNT), 0, st, a.def_list, a.def_count, cap_d, OX, OY, dc, a, ma); \
    } while (0)
        if (p->C == 1) { SHADE_MIP(1); FIX(1, 0); if (grads) FIX_MIP(1); FINISH(1); }
        else if (p->C == 3) { SHADE_MIP(3); FIX(3, 0); if (grads) FIX_MIP(3); FINISH(3); }
        else { SHADE_MIP(4); FIX(4, 0); if (grads) FIX_MIP(4); FINISH(4); }
#undef SHADE_MIP
#undef FIX_MIP
    } else
    if (p->C == 1) {
        if (p->boundary_mode == FPCDR_BOUNDARY_WRAP && p->tri_uv) SHADE(1, FPCDR_BOUNDARY_WRAP);      // the reference's case, as compile-time constants
        else SHADE(1, -1);
        FIX(1, 0);
        if (grads) FIX(1, 1);
        FINISH(1);
    } else if (p->C == 3) {
        SHADE(3, -1);
        FIX(3, 0);
        if (grads) FIX(3, 1);
        FINISH(3);
    } else {
        SHADE(4, -1);
        FIX(4, 0);
        if (grads) FIX(4, 1);
        FINISH(4);
    }
#undef SHADE
#undef FIX
#undef FINISH
    FPCDR_CHECK_LAUNCH();
    return FPCDR_OK;
}

