// antialias forward / backward + topology for gfx950 (MI355X).
//
// Performs the work of `dr.antialias(colour, rast_out, pos_clip, pos_idx)` at reference
// src/torch/fit.py:160 (nvdiffrast op, absent from the reference tree): at horizontally / vertically
// adjacent pixel pairs showing different triangles, if a silhouette edge of the nearer triangle crosses
// the segment between the two pixel centres, the two colours are blended by the crossing position.
// This is the only source of visibility (silhouette) gradients in the fit loop.
//
// MI355X-first structure -- no global work list, no colour atomics:
//   k_topo_*   once per index buffer: open-addressed edge table (atomicCAS on 64-bit keys; commutative
//              count / sum updates so the result does not depend on insertion order) resolved into
//              adj[T][3] = opposite vertex of the one neighbouring triangle (or -1 boundary, -2 non-manifold).
//   k_sil      per (image, triangle): 3 bits "edge e is a silhouette edge in this image".  Interior
//              pixel pairs (the vast majority of id discontinuities on a 30k-triangle mesh) are then
//              rejected by one byte load.
//   k_aa_fwd   one pixel per lane, GATHER form: every pixel evaluates its four pairs and adds only
//              the blends it receives, so the colour is written once with plain stores, the result is
//              deterministic, and HBM traffic is the algorithmic 16 + 8C bytes per pixel.  The wave
//              (64 consecutive pixels of a row) publishes two 64-bit ballot words: pair (p,p+x) /
//              (p,p+y) blended.
//   backward   grad_colour = dy is one device-to-device copy (the HBM-bound bulk, 8C bytes per pixel);
//              k_aa_bwd_fix then scans the two bit planes (2 bits per pixel) and only where a pair was
//              blended re-evaluates it, rewrites the affected pixels' grad_colour in gather form and
//              scatters d(alpha)/d(pos) sparsely.
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------------
// topology
// ------------------------------------------------------------------------------------------------
constexpr unsigned long long EMPTY_KEY = ~0ull;

__host__ __device__ inline unsigned int topo_capacity(int T) {
    unsigned int need = (unsigned int)T * 6u, cap = 64;
    while (cap < need) cap <<= 1;
    return cap;
}

__device__ __forceinline__ unsigned int hash_key(unsigned long long k) {
    k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
    return (unsigned int)k;
}

__global__ void k_topo_init(unsigned long long *keys, unsigned int *cnt, unsigned int *sum, unsigned int cap) {
    unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cap) { keys[i] = EMPTY_KEY; cnt[i] = 0; sum[i] = 0; }
}

__device__ __forceinline__ unsigned long long edge_key(int a, int b) {
    unsigned int lo = (unsigned int)min(a, b), hi = (unsigned int)max(a, b);
    return ((unsigned long long)lo << 32) | hi;
}

__global__ void k_topo_insert(const int32_t *__restrict__ tri, int T, unsigned long long *keys, unsigned int *cnt,
                              unsigned int *sum, unsigned int cap) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 3 * T) return;
    const int t = i / 3, e = i - 3 * t;
    const int a = tri[3 * t + (e + 1) % 3], b = tri[3 * t + (e + 2) % 3], o = tri[3 * t + e];
    const unsigned long long key = edge_key(a, b);
    unsigned int slot = hash_key(key) & (cap - 1);
    for (unsigned int probe = 0; probe < cap; ++probe) {
        const unsigned long long prev = atomicCAS(&keys[slot], EMPTY_KEY, key);
        if (prev == EMPTY_KEY || prev == key) {
            atomicAdd(&cnt[slot], 1u);
            atomicAdd(&sum[slot], (unsigned int)o);
            return;
        }
        slot = (slot + 1) & (cap - 1);
    }
}

__global__ void k_topo_resolve(const int32_t *__restrict__ tri, int T, const unsigned long long *keys, const unsigned int *cnt,
                               const unsigned int *sum, unsigned int cap, int32_t *__restrict__ adj) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 3 * T) return;
    const int t = i / 3, e = i - 3 * t;
    const int a = tri[3 * t + (e + 1) % 3], b = tri[3 * t + (e + 2) % 3], o = tri[3 * t + e];
    const unsigned long long key = edge_key(a, b);
    unsigned int slot = hash_key(key) & (cap - 1);
    int res = -2;
    for (unsigned int probe = 0; probe < cap; ++probe) {
        const unsigned long long k = keys[slot];
        if (k == key) {
            const unsigned int c = cnt[slot];
            res = (c == 1) ? -1 : (c == 2 ? (int)(sum[slot] - (unsigned int)o) : -2);
            break;
        }
        if (k == EMPTY_KEY) break;
        slot = (slot + 1) & (cap - 1);
    }
    adj[i] = res;
}

// ------------------------------------------------------------------------------------------------
// per-image silhouette classification (uncentred pixel-scaled homogeneous coordinates)
// ------------------------------------------------------------------------------------------------
#include "sil_bits.h"
__global__ void __launch_bounds__(256) k_sil(const float4 *__restrict__ pos, const int32_t *__restrict__ tri,
                                             const int32_t *__restrict__ adj, int B, int V, int T, float hw, float hh,
                                             uint8_t *__restrict__ sil) {
    // grid (triangle chunks, groups of SIL_NI images): a flat thread index would cost every thread a 64-bit division
    const int b0 = blockIdx.y * SIL_NI, t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    sil_classify(pos, tri, adj, B, V, T, hw, hh, sil, b0, t);
}

#include "aa_pairs.h"

constexpr int AROWS = 8;   // rows per wave of k_aa_fwd (4 waves: 32 rows = one hint bin)

// ------------------------------------------------------------------------------------------------
// does pair (p0 first pixel, p1 second) have a triangle to analyse that owns a silhouette edge?  (one byte load)
__device__ __forceinline__ bool pair_candidate(const uint8_t *__restrict__ sil, int T, float2 p0, float2 p1) {
    if ((int)p0.y == (int)p1.y) return false;
    const PairSel ps = pair_select((int)p0.y, p0.x, (int)p1.y, p1.x, T);
    return ps.tau >= 0 && sil[ps.tau] != 0;
}

template <int CS>
__global__ void __launch_bounds__(256) k_aa_fwd(const float *__restrict__ color, const float4 *__restrict__ rast,
                                                const float4 *__restrict__ pos, const int32_t *__restrict__ tri,
                                                const uint8_t *__restrict__ sil, int B, int H, int W, int C_dyn, int V, int T,
                                                unsigned long long *__restrict__ flags, float *__restrict__ out,
                                                const uint8_t *__restrict__ hint, const float *__restrict__ empty_color, int filled,
                                                int flags_zeroed) {
    // 64 x (4 AROWS) pixels per workgroup: a wave owns AROWS consecutive rows of a 64-pixel column strip (one 32-row hint bin)
    // and has the (z/w, id) loads of all of them and of the rows above and below in flight at once (one pixel per thread:
    // 2.3 M tiny workgroups, see k_interp_fwd in interpolate.hip).
    // Phase 1, every pixel: out = colour unless one of its four pairs is a CANDIDATE -- different ids and the triangle to
    // analyse owns a silhouette edge (a byte per (image, triangle)): ~3 % of the covered pixels of a closed mesh, while ~40 %
    // see an id discontinuity.  Phase 2, rows with a candidate: the pair analysis (divergent, a dozen dependent loads).
    const int C = CS > 0 ? CS : C_dyn;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = blockIdx.x * 64 + lane, ybase = (blockIdx.y * 4 + wave) * AROWS, b = blockIdx.z;
    if (ybase >= H) return;
    const int Wq = FPCDR_AA_ROW_WORDS(W);
    const size_t img = (size_t)b * H * W;
    const uint8_t *silb = sil + (size_t)b * T;
    // region hint (plane 1): the pixel's bin and its eight neighbours are empty, so neither the pixel nor any of its four
    // neighbours is covered -- no pair to blend: out = colour, without reading rast (nor colour, when its value there is known)
    const bool skip = x < W && hint && !fpcdr_hint_on(hint, 1, B, H, W, b, ybase, x);
    bool live = x < W && !skip;
    if (live && hint && !fpcdr_hint_on(hint, 0, B, H, W, b, ybase, x)) {
        // an empty bin beside an occupied one: a pair needs a covered pixel, and pairs are horizontal or vertical, so only the pixels on
        // a side that FACES an occupied bin can be blended -- a column of the bin, or the wave's rows if they hold the facing row (a
        // third of the bins whose rast the kernel reads are of this kind)
        const int cx = x & 31;
        bool face = false;
        if (cx == 0 && x > 0) face |= fpcdr_hint_on(hint, 0, B, H, W, b, ybase, x - 1);
        if (cx == 31 && x + 1 < W) face |= fpcdr_hint_on(hint, 0, B, H, W, b, ybase, x + 1);
        if ((ybase & 31) == 0 && ybase > 0) face |= fpcdr_hint_on(hint, 0, B, H, W, b, ybase - 1, x);
        if (((ybase + AROWS) & 31) == 0 && ybase + AROWS < H) face |= fpcdr_hint_on(hint, 0, B, H, W, b, ybase + AROWS, x);
        live = face;
    }
    if (filled && flags_zeroed && !__builtin_amdgcn_readfirstlane(__ballot(live) != 0ull)) return;   // nothing left to do for this wave
    float2 mer[AROWS + 2];
#pragma unroll
    for (int r = -1; r <= AROWS; ++r) {
        const int y = min(max(ybase + r, 0), H - 1);      // (clamped: a pixel is its own neighbour beyond the image)
        mer[r + 1] = live ? load_zid(rast, img + (size_t)y * W + x) : make_float2(0.f, 0.f);
    }
    unsigned int cand = 0;
#pragma unroll
    for (int r = 0; r < AROWS; ++r) {
        const int y = ybase + r;
        if (y >= H) break;
        const size_t off = img + (size_t)y * W + x;
        if (!flags_zeroed && lane == 0) {
            const size_t wi = ((size_t)b * H + y) * Wq + blockIdx.x;
            flags[wi] = 0ull;
            flags[(size_t)B * H * Wq + wi] = 0ull;
        }
        if (skip) {
            if (!filled) {      // (filled: k_aa_fill_bin1 has written these pixels with 16-byte stores)
                if (empty_color) { for (int c = 0; c < C; ++c) out[off * C + c] = empty_color[c]; }
                else { for (int c = 0; c < C; ++c) out[off * C + c] = color[off * C + c]; }
            }
        } else if (x < W && !live) {      // (an empty bin's pixel that no pair can reach)
            if (filled < 2) { for (int c = 0; c < C; ++c) out[off * C + c] = color[off * C + c]; }
        } else if (x < W) {
            const float2 me = mer[r + 1];
            const int id = (int)me.y;
            const float2 nR = x + 1 < W ? load_zid(rast, off + 1) : me;
            const float2 nL = x > 0 ? load_zid(rast, off - 1) : me;
            const float2 nU = mer[r + 2], nD = mer[r];
            bool c4 = false;
            if (((int)nR.y != id) | ((int)nL.y != id) | ((int)nU.y != id) | ((int)nD.y != id))
                c4 = (int)pair_candidate(silb, T, me, nR) | (int)pair_candidate(silb, T, me, nU) | (int)pair_candidate(silb, T, nL, me) |
                     (int)pair_candidate(silb, T, nD, me);
            if (c4) cand |= 1u << r;
            else if (filled < 2) { for (int c = 0; c < C; ++c) out[off * C + c] = color[off * C + c]; }      // (2: the fill kernel copied it)
        }
    }
    unsigned int rows = 0;
#pragma unroll
    for (int r = 0; r < AROWS; ++r)
        if (__ballot((cand >> r) & 1u)) rows |= 1u << r;
#pragma unroll 1
    for (int r = 0; r < AROWS; ++r) {
        if (!((rows >> r) & 1u)) continue;
        const int y = ybase + r;
        bool fx_flag = false, fy_flag = false;
        if ((cand >> r) & 1u) {
            const size_t off = img + (size_t)y * W + x;
            const bool hasR = x + 1 < W, hasL = x > 0, hasU = y + 1 < H, hasD = y > 0;
            const float2 me = load_zid(rast, off);
            const float2 nR = hasR ? load_zid(rast, off + 1) : me;
            const float2 nL = hasL ? load_zid(rast, off - 1) : me;
            const float2 nU = hasU ? load_zid(rast, off + W) : me;
            const float2 nD = hasD ? load_zid(rast, off - W) : me;
            float acc[CS > 0 ? CS : 1];
            const float *cme = color + off * C;
            if (CS > 0) {
#pragma unroll
                for (int c = 0; c < CS; ++c) acc[c] = cme[c];
            }
            AAGeom g = {pos + (size_t)b * V, tri, silb, T, W, H, 0.5f * (float)W, 0.5f * (float)H};
            // gather: add what THIS pixel receives from each of its four pairs
            auto visit = [&](int x0, int y0, int d, float2 p0, float2 p1, bool own, bool &flag) {
                if ((int)p0.y == (int)p1.y) return;
                bool hit = for_active_edges(g, x0, y0, d, (int)p0.y, p0.x, (int)p1.y, p1.x,
                    [&](float t, int Px, int Py, int Qx, int Qy, int, int, const EdgeEval &, float) {
                        const bool far = t >= 0.5f;
                        const int rx = far ? Qx : Px, ry = far ? Qy : Py;
                        if (rx != x || ry != y) return;
                        const int ox = far ? Px : Qx, oy = far ? Py : Qy;
                        const float amt = far ? t - 0.5f : 0.5f - t;
                        const float *co = color + (img + (size_t)oy * W + ox) * C;
                        if (CS > 0) {
#pragma unroll
                            for (int c = 0; c < CS; ++c) acc[c] += amt * (co[c] - cme[c]);
                        } else {
                            for (int c = 0; c < C; ++c) out[off * C + c] += amt * (co[c] - cme[c]);
                        }
                    });
                if (own && hit) flag = true;
            };
            if (CS == 0) for (int c = 0; c < C; ++c) out[off * C + c] = cme[c];
            bool dummy = false;
            if (hasR) visit(x, y, 0, me, nR, true, fx_flag);
            if (hasU) visit(x, y, 1, me, nU, true, fy_flag);
            if (hasL) visit(x - 1, y, 0, nL, me, false, dummy);
            if (hasD) visit(x, y - 1, 1, nD, me, false, dummy);
            if (CS > 0) {
#pragma unroll
                for (int c = 0; c < CS; ++c) out[off * C + c] = acc[c];
            }
        }
        const unsigned long long bx = __ballot(fx_flag), by = __ballot(fy_flag);
        if (lane == 0 && (bx | by)) {
            const size_t wi = ((size_t)b * H + y) * Wq + blockIdx.x;
            flags[wi] = bx;
            flags[(size_t)B * H * Wq + wi] = by;
        }
    }
}

// C = 1, W % 4 == 0, with a region hint: EVERY pixel's un-antialiased value is written here with one 16-byte store per lane, one
// workgroup per 32 x 32-pixel bin (4-byte-per-lane stores reach ~1 TB/s, 16-byte ones 5): the bins that (with their eight neighbours)
// are empty -- three quarters of a face-rig frame -- get their known constant without a read, the others a copy of colour.  k_aa_fwd
// then only overwrites the pixels it blends (~1 % of a frame) and issues no bulk stores at all.
__global__ void __launch_bounds__(256) k_aa_fill_bin1(const float4 *__restrict__ color4, int B, int H, int W, float4 *__restrict__ out4,
                                                      const uint8_t *__restrict__ hint, const float *__restrict__ empty_color) {
    const int tid = threadIdx.x, b = blockIdx.z;
    const int px = blockIdx.x * 32 + (tid & 7) * 4, py = blockIdx.y * 32 + (tid >> 3);
    if (px >= W || py >= H) return;
    const size_t i4 = (((size_t)b * H + py) * W + px) / 4;
    if (!fpcdr_hint_on(hint, 1, B, H, W, b, py, px) && empty_color) { const float e = empty_color[0]; out4[i4] = make_float4(e, e, e, e); }
    else out4[i4] = color4[i4];
}

// ------------------------------------------------------------------------------------------------
// Backward, bulk part: grad_colour = dy.  Each 256-thread workgroup streams ONE contiguous 8 KiB piece with two
// 16-byte non-temporal loads in flight per lane.  Measured at cfg3 (2.4 GB in, 2.4 GB out, with the fix-up kernel):
// hipMemcpyAsync 1.08 ms; a grid-stride float4 copy 1.08-1.19 ms for 1k-64k workgroups; contiguous pieces of
// 64 / 32 / 16 / 8 KiB per workgroup 0.98 / 0.93 / 0.89 / 0.88 ms -- many small contiguous pieces keep every HBM
// channel busy.  0.88 ms = 5.45 TB/s of algorithmic traffic = 68 % of the 8 TB/s peak.
typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int COPY_U = 2;
__global__ void __launch_bounds__(256) k_copy_f4_chunk(const v4f *__restrict__ src, v4f *__restrict__ dst, size_t n4) {
    const size_t lo = (size_t)blockIdx.x * (COPY_U * 256);
    const size_t i = lo + threadIdx.x;
    if (lo + COPY_U * 256 <= n4) {
        v4f r[COPY_U];
#pragma unroll
        for (int k = 0; k < COPY_U; ++k) r[k] = __builtin_nontemporal_load(src + i + k * 256);
#pragma unroll
        for (int k = 0; k < COPY_U; ++k) __builtin_nontemporal_store(r[k], dst + i + k * 256);
    } else {
        for (size_t j = i; j < n4; j += 256) dst[j] = src[j];
    }
}

// Backward, sparse part.  The bulk (grad_colour = dy) is a device-to-device copy issued before this kernel;
// here each wave scans 64 consecutive flag words (= 4096 pixels, 2 KB of flags) and, for the few words
// that carry a blended pair, turns into one lane per pixel of that 64-pixel span: a flagged pixel re-evaluates
// its pairs, rewrites its OWN grad_colour entry completely (gather form: dy + corrections, plain store, no
// atomics, deterministic) and the owner of each pair scatters d(alpha)/d(pos).
template <int CS>
__global__ void __launch_bounds__(256) k_aa_bwd_fix(const float *__restrict__ color, const float4 *__restrict__ rast,
                                                    const float4 *__restrict__ pos, const int32_t *__restrict__ tri,
                                                    const uint8_t *__restrict__ sil, const float *__restrict__ dy, int B, int H,
                                                    int W, int C_dyn, int V, int T, const unsigned long long *__restrict__ flags,
                                                    float boost, float *__restrict__ grad_color, float *__restrict__ grad_pos) {
    const int C = CS > 0 ? CS : C_dyn;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int Wq = FPCDR_AA_ROW_WORDS(W);
    const size_t plane = (size_t)B * H * Wq;
    const size_t wi_l = ((size_t)blockIdx.x * 4 + wave) * 64 + lane;
    unsigned long long w_fx = 0ull, w_fy = 0ull, w_fxl = 0ull, w_fyd = 0ull;
    int w_wq = 0, w_y = 0, w_b = 0;
    if (wi_l < plane) {
        w_wq = (int)(wi_l % Wq);
        const size_t row = wi_l / Wq;
        w_y = (int)(row % H);
        w_b = (int)(row / H);
        w_fx = flags[wi_l];
        w_fy = flags[plane + wi_l];
        if (w_wq > 0) w_fxl = flags[wi_l - 1];
        if (w_y > 0) w_fyd = flags[plane + wi_l - Wq];
    }
    unsigned long long todo = __ballot((w_fx | w_fy | w_fyd | (w_fxl >> 63)) != 0ull);
    while (todo) {
        const int src = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const unsigned long long fx = __shfl(w_fx, src, 64), fy = __shfl(w_fy, src, 64);
        const unsigned long long fxl = __shfl(w_fxl, src, 64), fyd = __shfl(w_fyd, src, 64);
        const int b = __shfl(w_b, src, 64), y = __shfl(w_y, src, 64);
        const int x = __shfl(w_wq, src, 64) * 64 + lane;
        if (x >= W) continue;
    const bool own_x = (fx >> lane) & 1ull, own_y = (fy >> lane) & 1ull;
    const bool left_x = lane > 0 ? ((fx >> (lane - 1)) & 1ull) : ((fxl >> 63) & 1ull);
    const bool down_y = (fyd >> lane) & 1ull;
    if (!(own_x | own_y | left_x | down_y)) continue;
    const size_t img = (size_t)b * H * W;
    const size_t off = img + (size_t)y * W + x;
    const float *g = dy + off * C;
    float go_[CS > 0 ? CS : 1];
    float *go = CS > 0 ? go_ : grad_color + off * C;   // generic C accumulates in place (already holds dy)
    if (CS > 0) {
#pragma unroll
        for (int c = 0; c < CS; ++c) go_[c] = g[c];
    }
    AAGeom geo = {pos + (size_t)b * V, tri, sil + (size_t)b * T, T, W, H, 0.5f * (float)W, 0.5f * (float)H};
    float *gp = grad_pos + (size_t)b * V * 4;
    const float2 me = load_zid(rast, off);
    auto visit = [&](int x0, int y0, int d, float2 p0, float2 p1, bool own) {
        for_active_edges(geo, x0, y0, d, (int)p0.y, p0.x, (int)p1.y, p1.x,
            [&](float t, int Px, int Py, int Qx, int Qy, int va, int vb, const EdgeEval &ev, float s) {
                const bool far = t >= 0.5f;
                const int rx = far ? Qx : Px, ry = far ? Qy : Py;
                const float amt = far ? t - 0.5f : 0.5f - t;
                const size_t roff = img + (size_t)ry * W + rx;
                const float *gr = dy + roff * C;
                // colour gradient: out[r] = c[r] + amt (c[o] - c[r])
                if (rx == x && ry == y) {
                    for (int c = 0; c < C; ++c) go[c] -= amt * gr[c];
                } else {  // this pixel is the "other" one
                    for (int c = 0; c < C; ++c) go[c] += amt * gr[c];
                }
                if (!own) return;
                // position gradient (once per pair, by the pair's owner)
                const float *cP = color + (img + (size_t)Py * W + Px) * C;
                const float *cQ = color + (img + (size_t)Qy * W + Qx) * C;
                float G = 0.f;
                for (int c = 0; c < C; ++c) G += gr[c] * (cP[c] - cQ[c]);
                G *= boost;
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
    if (own_x) visit(x, y, 0, me, load_zid(rast, off + 1), true);
    if (own_y) visit(x, y, 1, me, load_zid(rast, off + W), true);
    if (left_x) visit(x - 1, y, 0, load_zid(rast, off - 1), me, false);
    if (down_y) visit(x, y - 1, 1, load_zid(rast, off - W), me, false);
    if (CS > 0) {
#pragma unroll
        for (int c = 0; c < CS; ++c) grad_color[off * C + c] = go_[c];
    }
    }
}

}  // namespace

extern "C" size_t fpcdr_topology_scratch_bytes(int32_t T) {
    if (T <= 0) return 0;
    return (size_t)topo_capacity(T) * 16;
}

extern "C" int fpcdr_topology_build(const int32_t *tri, int32_t T, void *scratch, int32_t *adj, void *stream) {
    FPCDR_REQUIRE(tri && scratch && adj, "null pointer");
    FPCDR_REQUIRE(T > 0 && T < (1 << 28), "T out of range");
    const unsigned int cap = topo_capacity(T);
    unsigned long long *keys = (unsigned long long *)scratch;
    unsigned int *cnt = (unsigned int *)(keys + cap);
    unsigned int *sum = cnt + cap;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_topo_init, dim3(fpcdr_cdiv(cap, 256)), dim3(256), 0, st, keys, cnt, sum, cap);
    hipLaunchKernelGGL(k_topo_insert, dim3(fpcdr_cdiv(3ll * T, 256)), dim3(256), 0, st, tri, T, keys, cnt, sum, cap);
    hipLaunchKernelGGL(k_topo_resolve, dim3(fpcdr_cdiv(3ll * T, 256)), dim3(256), 0, st, tri, T, keys, cnt, sum, cap, adj);
    FPCDR_CHECK_LAUNCH();
    return FPCDR_OK;
}

extern "C" size_t fpcdr_antialias_flags_bytes(int32_t B, int32_t H, int32_t W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    return (size_t)2 * B * H * FPCDR_AA_ROW_WORDS(W) * sizeof(uint64_t);
}

extern "C" int fpcdr_antialias_fwd(const fpcdr_antialias_fwd_params *p, void *stream) {
    FPCDR_REQUIRE(p != nullptr, "null params");
    FPCDR_REQUIRE(p->color && p->rast && p->pos && p->tri && p->adj && p->sil && p->flags && p->out, "null pointer");
    FPCDR_REQUIRE(p->B > 0 && p->H > 0 && p->W > 0 && p->C > 0 && p->V > 0 && p->T > 0, "sizes must be positive");
    FPCDR_REQUIRE(p->B <= 65535 && fpcdr_cdiv(p->H, 4) <= 65535, "image batch / height too large for one launch");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_sil, dim3(fpcdr_cdiv(p->T, 256), fpcdr_cdiv(p->B, SIL_NI)), dim3(256), 0, st, (const float4 *)p->pos, p->tri,
                       p->adj, p->B, p->V, p->T, 0.5f * (float)p->W, 0.5f * (float)p->H, p->sil);
    dim3 grid(fpcdr_cdiv(p->W, 64), fpcdr_cdiv(p->H, 4 * AROWS), p->B);
    int filled = 0, flags_zeroed = 0;
    if (p->hint) {      // most waves will find nothing to do and leave without touching their flag words
        FPCDR_REQUIRE(hipMemsetAsync(p->flags, 0, fpcdr_antialias_flags_bytes(p->B, p->H, p->W), st) == hipSuccess, "memset of the flag planes failed");
        flags_zeroed = 1;
    }
    if (p->hint && p->C == 1 && (p->W & 3) == 0 && (((size_t)p->color | (size_t)p->out) & 15) == 0) {
        hipLaunchKernelGGL(k_aa_fill_bin1, dim3(fpcdr_cdiv(p->W, 32), fpcdr_cdiv(p->H, 32), p->B), dim3(256), 0, st, (const float4 *)p->color,
                           p->B, p->H, p->W, (float4 *)p->out, p->hint, p->empty_color);
        filled = 2;      // every pixel holds its un-antialiased value: k_aa_fwd writes the blended ones only
    }
#define LAUNCH_FWD(CS)                                                                                                   \
    hipLaunchKernelGGL(k_aa_fwd<CS>, grid, dim3(256), 0, st, p->color, (const float4 *)p->rast, (const float4 *)p->pos, \
                       p->tri, p->sil, p->B, p->H, p->W, p->C, p->V, p->T, (unsigned long long *)p->flags, p->out, p->hint,   \
                       p->hint ? p->empty_color : nullptr, filled, flags_zeroed)
    if (p->C == 1) LAUNCH_FWD(1);
    else if (p->C == 3) LAUNCH_FWD(3);
    else if (p->C == 4) LAUNCH_FWD(4);
    else LAUNCH_FWD(0);
#undef LAUNCH_FWD
    FPCDR_CHECK_LAUNCH();
    return FPCDR_OK;
}

extern "C" int fpcdr_antialias_bwd(const fpcdr_antialias_bwd_params *p, void *stream) {
    FPCDR_REQUIRE(p != nullptr, "null params");
    FPCDR_REQUIRE(p->color && p->rast && p->pos && p->tri && p->adj && p->dy && p->sil && p->flags && p->grad_color && p->grad_pos,
                  "null pointer");
    FPCDR_REQUIRE(p->B > 0 && p->H > 0 && p->W > 0 && p->C > 0 && p->V > 0 && p->T > 0, "sizes must be positive");
    hipStream_t st = (hipStream_t)stream;
    // bulk: grad_colour = dy (8C bytes per pixel, the HBM-bound part), then the sparse fix-up
    const size_t nfl = (size_t)p->B * p->H * p->W * p->C;
    if ((((uintptr_t)p->dy | (uintptr_t)p->grad_color) & 15) == 0) {
        const size_t n4 = nfl / 4;
        const size_t g = (n4 + (size_t)COPY_U * 256 - 1) / ((size_t)COPY_U * 256);
        FPCDR_REQUIRE(g <= 0x7fffffffull, "tensor too large for one launch");
        hipLaunchKernelGGL(k_copy_f4_chunk, dim3((unsigned)g), dim3(256), 0, st, (const v4f *)p->dy, (v4f *)p->grad_color, n4);
        if (nfl & 3)
            (void)hipMemcpyAsync(p->grad_color + n4 * 4, p->dy + n4 * 4, (nfl & 3) * sizeof(float), hipMemcpyDeviceToDevice, st);
    } else if (hipMemcpyAsync(p->grad_color, p->dy, nfl * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) {
        fpcdr_set_error("fpcdr_antialias_bwd: device copy failed");
        return FPCDR_ELAUNCH;
    }
    const size_t words = (size_t)p->B * p->H * FPCDR_AA_ROW_WORDS(p->W);
    dim3 grid(fpcdr_cdiv((long long)words, 256));
#define LAUNCH_BWD(CS)                                                                                                       \
    hipLaunchKernelGGL(k_aa_bwd_fix<CS>, grid, dim3(256), 0, st, p->color, (const float4 *)p->rast, (const float4 *)p->pos, \
                       p->tri, p->sil, p->dy, p->B, p->H, p->W, p->C, p->V, p->T, (const unsigned long long *)p->flags,      \
                       p->pos_gradient_boost, p->grad_color, p->grad_pos)
    if (p->C == 1) LAUNCH_BWD(1);
    else if (p->C == 3) LAUNCH_BWD(3);
    else if (p->C == 4) LAUNCH_BWD(4);
    else LAUNCH_BWD(0);
#undef LAUNCH_BWD
    FPCDR_CHECK_LAUNCH();
    return FPCDR_OK;
}
