// rasterize forward / backward for gfx950 (MI355X).
//
// Performs the work of `dr.rasterize(glctx, pos_clip, pos_idx, resolution)` at reference
// src/torch/fit.py:151, which the reference delegates to nvdiffrast + the OpenGL hardware
// rasteriser (context at fit.py:484).  MI355X exposes no graphics pipeline, so this is a complete
// software pipeline designed for CDNA4:
//
//   k_setup   one thread per (image, triangle): clip -> 24.8 fixed point (double arithmetic, rules
//             R1-R3 of DESIGN.md) and the float32 depth plane of R6; writes a 40-byte record + an 8-byte
//             pixel bounding box, the 256-triangle chunk box and folds the image-wide bounding box.
//   k_bins    one 256-thread workgroup per 32x32-pixel bin (one 16x16 tile per wave, four pixels per lane).  The bin scans the image's bounding
//             boxes (8 B per triangle, L2 resident), keeps the overlapping triangles IN ORDER in
//             LDS, 256 at a time, expands them to edge equations in LDS and marks which of the bin's
//             64 8x8 tiles each one touches (bit masks in LDS).  Each wave owns 16 tiles, one pixel
//             per lane: coverage by exact integer edge functions, depth by a float32 plane, winner kept in
//             registers -- no global atomics, no per-pixel depth buffer in HBM, and the result does
//             not depend on scheduling (ascending triangle order + strict "<").  The same wave then
//             shades its pixels (perspective-correct barycentrics and their screen-space derivatives
//             in f32) and writes rast / rast_db exactly once, 128-byte row segments per tile row.
//   k_grad    one thread per pixel: recomputes the shading terms, chains (dL/du, dL/dv, dL/d db) to
//             the three clip-space vertices and scatters with wave-level pre-reduction by vertex.
//
// Arithmetic that decides integer outputs uses only + - * / floor on IEEE doubles, exact integers and
// explicit fmaf, and is compiled with -ffp-contract=off, so it agrees bit for bit with oracle/raster_ref.c.
#include "common.h"

#include <stdlib.h>
#include <algorithm>

namespace {

#include "texsample.h"
#include "raster_math.h"
#include "aa_pairs.h"
#include "sil_bits.h"

#ifndef FPCDR_TWOCALL
#define FPCDR_TWOCALL 0
#endif

constexpr int SUBPIX = 256;
constexpr int HALFPIX = 128;
constexpr double GUARD = 16777216.0;  // 2^24

#define FPCDR_BIN 32   // measured at cfg3: 64 -> 5.2 ms, 32 -> 3.6 ms (fused forward): smaller bins balance the rim better
constexpr int BIN = FPCDR_BIN;   // pixels per bin side
constexpr int TILE = 16;         // pixels per tile side; a wave covers a tile with 4 pixels per lane (2x2 quads of 8x8)
constexpr int QUAD = 8;
constexpr int TILES_X = BIN / TILE;
constexpr int NTILES = TILES_X * TILES_X;   // 4 tiles per bin
constexpr int BATCH = 256;       // triangles expanded in LDS at a time (= block size)
constexpr int TILES_PER_WAVE = NTILES / 4;  // 1 (x 4 pixels per lane)
static_assert(TILES_PER_WAVE >= 1, "a bin needs at least one tile per wave");

struct __attribute__((aligned(8))) TriRec {  // 40 bytes
    int32_t X0, Y0, X1, Y1, X2, Y2;   // snapped vertices (R2)
    float zA, zB, z0;                 // depth plane anchored at vertex 0 (R6)
    int32_t tid;                      // the triangle this record is (a piece of): its slot, except for the second piece of a clipped one
};
struct __attribute__((aligned(8))) TriBox {  // inclusive pixel bbox; x0 > x1 = dropped
    int16_t x0, y0, x1, y1;
};
struct ImgBox { int32_t x0, y0, x1, y1, n_over, n_clip, pad1, pad2; };  // box folded with atomicMin/atomicMax; n_over: overflow records (below); n_clip: entries of the image's clip list

// Per-image record slots.  A triangle that crosses the near plane is clipped into one or two pieces (rule R1): the first takes the
// triangle's own slot t, the second is appended to the image's OVERFLOW region -- slots [Tp, Tp + n_over), Tp = T rounded up to whole
// 256-slot chunks -- with an atomic counter.  The region has room for every triangle (it is only ever touched where clipping
// happens: none of it in the fit loop), so the scratch buffer holds 2 Tp slots per image.  Overflow chunks carry no chunk box: a bin
// scans all of them.
__host__ __device__ inline int padded_slots(int T) { return (T + 255) / 256 * 256; }

struct __attribute__((aligned(16))) EdgeRec {  // LDS, 80 bytes
    int32_t A0, B0, A1, B1, A2, B2;
    int32_t nb;       // bit e (0..2) set: edge e does NOT own ties (E == 0 is outside); bit 3: small triangle
    int32_t id;       // triangle index
    union {
        long long C[3];                              // general path: biased constants C_e - nb_e (int64)
        struct { int32_t X1, Y1, X2, Y2, X0, Y0; } v;  // small path: anchor vertex of edge 0, 1, 2
    };
    float zA, zB, z0;   // depth plane (R6), anchored at (X0, Y0)
    int32_t X0, Y0;     // (also for the general path, whose union holds C[])
    int32_t pad[3];
};
static_assert(sizeof(EdgeRec) == 96, "EdgeRec layout");
// A triangle whose extent is at most 64 pixels in x and y takes the 32-bit path: for every pixel of a tile its
// bounding box touches, |P - anchor| <= 16384 + 2048 sub-pixel units and |A|,|B| <= 16384, so each edge
// function fits in int32 and the products fit v_mad_i32_i24.  Same integers, fewer and full-rate instructions.
constexpr int SMALL_EXTENT = 16384;
// Triangles of the 32-bit class whose bounding box covers at most LANE_MAX pixels of the bin are rasterised by ONE LANE
// each (a loop over the box, winners folded into the bin's LDS depth buffer with 64-bit atomic min); the others go
// through the tile path in rounds of BIGB.  On the 30k-triangle rig a triangle's box holds ~50 pixels, a 16x16 tile
// 256: one lane per triangle issues ~4x fewer instructions than one wave per (triangle, tile).
#define FPCDR_LANE_MAX 256
constexpr int LANE_MAX = FPCDR_LANE_MAX;
constexpr int BIGB = 64;         // triangles per round of the tile path
// (r5, measured and dropped: EARLY Z -- the lane path in two halves by the sign of the triangle's area, i.e. the layer of a closed mesh that
//  faces the camera first, the bin's depth buffer summarised as one maximum per 8 x 8 block in between, and a triangle of the second half
//  whose nearest depth lies behind every block its box touches skips its walk.  Exact (ids bit-identical at 1080p / 4K), and slower: 571 ->
//  681 us with the facing layer first, 881 with the other first -- the lanes of a wave walk in lockstep, so two half-empty passes cost two
//  walks unless a whole wave's triangles are culled; compacting the survivors first would need two more barriers and a prefix per batch.)
// latency-bound: 7 waves per SIMD (r2 sweep of the LOSS list kernel: 5 -> 2.23 ms, 6 -> 2.23, 7 -> 2.11, 8 -> 2.20; its 22 KB of LDS allow 7 workgroups per CU)
#define FPCDR_BINS_WPE __attribute__((amdgpu_waves_per_eu(7, 8)))
#define FPCDR_SCAN_K 4
constexpr int SCAN_K = FPCDR_SCAN_K;        // live chunks whose bounding boxes are tested per scan iteration

// (depth, triangle) as one ordered 64-bit key: smaller depth first, ties to the smaller triangle index (R6).
// d + 0.0f turns -0.0 into +0.0, so that the integer order of the keys is the float order of the depths.
__device__ __forceinline__ unsigned long long zpack(float d, int id) {
    unsigned int u = __float_as_uint(d + 0.0f);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ((unsigned long long)u << 32) | (unsigned int)id;
}
constexpr unsigned long long Z_EMPTY = ~0ull;

__device__ __forceinline__ long long floordiv256(long long a) { return a >> 8; }  // arithmetic shift = floor

// ---------------------------------------------------------------------------------------------
// One piece (a triangle, or a piece of a clipped one) in double clip coordinates -> record + pixel box: rules R2, R3, R6.
__device__ __forceinline__ bool setup_piece(const double (&v)[3][4], int H, int W, int tid, TriRec &r, TriBox &box) {
    long long X[3], Y[3];
    double zw[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const double dw = v[i][3];
        if (!(dw > 0.0)) return false;
        const double rw = 1.0 / dw;      // (R2: one double division per vertex, not three: k_setup is bound by vector issue)
        const double xs = v[i][0] * rw;
        const double ys = v[i][1] * rw;
        const double fx = floor((xs * 0.5 + 0.5) * (double)(W * SUBPIX) + 0.5);
        const double fy = floor((ys * 0.5 + 0.5) * (double)(H * SUBPIX) + 0.5);
        if (!(fabs(fx) <= GUARD) || !(fabs(fy) <= GUARD)) return false;
        X[i] = (long long)fx;
        Y[i] = (long long)fy;
        zw[i] = v[i][2] * rw;
    }
    const long long D = (X[1] - X[0]) * (Y[2] - Y[0]) - (Y[1] - Y[0]) * (X[2] - X[0]);
    if (D == 0) return false;
    const double Dd = (double)D;
    const long long xmin = min(X[0], min(X[1], X[2])), xmax = max(X[0], max(X[1], X[2]));
    const long long ymin = min(Y[0], min(Y[1], Y[2])), ymax = max(Y[0], max(Y[1], Y[2]));
    long long px0 = floordiv256(xmin - HALFPIX + SUBPIX - 1), px1 = floordiv256(xmax - HALFPIX);
    long long py0 = floordiv256(ymin - HALFPIX + SUBPIX - 1), py1 = floordiv256(ymax - HALFPIX);
    px0 = max(px0, 0ll); py0 = max(py0, 0ll);
    px1 = min(px1, (long long)W - 1); py1 = min(py1, (long long)H - 1);
    if (px0 > px1 || py0 > py1) return false;
    box = {(int16_t)px0, (int16_t)py0, (int16_t)px1, (int16_t)py1};
    r.X0 = (int32_t)X[0]; r.Y0 = (int32_t)Y[0];
    r.X1 = (int32_t)X[1]; r.Y1 = (int32_t)Y[1];
    r.X2 = (int32_t)X[2]; r.Y2 = (int32_t)Y[2];
    const double dz1 = zw[1] - zw[0], dz2 = zw[2] - zw[0];
    r.zA = (float)((dz1 * (double)(Y[2] - Y[0]) - dz2 * (double)(Y[1] - Y[0])) / Dd);
    r.zB = (float)((dz2 * (double)(X[1] - X[0]) - dz1 * (double)(X[2] - X[0])) / Dd);
    r.z0 = (float)zw[0];
    r.tid = tid;
    return true;
}

// NEAR-PLANE CLIPPING LIVES IN ITS OWN KERNEL (r5).  clip_pieces works on small double arrays indexed at run time -- 400 bytes of private
// segment per lane -- and a kernel with a private segment pays for it on every dispatch, taken or not (the fit loop's rig stays 140 units in
// front of the near plane: never).  k_setup therefore only APPENDS the index of a triangle with a vertex at w <= 0 to its image's clip list
// and leaves the slot dropped; k_setup_clip -- one small workgroup per image, normally finding an empty list -- clips those triangles,
// fills the slot with the first piece (folding its box into the chunk box the set-up kernel stored, the image box and the bin lists) and
// appends the second to the overflow slots.  The rasteriser's result does not depend on the order of records or list entries.
//
// Rule R1 for a triangle with some w <= 0: Sutherland-Hodgman against the near plane z + w >= 0, in double, every crossing computed
// from the vertex inside to the one outside (the same arithmetic, in the same order, as oracle/raster_ref.c clip_pieces).  Returns the
// number of pieces (0 = dropped) and their vertices.
__device__ __noinline__ int clip_pieces(const float4 (&v)[3], double (&pc)[2][3][4]) {
    double d[3], poly[4][4], vd[3][4];
    for (int i = 0; i < 3; ++i) {
        vd[i][0] = (double)v[i].x; vd[i][1] = (double)v[i].y; vd[i][2] = (double)v[i].z; vd[i][3] = (double)v[i].w;
        d[i] = vd[i][2] + vd[i][3];
        if (!(d[i] == d[i])) return 0;
    }
    int n = 0;
    for (int i = 0; i < 3; ++i) {
        const int j = (i + 1) % 3;
        const bool in_i = d[i] >= 0.0, in_j = d[j] >= 0.0;
        if (in_i) {
            for (int c = 0; c < 4; ++c) poly[n][c] = vd[i][c];
            ++n;
        }
        if (in_i != in_j) {
            const int a = in_i ? i : j, bb = in_i ? j : i;
            const double t = d[a] / (d[a] - d[bb]);
            for (int c = 0; c < 4; ++c) poly[n][c] = vd[a][c] + t * (vd[bb][c] - vd[a][c]);
            ++n;
        }
    }
    if (n < 3) return 0;
    for (int i = 0; i < n; ++i)
        if (!(poly[i][3] > 0.0)) return 0;
    for (int k = 0; k + 2 < n; ++k)
        for (int c = 0; c < 4; ++c) { pc[k][0][c] = poly[0][c]; pc[k][1][c] = poly[k + 1][c]; pc[k][2][c] = poly[k + 2][c]; }
    return n - 2;
}

// Per-bin triangle lists (BINLIST instantiation, the one-pass objective): the raster kernel's search for "which triangles touch my bin"
// -- chunk boxes, then the 8-byte boxes of ~2 000 triangles of the live chunks, three barriers -- was 290 us of its 700 at cfg3 for a
// quarter of its instructions: a chain of dependent loads, not work (profiles/r04_raster_ablation.txt).  Here the set-up kernel, which
// holds every triangle's box anyway, bins its 256 triangles over the <= BL_LOCAL bins of its chunk box in LDS (count, one global
// fetch-add per touched bin for the segment's place in the bin's list, fill) and the raster kernel reads its list: one count, one
// gather.  A bin's list holds BL_CAP record slots; a bin that overflows (thousands of tiny triangles in one bin) keeps its count and
// falls back to the chunk scan.  Order inside a list depends on scheduling; the winner of a pixel -- min over (depth, index) -- does not.
constexpr int BL_CAP = 256;        // entries per bin list (a bin of the 30k rig holds ~120)
constexpr int BL_LOCAL = 64;       // local bins of a chunk box binned in LDS (8 x 8); larger chunk boxes: per-triangle global appends

__device__ __forceinline__ void binlist_append_global(int32_t *__restrict__ bin_cnt, int32_t *__restrict__ bin_list, uint8_t *__restrict__ live,
                                                      size_t img_bins, int OX, TriBox bx, int slot) {
    for (int gy = bx.y0 / BIN; gy <= bx.y1 / BIN; ++gy)
        for (int gx = bx.x0 / BIN; gx <= bx.x1 / BIN; ++gx) {
            const size_t gb = img_bins + (size_t)gy * OX + gx;
            const int pos = atomicAdd(&bin_cnt[gb], 1);
            if (pos < BL_CAP) bin_list[gb * BL_CAP + pos] = slot;
            live[gb] = 1;
        }
}

// SIL: the thread also classifies its triangle's three edges for the antialias step (sil_bits.h) -- it holds the triangle's own three
// vertices already, the three across the edges are gathered beside them (the stand-alone kernel, k_sil2, is 62 us at 288 x 30 k of
// which this form leaves ~10: the same chain of index load -> position gather, walked once instead of twice).
template <bool BINLIST = false, bool SIL = false>
__global__ void __launch_bounds__(256) k_setup(const float4 *__restrict__ pos, const int32_t *__restrict__ tri,
                                                int B, int V, int T, int H, int W, TriRec *__restrict__ recs,
                                                TriBox *__restrict__ boxes, TriBox *__restrict__ cboxes,
                                                ImgBox *__restrict__ ibox, uint8_t *__restrict__ live,
                                                const int32_t *__restrict__ ranges, int32_t *__restrict__ bin_cnt = nullptr,
                                                int32_t *__restrict__ bin_list = nullptr, const int32_t *__restrict__ adj = nullptr,
                                                uint8_t *__restrict__ sil = nullptr, int32_t *__restrict__ clip = nullptr,
                                                uint4 *__restrict__ zero16 = nullptr, long long n_zero16 = 0) {
    // (one-pass objective: the caller's position-gradient table -- 69 MB at 288 x 15 k vertices -- is zero-filled HERE, a 16-byte store or
    //  two per thread beside the gathers this kernel waits for, instead of by the call's first kernel, where it was 12 us on its own)
    if (zero16) {
        const long long nthreads = (long long)gridDim.x * gridDim.y * blockDim.x;
        for (long long i = ((long long)blockIdx.y * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x; i < n_zero16; i += nthreads)
            zero16[i] = make_uint4(0u, 0u, 0u, 0u);
    }
    // grid: x over 256-triangle chunks, y = image.  A block never straddles two images, so the union of
    // its triangles' bounding boxes can be reduced in the block: it is stored as the CHUNK box (meshes
    // keep neighbouring triangles at neighbouring indices, so a bin later skips most chunks with one
    // test) and folded into the image box with at most four atomics per block.
    __shared__ int s_box[4][4];
    const int b = blockIdx.y;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int Tp = padded_slots(T);
    const size_t img_slot = (size_t)b * 2 * Tp;
    const int OX = (W + BIN - 1) / BIN, OY = (H + BIN - 1) / BIN;
    int bx0 = 0x7fffffff, by0 = 0x7fffffff, bx1 = -1, by1 = -1;
    if (t < T) {
        const size_t gid = img_slot + t;
        TriBox box = {1, 1, 0, 0};
        int i0 = tri[3 * t], i1 = tri[3 * t + 1], i2 = tri[3 * t + 2];
        bool ok = !(i0 < 0 || i0 >= V || i1 < 0 || i1 >= V || i2 < 0 || i2 >= V);
        if (ranges) {      // range mode: the image renders its own slice of the triangle list
            const long long first = ranges[2 * b], count = ranges[2 * b + 1];
            ok = ok && t >= first && t < first + count;
        }
        if (SIL) {      // (index validity only decides, as in sil_classify: `ok` below may also carry the range test)
            const bool vok = !(i0 < 0 || i0 >= V || i1 < 0 || i1 >= V || i2 < 0 || i2 >= V);
            unsigned int bits = 0;
            if (vok) {
                const int ad[3] = {adj[3 * t], adj[3 * t + 1], adj[3 * t + 2]};
                const float4 *p = pos + (size_t)b * V;
                const float4 o0 = ld32(p, (ad[0] >= 0 && ad[0] < V) ? (unsigned int)ad[0] : 0u);
                const float4 o1 = ld32(p, (ad[1] >= 0 && ad[1] < V) ? (unsigned int)ad[1] : 0u);
                const float4 o2 = ld32(p, (ad[2] >= 0 && ad[2] < V) ? (unsigned int)ad[2] : 0u);
                bits = sil_bits_of(p[i0], p[i1], p[i2], o0, o1, o2, ad, V, 0.5f * (float)W, 0.5f * (float)H);
            }
            sil[(size_t)b * T + t] = (uint8_t)bits;
        }
        if (ok) {
            const float4 *p = pos + (size_t)b * V;
            const float4 v0 = p[i0], v1 = p[i1], v2 = p[i2];      // (kept as scalars: an array handed to the out-of-line clipper would
            TriRec r;                                             // live in scratch memory on the common path too)
            if (v0.w > 0.0f && v1.w > 0.0f && v2.w > 0.0f) {      // (R1) the usual case: the triangle as it is
                const double vd[3][4] = {{(double)v0.x, (double)v0.y, (double)v0.z, (double)v0.w},
                                         {(double)v1.x, (double)v1.y, (double)v1.z, (double)v1.w},
                                         {(double)v2.x, (double)v2.y, (double)v2.z, (double)v2.w}};
                if (setup_piece(vd, H, W, t, r, box)) {
                    recs[gid] = r;
                    bx0 = box.x0; by0 = box.y0; bx1 = box.x1; by1 = box.y1;
                } else {
                    box = {1, 1, 0, 0};
                }
            } else {      // a vertex at or behind w = 0: k_setup_clip's (the slot stays dropped until then)
                clip[(size_t)b * Tp + atomicAdd(&ibox[b].n_clip, 1)] = t;
            }
        }
        boxes[gid] = box;
    }
    const TriBox mybox = {(int16_t)(bx1 >= 0 ? bx0 : 1), (int16_t)(bx1 >= 0 ? by0 : 1), (int16_t)(bx1 >= 0 ? bx1 : 0), (int16_t)(bx1 >= 0 ? by1 : 0)};
    // image bounding box: wave reduce -> block reduce -> at most four atomics per block
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        bx0 = min(bx0, __shfl_xor(bx0, o, 64)); by0 = min(by0, __shfl_xor(by0, o, 64));
        bx1 = max(bx1, __shfl_xor(bx1, o, 64)); by1 = max(by1, __shfl_xor(by1, o, 64));
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_box[wave][0] = bx0; s_box[wave][1] = by0; s_box[wave][2] = bx1; s_box[wave][3] = by1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) {
            bx0 = min(bx0, s_box[w][0]); by0 = min(by0, s_box[w][1]);
            bx1 = max(bx1, s_box[w][2]); by1 = max(by1, s_box[w][3]);
        }
        TriBox cb = {1, 1, 0, 0};
        if (bx1 >= 0) {
            cb = {(int16_t)bx0, (int16_t)by0, (int16_t)bx1, (int16_t)by1};
            atomicMin(&ibox[b].x0, bx0); atomicMin(&ibox[b].y0, by0);
            atomicMax(&ibox[b].x1, bx1); atomicMax(&ibox[b].y1, by1);
        }
        cboxes[(size_t)b * gridDim.x + blockIdx.x] = cb;
        if (live) { s_box[0][0] = bx0; s_box[0][1] = by0; s_box[0][2] = bx1; s_box[0][3] = by1; }
    }
    if (BINLIST) {
        // ---- per-bin lists: the chunk's triangles over the bins of the chunk box ----
        __shared__ int s_cnt[BL_LOCAL], s_base[BL_LOCAL];
        __syncthreads();
        const int cx0 = s_box[0][0], cy0 = s_box[0][1], cx1 = s_box[0][2], cy1 = s_box[0][3];
        if (cx1 < 0) return;                                  // (uniform) no triangle of this chunk reaches the image
        const int gx0 = cx0 / BIN, gy0 = cy0 / BIN, nx = cx1 / BIN - gx0 + 1, ny = cy1 / BIN - gy0 + 1;
        const size_t img_bins = (size_t)b * OY * OX;
        const bool has = mybox.x0 <= mybox.x1;
        if (nx * ny > BL_LOCAL) {                             // (uniform) a huge chunk box: every triangle appends for itself
            if (has) binlist_append_global(bin_cnt, bin_list, live, img_bins, OX, mybox, t);
            return;
        }
        if ((int)threadIdx.x < BL_LOCAL) s_cnt[threadIdx.x] = 0;
        __syncthreads();
        const int tx0 = has ? mybox.x0 / BIN - gx0 : 0, tx1 = has ? mybox.x1 / BIN - gx0 : -1;
        const int ty0 = has ? mybox.y0 / BIN - gy0 : 0, ty1 = has ? mybox.y1 / BIN - gy0 : -1;
        for (int ly = ty0; ly <= ty1; ++ly)
            for (int lxb = tx0; lxb <= tx1; ++lxb) atomicAdd(&s_cnt[ly * nx + lxb], 1);
        __syncthreads();
        if ((int)threadIdx.x < nx * ny) {
            const int lb = threadIdx.x, c = s_cnt[lb];
            if (c > 0) {
                const size_t gb = img_bins + (size_t)(gy0 + lb / nx) * OX + gx0 + lb % nx;
                s_base[lb] = atomicAdd(&bin_cnt[gb], c);      // this chunk's segment of the bin's list
                live[gb] = 1;
            }
            s_cnt[lb] = 0;                                    // (reused as the fill cursor)
        }
        __syncthreads();
        for (int ly = ty0; ly <= ty1; ++ly)
            for (int lxb = tx0; lxb <= tx1; ++lxb) {
                const int lb = ly * nx + lxb;
                const int pos = s_base[lb] + atomicAdd(&s_cnt[lb], 1);
                if (pos < BL_CAP) bin_list[(img_bins + (size_t)(gy0 + ly) * OX + gx0 + lxb) * BL_CAP + pos] = t;
            }
        return;
    }
    if (live) {
        // work-queue mode: every bin the chunk box touches becomes a work item of k_bins_queue (a superset of the bins
        // some triangle's box touches; all writers store 1)
        __syncthreads();
        const int cx0 = s_box[0][0], cy0 = s_box[0][1], cx1 = s_box[0][2], cy1 = s_box[0][3];
        if (cx1 >= 0) {
            const int gx0 = cx0 / BIN, gy0 = cy0 / BIN, nx = cx1 / BIN - gx0 + 1, ny = cy1 / BIN - gy0 + 1;
            for (int k = threadIdx.x; k < nx * ny; k += blockDim.x)
                live[((size_t)b * OY + gy0 + k / nx) * OX + gx0 + k % nx] = 1;
        }
    }
}

// The triangles k_setup put on the clip lists (see clip_pieces): grid (1, B), the workgroup strides over its image's list.
template <bool BINLIST>
__global__ void __launch_bounds__(256) k_setup_clip(const float4 *__restrict__ pos, const int32_t *__restrict__ tri, int V, int T, int H, int W,
                                                     TriRec *__restrict__ recs, TriBox *__restrict__ boxes, TriBox *__restrict__ cboxes,
                                                     ImgBox *__restrict__ ibox, uint8_t *__restrict__ live, int32_t *__restrict__ bin_cnt,
                                                     int32_t *__restrict__ bin_list, const int32_t *__restrict__ clip) {
    const int b = blockIdx.x;
    const int n = ibox[b].n_clip;
    if (n == 0) return;
    const int Tp = padded_slots(T);
    const size_t img_slot = (size_t)b * 2 * Tp;
    const int OX = (W + BIN - 1) / BIN, OY = (H + BIN - 1) / BIN;
    const size_t img_bins = (size_t)b * OY * OX;
    const float4 *p = pos + (size_t)b * V;
    auto mark = [&](const TriBox &bx, int slot) {      // image box, bin lists / live map of one piece
        atomicMin(&ibox[b].x0, (int)bx.x0); atomicMin(&ibox[b].y0, (int)bx.y0);
        atomicMax(&ibox[b].x1, (int)bx.x1); atomicMax(&ibox[b].y1, (int)bx.y1);
        if (BINLIST) binlist_append_global(bin_cnt, bin_list, live, img_bins, OX, bx, slot);
        else if (live)
            for (int gy = bx.y0 / BIN; gy <= bx.y1 / BIN; ++gy)
                for (int gx = bx.x0 / BIN; gx <= bx.x1 / BIN; ++gx) live[img_bins + (size_t)gy * OX + gx] = 1;
    };
    for (int k = threadIdx.x; k < n; k += blockDim.x) {
        const int t = clip[(size_t)b * Tp + k];
        const float4 vv[3] = {p[tri[3 * t]], p[tri[3 * t + 1]], p[tri[3 * t + 2]]};      // (k_setup checked the indices)
        double pc[2][3][4];
        const int np = clip_pieces(vv, pc);
        TriRec r;
        TriBox box;
        if (np >= 1 && setup_piece(pc[0], H, W, t, r, box)) {
            // first piece: the triangle's own slot; its box joins the chunk box k_setup stored (bins skip chunks by that box)
            recs[img_slot + t] = r;
            boxes[img_slot + t] = box;
            unsigned long long *cb = reinterpret_cast<unsigned long long *>(cboxes + (size_t)b * (Tp / 256) + t / 256);
            unsigned long long seen = *cb;
            for (;;) {
                TriBox o;
                __builtin_memcpy(&o, &seen, 8);
                TriBox u = box;
                if (o.x0 <= o.x1) u = {(int16_t)min(o.x0, box.x0), (int16_t)min(o.y0, box.y0), (int16_t)max(o.x1, box.x1), (int16_t)max(o.y1, box.y1)};
                unsigned long long want;
                __builtin_memcpy(&want, &u, 8);
                const unsigned long long got = atomicCAS(cb, seen, want);
                if (got == seen) break;
                seen = got;
            }
            mark(box, t);
        }
        if (np == 2 && setup_piece(pc[1], H, W, t, r, box)) {
            // second piece: appended to the image's overflow slots (no chunk box there: bins scan every overflow chunk)
            const int q = atomicAdd(&ibox[b].n_over, 1);
            recs[img_slot + Tp + q] = r;
            boxes[img_slot + Tp + q] = box;
            mark(box, Tp + q);
        }
    }
}

__global__ void k_init_ibox(ImgBox *ibox, int B) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B) ibox[i] = {0x7fffffff, 0x7fffffff, -1, -1, 0, 0, 0, 0};
}

// work-queue mode: image boxes + zeroed live map, raw occupancy map and queue header (grid-stride)
__global__ void __launch_bounds__(256) k_init_queue(ImgBox *ibox, int B, uint32_t *__restrict__ live_words, long long n_live_words,
                                                    uint32_t *__restrict__ occ_words, long long n_occ_words, int32_t *__restrict__ hdr,
                                                    int32_t *__restrict__ hdr_bwd, int32_t *__restrict__ bin_cnt = nullptr, long long n_bins = 0) {
    const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x, stride = (long long)gridDim.x * blockDim.x;
    if (i0 < B) ibox[i0] = {0x7fffffff, 0x7fffffff, -1, -1, 0, 0, 0, 0};
    if (i0 < 16) { hdr[i0] = 0; hdr_bwd[i0] = 0; }
    for (long long i = i0; i < n_live_words; i += stride) live_words[i] = 0u;
    for (long long i = i0; i < n_occ_words; i += stride) occ_words[i] = 0u;
    if (bin_cnt)
        for (long long i = i0; i < n_bins; i += stride) bin_cnt[i] = 0;      // per-bin triangle counts (k_setup<true>)
}

// the same for the one-pass objective, plus the caller's output buffers (zero_outputs): the gradient tables are 70 MB at 288 x 15 k
// vertices, so 16-byte stores and a grid that fills the chip
__global__ void __launch_bounds__(256) k_init_objective(ImgBox *ibox, int B, uint32_t *__restrict__ live_words, long long n_live_words,
                                                        uint32_t *__restrict__ occ_words, long long n_occ_words, int32_t *__restrict__ hdr,
                                                        int32_t *__restrict__ hdr_bwd, int32_t *__restrict__ bin_cnt, long long n_bins,
                                                        FpcdrZeroList zl) {
    const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x, stride = (long long)gridDim.x * blockDim.x;
    if (i0 < B) ibox[i0] = {0x7fffffff, 0x7fffffff, -1, -1, 0, 0, 0, 0};
    if (i0 < 16) { hdr[i0] = 0; hdr_bwd[i0] = 0; }
    for (long long i = i0; i < n_live_words; i += stride) live_words[i] = 0u;
    for (long long i = i0; i < n_occ_words; i += stride) occ_words[i] = 0u;
    if (bin_cnt)
        for (long long i = i0; i < n_bins; i += stride) bin_cnt[i] = 0;
    for (int r = 0; r < zl.count; ++r) {
        uint32_t *p = zl.p[r];
        const long long n = zl.n[r];
        long long done = 0;
        if (((size_t)p & 15) == 0) {
            uint4 *q = reinterpret_cast<uint4 *>(p);
            const long long n4 = n >> 2;
            const uint4 z = make_uint4(0u, 0u, 0u, 0u);
            long long i = i0;
            for (; i + 3 * stride < n4; i += 4 * stride) { q[i] = z; q[i + stride] = z; q[i + 2 * stride] = z; q[i + 3 * stride] = z; }
            for (; i < n4; i += stride) q[i] = z;
            done = n4 << 2;
        }
        for (long long i = done + i0; i < n; i += stride) p[i] = 0u;
    }
}

// ---- ordered compaction of per-bin flags into lists (ascending bin index: neighbouring bins, which share vertices and
// texels, stay neighbours in launch order; appending with atomics scrambles them at wave granularity and cost the
// backward pass 10 %) -- three tiny kernels: per-block counts, one-workgroup scan, ordered write ----
constexpr int NLISTS = 2;   // lists built per round (round A: live bins; round B: antialias-fix bins, backward bins)

// flags of bin i for round A (live map) / round B (window mask m of the raw occupancy map)
__device__ __forceinline__ unsigned int window_mask(const uint8_t *__restrict__ raw, long long i, int OY, int OX) {
    const int x = (int)(i % OX), y = (int)((i / OX) % OY);
    const uint8_t *img = raw + (i - (long long)y * OX - x);
    unsigned int m = 0;
    for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 2; ++dx) {
            const int cx = x + dx, cy = y + dy;
            if (cx >= 0 && cx < OX && cy >= 0 && cy < OY && img[cy * OX + cx]) m |= 1u << ((dy + 1) * 4 + dx + 1);
        }
    return m;
}
__device__ __forceinline__ bool wbit(unsigned int m, int dx, int dy) { return (m >> ((dy + 1) * 4 + dx + 1)) & 1u; }

// ROUND_B: 0 = round A (live map); 1 = round B of the two-pass objective; 2 = round B of the one-pass objective (objective.hip):
// list 0 = the occupied bins themselves, no second list
template <int ROUND_B>
__device__ __forceinline__ void bin_flags(const uint8_t *__restrict__ map, long long i, long long nbins, int OY, int OX, bool (&f)[NLISTS],
                                          unsigned int &m) {
    f[0] = false; f[1] = false; m = 0;
    if (i >= nbins) return;
    if (!ROUND_B) { f[0] = map[i] != 0; return; }
    m = window_mask(map, i, OY, OX);
    if (ROUND_B == 2) { f[0] = wbit(m, 0, 0); return; }
    f[0] = wbit(m, 0, 0) || wbit(m, 1, 0) || wbit(m, 0, 1);                                        // k_aa_fix
    f[1] = f[0] || wbit(m, -1, 0) || wbit(m, 0, -1);                                              // k_render_aa_bwd
}

template <int ROUND_B>
__global__ void __launch_bounds__(256) k_list_count(const uint8_t *__restrict__ map, long long nbins, int OY, int OX,
                                                    int32_t *__restrict__ blk_counts, uint16_t *__restrict__ win,
                                                    const float *__restrict__ tex, int Ht, int Wt, int C, int boundary,
                                                    float *__restrict__ empty_out) {
    __shared__ int s_c[NLISTS][4];
    if (!ROUND_B && blockIdx.x == 0 && threadIdx.x == 0) {   // the colour of an empty pixel: the texture at uv = (0,0)
        const Taps tp0 = make_taps(0.0f, 0.0f, Ht, Wt, C, boundary);
        for (int c = 0; c < 4; ++c) empty_out[c] = c < C ? bilerp(tex, tp0, c, C) : 0.0f;
    }
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    bool f[NLISTS];
    unsigned int m;
    bin_flags<ROUND_B>(map, i, nbins, OY, OX, f, m);
    if (ROUND_B && i < nbins) win[i] = (uint16_t)m;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NLISTS; ++k) {
        const int c = __popcll(__ballot(f[k]));
        if (lane == 0) s_c[k][wave] = c;
    }
    __syncthreads();
    if (threadIdx.x < NLISTS)
        blk_counts[(size_t)threadIdx.x * gridDim.x + blockIdx.x] = s_c[threadIdx.x][0] + s_c[threadIdx.x][1] + s_c[threadIdx.x][2] + s_c[threadIdx.x][3];
}

// one workgroup: exclusive scan of the per-block counts of each list, in place; totals to count[k]
__global__ void __launch_bounds__(1024) k_list_scan(int32_t *__restrict__ blk_counts, int nblk, int32_t *__restrict__ count0,
                                                    int32_t *__restrict__ count1) {
    __shared__ int s_w[16];
    __shared__ int s_carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int k = 0; k < NLISTS; ++k) {
        int32_t *c = blk_counts + (size_t)k * nblk;
        if (tid == 0) s_carry = 0;
        __syncthreads();
        for (int base = 0; base < nblk; base += 1024) {
            const int i = base + tid;
            const int v = i < nblk ? c[i] : 0;
            int incl = v;      // inclusive scan inside the wave
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_up(incl, o, 64);
                if (lane >= o) incl += t;
            }
            if (lane == 63) s_w[wave] = incl;
            __syncthreads();
            int woff = 0;
            for (int w = 0; w < wave; ++w) woff += s_w[w];
            const int carry = s_carry;
            if (i < nblk) c[i] = carry + woff + incl - v;
            __syncthreads();
            if (tid == 1023) s_carry = carry + woff + incl;
            __syncthreads();
        }
        if (tid == 0) { int32_t *dst = k == 0 ? count0 : count1; if (dst) *dst = s_carry; }
        __syncthreads();
    }
}

template <int ROUND_B>
__global__ void __launch_bounds__(256) k_list_write(const uint8_t *__restrict__ map, long long nbins, int OY, int OX,
                                                    const int32_t *__restrict__ blk_offsets, int32_t *__restrict__ list0,
                                                    int32_t *__restrict__ list1) {
    __shared__ int s_c[NLISTS][4];
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    bool f[NLISTS];
    unsigned int m;
    bin_flags<ROUND_B>(map, i, nbins, OY, OX, f, m);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long bal[NLISTS];
#pragma unroll
    for (int k = 0; k < NLISTS; ++k) {
        bal[k] = __ballot(f[k]);
        if (lane == 0) s_c[k][wave] = __popcll(bal[k]);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NLISTS; ++k) {
        int32_t *list = k == 0 ? list0 : list1;
        if (!list || !f[k]) continue;
        int off = blk_offsets[(size_t)k * gridDim.x + blockIdx.x];
        for (int w = 0; w < wave; ++w) off += s_c[k][w];
        list[off + __popcll(bal[k] & ((1ull << lane) - 1ull))] = (int32_t)i;
    }
}

// ---- the same ordered compaction as TWO launches (one-pass objective: each round's three launches sit in the call's serial path):
// k_list_count as above, then a write kernel in which every workgroup sums the counts of the workgroups before it itself (2 295 counts at
// 288 x 1080p: nine loads per thread) instead of reading offsets a one-workgroup scan kernel has left.  Up to LW_MAX_BLOCKS workgroups;
// the three-launch form beyond.  (A ONE-launch form -- workgroups publish their totals with a valid bit and wait for their predecessors'
// -- was measured at 587 k bins: 21 / 33 us per round with 2 048 bins per workgroup, 40 / 41 us with 256, against 17 / 24 us for the three
// launches: the waiting costs more than the launches.)
constexpr int LW_MAX_BLOCKS = 16384;
template <int ROUND_B>
__global__ void __launch_bounds__(256) k_list_write_sum(const uint8_t *__restrict__ map, long long nbins, int OY, int OX,
                                                        const int32_t *__restrict__ blk_counts, int32_t *__restrict__ list0,
                                                        int32_t *__restrict__ list1, int32_t *__restrict__ count0, int32_t *__restrict__ count1) {
    __shared__ int s_c[NLISTS][4];
    __shared__ int s_pre[NLISTS][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nblk = gridDim.x, b = blockIdx.x;
    int pre[NLISTS] = {0, 0};
    for (int j = tid; j < b; j += 256) {
        pre[0] += blk_counts[j];
        if (list1) pre[1] += blk_counts[(size_t)nblk + j];
    }
    const long long i = (long long)b * 256 + tid;
    bool f[NLISTS];
    unsigned int m;
    bin_flags<ROUND_B>(map, i, nbins, OY, OX, f, m);
    unsigned long long bal[NLISTS];
#pragma unroll
    for (int k = 0; k < NLISTS; ++k) {
        bal[k] = __ballot(f[k]);
        int v = pre[k];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0) { s_c[k][wave] = __popcll(bal[k]); s_pre[k][wave] = v; }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NLISTS; ++k) {
        int32_t *list = k == 0 ? list0 : list1;
        if (!list) continue;
        const int base = s_pre[k][0] + s_pre[k][1] + s_pre[k][2] + s_pre[k][3];
        int off = base;
        for (int w = 0; w < wave; ++w) off += s_c[k][w];
        if (f[k]) list[off + __popcll(bal[k] & ((1ull << lane) - 1ull))] = (int32_t)i;
        if (b == nblk - 1 && tid == 0) {
            int32_t *dst = k == 0 ? count0 : count1;
            if (dst) *dst = base + s_c[k][0] + s_c[k][1] + s_c[k][2] + s_c[k][3];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Fine raster of one 16x16 tile against the triangles of the current batch whose bit is set in `mrow`.  Each lane
// owns FOUR pixels of the tile -- (lx,ly) in each 8x8 quadrant -- so one LDS fetch of a triangle feeds four
// independent coverage / depth chains (the loop is latency-bound, not ALU-bound).  (Px,Py) is the lane's sample
// point in quadrant 0; quadrant q adds (q&1, q>>1) * 8 pixels.  SMALL selects the 32-bit path (SMALL_EXTENT).
template <bool SMALL>
__device__ __forceinline__ void fine_tile(const unsigned long long *mrow, const EdgeRec *s_tri, int Px, int Py, float (&bd)[4],
                                          int (&bi)[4]) {
    constexpr int STEP = QUAD * SUBPIX;   // 2048 sub-pixel units between quadrants
    for (int wd = 0; wd < BIGB / 64; ++wd) {
        unsigned long long m = mrow[wd];
        const unsigned int mlo = __builtin_amdgcn_readfirstlane((unsigned int)m);
        const unsigned int mhi = __builtin_amdgcn_readfirstlane((unsigned int)(m >> 32));
        m = ((unsigned long long)mhi << 32) | mlo;
        while (m) {
            const int j = __builtin_ctzll(m);
            m &= m - 1;
            const EdgeRec &e = s_tri[wd * 64 + j];
            const int nb = e.nb;
            const int id = e.id;
            const float zA = e.zA, zB = e.zB, z0 = e.z0;
            const int rx = Px - e.X0, ry = Py - e.Y0;   // offset from the depth plane's anchor (quadrant 0)
            if (SMALL) {
                // E'_e = A (Px - Xa) + B (Py - Ya) - nb_e in int32 with 24-bit multiplies; other quadrants by addition
                const int A0 = e.A0, B0 = e.B0, A1 = e.A1, B1 = e.B1, A2 = e.A2, B2 = e.B2;
                const int b0 = __mul24(A0, Px - e.v.X1) + __mul24(B0, Py - e.v.Y1) - (nb & 1);
                const int b1 = __mul24(A1, Px - e.v.X2) + __mul24(B1, Py - e.v.Y2) - ((nb >> 1) & 1);
                const int b2 = __mul24(A2, rx) + __mul24(B2, ry) - ((nb >> 2) & 1);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int E0 = b0 + ((q & 1) ? A0 * STEP : 0) + ((q >> 1) ? B0 * STEP : 0);
                    const int E1 = b1 + ((q & 1) ? A1 * STEP : 0) + ((q >> 1) ? B1 * STEP : 0);
                    const int E2 = b2 + ((q & 1) ? A2 * STEP : 0) + ((q >> 1) ? B2 * STEP : 0);
                    if ((E0 | E1 | E2) >= 0) {
                        const float d = __fmaf_rn(zA, (float)(rx + (q & 1) * STEP), __fmaf_rn(zB, (float)(ry + (q >> 1) * STEP), z0));
                        if (d >= -1.0f && d <= 1.0f && (d < bd[q] || (d == bd[q] && id < bi[q]))) { bd[q] = d; bi[q] = id; }
                    }
                }
            } else {
                const long long A0 = e.A0, B0 = e.B0, A1 = e.A1, B1 = e.B1, A2 = e.A2, B2 = e.B2;
                const long long b0 = A0 * Px + (B0 * Py + e.C[0]);
                const long long b1 = A1 * Px + (B1 * Py + e.C[1]);
                const long long b2 = A2 * Px + (B2 * Py + e.C[2]);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const long long E0 = b0 + ((q & 1) ? A0 * STEP : 0) + ((q >> 1) ? B0 * STEP : 0);
                    const long long E1 = b1 + ((q & 1) ? A1 * STEP : 0) + ((q >> 1) ? B1 * STEP : 0);
                    const long long E2 = b2 + ((q & 1) ? A2 * STEP : 0) + ((q >> 1) ? B2 * STEP : 0);
                    if ((E0 | E1 | E2) >= 0) {
                        const float d = __fmaf_rn(zA, (float)(rx + (q & 1) * STEP), __fmaf_rn(zB, (float)(ry + (q >> 1) * STEP), z0));
                        if (d >= -1.0f && d <= 1.0f && (d < bd[q] || (d == bd[q] && id < bi[q]))) { bd[q] = d; bi[q] = id; }
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// SHADE: the fused render path (fpcdr_render_fwd) -- the wave that resolved a pixel also interpolates its
// texture coordinate and taps the texture, so texc never exists in HBM.
struct ShadeArgs {
    const float2 *uv;        // [Vt] texture coordinates
    const int32_t *uv_tri;   // [T,3]
    const float *tex;        // [Ht,Wt,C]
    float *color;            // out [B,H,W,C]
    int Ht, Wt, C, boundary;
    uint8_t *occ;            // sparse mode (else null): out [B, gridDim.y, gridDim.x], 1 = some triangle's bounding box touches
                             // the bin.  Bins with 0 are NOT written; the consumers (fused.hip) treat their pixels as empty.
                             // (k_occ_window turns this byte map into the per-bin window masks the consumers read)
    float *empty_out;        // sparse mode: out [4], the colour an empty pixel gets (texture at uv = (0,0))
    const float2 *tri_uv;    // optional [T,3]: uv[uv_tri] pre-gathered
    // LOSS (fpcdr_render_loss_fwd): background + squared error of every pixel that antialiasing cannot touch
    const uint8_t *sil;      // [B,T] silhouette bits (k_sil2)
    const uint8_t *ref;      // [B,H,W]
    float *g_aa;             // out [B,H,W,C]
    uint32_t *cmask;         // out [B*bins][32]: row masks of the bin's CANDIDATE pixels, left to k_aa_fix
    unsigned long long *edges;   // out [B*bins][4][32]: (z/w bits << 32 | silhouette bits << 24 | id + 1) of the four border lines
    double *loss_sum;        // [FPCDR_LOSS_SLOTS]
    float bg, color_scale, grad_scale;
    uint8_t *op_hint;        // operator form (fpcdr_rasterize_fwd): out, plane 0 of the region hint, or null
    // MIP instantiations (the reference's enable_mip branch, fit.py:153-155): tex = level 0, mip[l - 1] = level l of the chain
    const float *mip[FPCDR_MAX_MIP];
    int n_levels;
    // IDS instantiation (one-pass objective, objective.hip): out, per-bin planes of (triangle + 1) | silhouette bits << 24, [bin][32][32]
    uint32_t *idp;
    // per-bin triangle lists of k_setup<true> (or null: every bin scans the chunk boxes)
    const int32_t *bin_cnt, *bin_list;
};

// texture coordinates of triangle t through the index buffer (callers that did not pre-gather uv[uv_tri])
__device__ __noinline__ void uv_indirect(const float2 *__restrict__ uv, const int32_t *__restrict__ uv_tri, int t, float2 &q0, float2 &q1,
                                         float2 &q2) {
    q0 = uv[uv_tri[3 * t]]; q1 = uv[uv_tri[3 * t + 1]]; q2 = uv[uv_tri[3 * t + 2]];
}

// One 32x32 bin (bxi, byi) of image b; OX x OY bins per image.  Every branch that leaves is uniform over the workgroup.
// CS / BMODE: channel count and texture boundary mode as compile-time constants (0 / -1 = read them from ShadeArgs); the list
// kernels of the objective are instantiated for the reference's case (one channel, 'wrap'), which strips the channel loops, the
// index scaling and the mode branches from the ~200 instructions a shaded pixel costs
// IDS: raster only -- the bin's winners leave as a 4 KB plane of (triangle + 1) | silhouette bits << 24 (sh.idp) and nothing else is written
template <bool WRITE_DB, bool SHADE, bool LOSS, bool QUEUE, int CS = 0, int BMODE = -1, bool MIP = false, bool IDS = false>
__device__ __forceinline__ void bins_body(const int b, const int bxi, const int byi, const int OX, const int OY,
                                          const float4 *__restrict__ pos, const int32_t *__restrict__ tri,
                                          int V, int T, int H, int W, const TriRec *__restrict__ recs,
                                          const TriBox *__restrict__ boxes, const TriBox *__restrict__ cboxes,
                                          const ImgBox *__restrict__ ibox, float4 *__restrict__ rast,
                                          float4 *__restrict__ rast_db, const ShadeArgs &sh) {
    __shared__ unsigned long long s_z[BIN * BIN];   // the bin's depth buffer: zpack(depth, triangle), Z_EMPTY = nothing yet
    __shared__ EdgeRec s_tri[BIGB];     // tile path: edge equations of the current round
    __shared__ int s_big[BATCH];        // tile path: triangles of the current batch waiting for a round
    __shared__ int s_nbig;
    __shared__ int s_clist[256];        // live chunks of the current segment (ascending)
    __shared__ unsigned long long s_mask[2][2][NTILES][BIGB / 64];   // [round parity][0 = 32-bit class, 1 = the rest]
    __shared__ int s_list[(SCAN_K + 1) * BATCH];   // pending triangle indices (any order), consumed from the top
    __shared__ int s_pending, s_nlive;
    __shared__ int s_wtot[2][4];        // hits each wave appended in a scan iteration (double-buffered by iteration parity)

    const int bin_x0 = bxi * BIN, bin_y0 = byi * BIN;
    const int bin_x1 = min(bin_x0 + BIN, W) - 1, bin_y1 = min(bin_y0 + BIN, H) - 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lx = lane & 7, ly = lane >> 3;
    const size_t bin_lin = ((size_t)b * OY + byi) * OX + bxi;

    float best_d[TILES_PER_WAVE][4];
    int best_id[TILES_PER_WAVE][4];
#pragma unroll
    for (int k = 0; k < TILES_PER_WAVE; ++k)
#pragma unroll
        for (int q = 0; q < 4; ++q) { best_d[k][q] = 2.0f; best_id[k][q] = -1; }

    int total_hits = 0;   // block-uniform: triangles whose bounding box touches this bin
    // IDS: the bin's list count and this thread's entry of the list are requested HERE, beside the image box, not behind the depth
    // buffer's initialisation and its barrier: two hops less in the chain count -> entry -> record that every workgroup waits through.
    // The entry goes to s_list with the initialisation (it must not live in a register across the walk: 72 are what seven workgroups per
    // CU leave); entries at or beyond the count are never read, and the list holds BL_CAP = 256 slots per bin: the address is valid.
    static_assert(BL_CAP == BATCH, "one list entry per thread");
    const int cnt_early = (IDS && sh.bin_cnt) ? sh.bin_cnt[bin_lin] : -1;
    const int entry_early = (IDS && sh.bin_cnt) ? sh.bin_list[bin_lin * BL_CAP + tid] : 0;
    const ImgBox ib = ibox[b];
    const bool bin_live = !(ib.x1 < bin_x0 || ib.x0 > bin_x1 || ib.y1 < bin_y0 || ib.y0 > bin_y1);
    const bool sparse = (SHADE || IDS) && sh.occ != nullptr;
    if (SHADE && sparse && !QUEUE && b == 0 && bxi == 0 && byi == 0 && tid == 0) {   // (queue form: k_worklist writes it)
        const Taps tp0 = make_taps(0.0f, 0.0f, sh.Ht, sh.Wt, sh.C, sh.boundary);
        for (int c = 0; c < 4; ++c) sh.empty_out[c] = c < sh.C ? bilerp(sh.tex, tp0, c, sh.C) : 0.0f;
    }
    if (sparse && !bin_live) {   // nothing of this image near the bin: no pixel is written, the consumers skip it too
        if (tid == 0) sh.occ[bin_lin] = 0;
        return;
    }

    if (bin_live) {
        for (int k = tid; k < BIN * BIN; k += 256) s_z[k] = Z_EMPTY;
        for (int k = tid; k < 4 * NTILES * (BIGB / 64); k += 256) (&s_mask[0][0][0][0])[k] = 0ull;
        if (tid == 0) { s_nbig = 0; s_pending = 0; s_nlive = 0; }
        if (IDS && sh.bin_cnt) s_list[tid] = entry_early;
        __syncthreads();
        // record slots of the image: [0, T) one per triangle, [Tp, Tp + n_over) the second pieces of clipped triangles (k_setup); slot =
        // chunk * 256 + lane in both regions (Tp is chunk-aligned), and the overflow chunks, which carry no chunk box, are all scanned
        const int Tp = padded_slots(T), n_over = ib.n_over, slot_end = Tp + n_over;
        const TriBox *bx = boxes + (size_t)b * 2 * Tp;
        const TriRec *rc = recs + (size_t)b * 2 * Tp;
        const int n_prim = Tp / 256, n_chunks = n_prim + (n_over + 255) / 256;
        const TriBox *cbx = cboxes + (size_t)b * n_prim;
        const unsigned long long below = (1ull << lane) - 1ull;
        int pending = 0;  // block-uniform copy of s_pending: entries waiting in s_list
        int round_no = 0; // block-uniform: tile-path rounds done (parity selects the mask buffer)
        // The winner of a pixel is the minimum of (depth, triangle index), which does not depend on the order in which
        // triangles arrive: lists are filled with one LDS atomic per wave and consumed from the top, no ordered compaction.
        auto process_batch = [&](const int n) {
            // consumes entries [pending - n, pending) of s_list; a barrier has been passed since they were appended
            // ---- lane path: a thread rasterises one triangle of the batch over its bounding box; a batch of at most 128 (64)
            // triangles -- the usual bin of a face mesh holds ~120 -- is walked by 2 (4) threads per triangle, rows interleaved,
            // so that all four waves share the work ----
            const int split = n <= 64 ? 4 : (n <= 128 ? 2 : 1);
            if (tid < n * split) {
                const int part = (tid >= n) + (tid >= 2 * n) + (tid >= 3 * n);
                const int t = s_list[pending - n + (tid - part * n)];
                // IDS: the triangle's silhouette bits ride BELOW its index in the depth key -- (index << 3) | bits keeps the order of the
                // indices (rule R6) -- so that the read-out has them with the winner: one byte gather per (bin, triangle), in flight beside
                // the record's, instead of a dependent one per pixel at the workgroup's end
                unsigned int silb = IDS ? (unsigned int)ld32(sh.sil + (size_t)b * T, (unsigned int)(t < T ? t : 0)) : 0u;
                const TriRec r = ld32(rc, t);
                const TriBox q = ld32(bx, t);
                if (IDS && t >= T) silb = (unsigned int)ld32(sh.sil + (size_t)b * T, (unsigned int)r.tid);      // (second piece of a clipped triangle)
                const int zkey = IDS ? (int)(((unsigned int)r.tid << 3) | silb) : r.tid;
                const int ext_x = max(r.X0, max(r.X1, r.X2)) - min(r.X0, min(r.X1, r.X2));
                const int ext_y = max(r.Y0, max(r.Y1, r.Y2)) - min(r.Y0, min(r.Y1, r.Y2));
                const int x0 = max((int)q.x0, bin_x0), x1 = min((int)q.x1, bin_x1);
                const int y0 = max((int)q.y0, bin_y0), y1 = min((int)q.y1, bin_y1);
                const int bw = x1 - x0 + 1, area = bw * (y1 - y0 + 1);
                if (ext_x <= SMALL_EXTENT && ext_y <= SMALL_EXTENT && area <= LANE_MAX) {
                    // every sample lies inside the triangle's box: |P - anchor| <= 16384 + 128, |A|,|B| <= 16384, so the
                    // edge functions fit int32 and their products v_mad_i32_i24 (same integers as the other paths)
                    const int D = (r.X1 - r.X0) * (r.Y2 - r.Y0) - (r.Y1 - r.Y0) * (r.X2 - r.X0);
                    const int sg = D > 0 ? 1 : -1;
                    const int A0 = -(r.Y2 - r.Y1) * sg, B0 = (r.X2 - r.X1) * sg;
                    const int A1 = -(r.Y0 - r.Y2) * sg, B1 = (r.X0 - r.X2) * sg;
                    const int A2 = -(r.Y1 - r.Y0) * sg, B2 = (r.X1 - r.X0) * sg;
                    // R5 tie rule, folded into the constants: E' = E - 1 for an edge that does not own E == 0
                    const int n0 = ((-A0 > 0) || (A0 == 0 && B0 < 0)) ? 0 : 1;
                    const int n1 = ((-A1 > 0) || (A1 == 0 && B1 < 0)) ? 0 : 1;
                    const int n2 = ((-A2 > 0) || (A2 == 0 && B2 < 0)) ? 0 : 1;
                    const int Px = x0 * SUBPIX + HALFPIX, Py = (y0 + part) * SUBPIX + HALFPIX;
                    int R0 = __mul24(A0, Px - r.X1) + __mul24(B0, Py - r.Y1) - n0;   // edge functions at the row start
                    int R1 = __mul24(A1, Px - r.X2) + __mul24(B1, Py - r.Y2) - n1;
                    int R2 = __mul24(A2, Px - r.X0) + __mul24(B2, Py - r.Y0) - n2;
                    const int A0s = A0 * SUBPIX, A1s = A1 * SUBPIX, A2s = A2 * SUBPIX;
                    const int rx0 = Px - r.X0;
                    int ry = Py - r.Y0;
                    unsigned long long *zrow = &s_z[(y0 + part - bin_y0) * BIN + (x0 - bin_x0)];
                    const int bh = y1 - y0 + 1;
                    // rows outside, columns inside: the inner trip is three additions and one test (the flattened loop that
                    // this replaces spent two thirds of its instructions on wrap-around selects)
                    const int B0s = B0 * SUBPIX * split, B1s = B1 * SUBPIX * split, B2s = B2 * SUBPIX * split;
                    for (int rr = part; rr < bh; rr += split) {
                        int E0 = R0, E1 = R1, E2 = R2, rx = rx0;
                        const float dzr = __fmaf_rn(r.zB, (float)ry, r.z0);
                        for (int c = 0; c < bw; ++c) {
                            if ((E0 | E1 | E2) >= 0) {
                                const float d = __fmaf_rn(r.zA, (float)rx, dzr);
                                if (d >= -1.0f && d <= 1.0f) atomicMin(&zrow[c], zpack(d, zkey));
                            }
                            E0 += A0s; E1 += A1s; E2 += A2s; rx += SUBPIX;
                        }
                        R0 += B0s; R1 += B1s; R2 += B2s;
                        ry += SUBPIX * split;
                        zrow += BIN * split;
                    }
                } else if (part == 0) {
                    s_big[atomicAdd(&s_nbig, 1)] = t;
                }
            }
            __syncthreads();
            // ---- tile path, BIGB triangles per round (block-uniform loop; rare on the meshes this is built for) ----
            const int nbig = __builtin_amdgcn_readfirstlane(s_nbig);
            for (int base = 0; base < nbig; base += BIGB, ++round_no) {
                const int m = min(BIGB, nbig - base);
                unsigned long long (*mask)[NTILES][BIGB / 64] = s_mask[round_no & 1];
                bool any_large = false;
                if (tid < m) {
                    const int t = s_big[base + tid];
                    const TriRec r = ld32(rc, t);
                    const TriBox q = ld32(bx, t);
                    EdgeRec e;
                    e.id = IDS ? (int)(((unsigned int)r.tid << 3) | (unsigned int)ld32(sh.sil + (size_t)b * T, (unsigned int)r.tid)) : r.tid;      // (the depth key's low word)
                    e.zA = r.zA; e.zB = r.zB; e.z0 = r.z0;
                    e.X0 = r.X0; e.Y0 = r.Y0;
                    const int ext_x = max(r.X0, max(r.X1, r.X2)) - min(r.X0, min(r.X1, r.X2));
                    const int ext_y = max(r.Y0, max(r.Y1, r.Y2)) - min(r.Y0, min(r.Y1, r.Y2));
                    const bool small = ext_x <= SMALL_EXTENT && ext_y <= SMALL_EXTENT;
                    int nb = 0;
                    // edge 0: (1,2)  edge 1: (2,0)  edge 2: (0,1);  A = -(Yb - Ya) s, B = (Xb - Xa) s
                    if (small) {   // everything fits in int32
                        const int D = (r.X1 - r.X0) * (r.Y2 - r.Y0) - (r.Y1 - r.Y0) * (r.X2 - r.X0);
                        const int sg = D > 0 ? 1 : -1;
                        e.A0 = -(r.Y2 - r.Y1) * sg; e.B0 = (r.X2 - r.X1) * sg;
                        e.A1 = -(r.Y0 - r.Y2) * sg; e.B1 = (r.X0 - r.X2) * sg;
                        e.A2 = -(r.Y1 - r.Y0) * sg; e.B2 = (r.X1 - r.X0) * sg;
                        nb = 8;
                        e.v.X1 = r.X1; e.v.Y1 = r.Y1; e.v.X2 = r.X2; e.v.Y2 = r.Y2; e.v.X0 = r.X0; e.v.Y0 = r.Y0;
                    } else {
                        any_large = true;
                        const long long X0 = r.X0, Y0 = r.Y0, X1 = r.X1, Y1 = r.Y1, X2 = r.X2, Y2 = r.Y2;
                        const long long D = (X1 - X0) * (Y2 - Y0) - (Y1 - Y0) * (X2 - X0);
                        const long long sg = D > 0 ? 1 : -1;
                        e.A0 = (int32_t)(-(Y2 - Y1) * sg); e.B0 = (int32_t)((X2 - X1) * sg);
                        e.A1 = (int32_t)(-(Y0 - Y2) * sg); e.B1 = (int32_t)((X0 - X2) * sg);
                        e.A2 = (int32_t)(-(Y1 - Y0) * sg); e.B2 = (int32_t)((X1 - X0) * sg);
                    }
                    // R5 tie rule: an edge owns E == 0 iff its direction (dx,dy) = (B,-A) has dy > 0, or dy == 0 and dx < 0
                    nb |= ((-e.A0 > 0) || (e.A0 == 0 && e.B0 < 0)) ? 0 : 1;
                    nb |= ((-e.A1 > 0) || (e.A1 == 0 && e.B1 < 0)) ? 0 : 2;
                    nb |= ((-e.A2 > 0) || (e.A2 == 0 && e.B2 < 0)) ? 0 : 4;
                    if (!small) {
                        e.C[0] = -((long long)e.A0 * r.X1 + (long long)e.B0 * r.Y1) - (nb & 1);
                        e.C[1] = -((long long)e.A1 * r.X2 + (long long)e.B1 * r.Y2) - ((nb >> 1) & 1);
                        e.C[2] = -((long long)e.A2 * r.X0 + (long long)e.B2 * r.Y0) - ((nb >> 2) & 1);
                    }
                    e.nb = nb;
                    s_tri[tid] = e;
                    const int tx0 = (max((int)q.x0, bin_x0) - bin_x0) >> 4, tx1 = (min((int)q.x1, bin_x1) - bin_x0) >> 4;
                    const int ty0 = (max((int)q.y0, bin_y0) - bin_y0) >> 4, ty1 = (min((int)q.y1, bin_y1) - bin_y0) >> 4;
                    const unsigned long long bit = 1ull << (tid & 63);
                    for (int ty = ty0; ty <= ty1; ++ty)
                        for (int tx = tx0; tx <= tx1; ++tx) atomicOr(&mask[small ? 0 : 1][ty * TILES_X + tx][tid >> 6], bit);
                }
                const int large_round = __builtin_amdgcn_readfirstlane(__syncthreads_or(any_large ? 1 : 0));
                // fine raster: this wave's 16x16 tile, four pixels per lane, winners in registers
#pragma unroll
                for (int k = 0; k < TILES_PER_WAVE; ++k) {
                    const int tile = wave * TILES_PER_WAVE + k;
                    const int px = bin_x0 + (tile % TILES_X) * TILE + lx, py = bin_y0 + (tile / TILES_X) * TILE + ly;
                    fine_tile<true>(mask[0][tile], s_tri, px * SUBPIX + HALFPIX, py * SUBPIX + HALFPIX, best_d[k], best_id[k]);
                    if (large_round)
                        fine_tile<false>(mask[1][tile], s_tri, px * SUBPIX + HALFPIX, py * SUBPIX + HALFPIX, best_d[k], best_id[k]);
                }
                __syncthreads();
                // this parity's masks are next written two rounds from now, with a barrier in between
                for (int k = tid; k < 2 * NTILES * (BIGB / 64); k += 256) (&mask[0][0][0])[k] = 0ull;
            }
            if (tid == 0) { s_pending = pending - n; s_nbig = 0; }
            pending -= n;
            __syncthreads();
        };
        // ---- the bin's own triangle list (one count, one gather), unless it overflowed ----
        const int n_listed = (IDS && sh.bin_cnt) ? __builtin_amdgcn_readfirstlane(cnt_early) : -1;
        if (n_listed >= 0 && n_listed <= BL_CAP) {
            total_hits = n_listed;
            if (n_listed > 0) {      // (the list is in s_list[0 .. n_listed): one batch)
                pending = n_listed;
                process_batch(n_listed);
            }
        } else
        for (int seg = 0; seg < n_chunks; seg += 256) {
            // ---- which of the next 256 chunks touch this bin?  (one box test per chunk) ----
            {
                const int c = seg + tid;
                bool live = false;
                if (c < n_prim) {
                    const TriBox q = ld32(cbx, c);
                    live = (q.x0 <= q.x1) && !(q.x1 < bin_x0 || q.x0 > bin_x1 || q.y1 < bin_y0 || q.y0 > bin_y1);
                } else if (c < n_chunks) {
                    live = true;
                }
                const unsigned long long bal = __ballot(live);
                int base = 0;
                if (lane == 0 && bal) base = atomicAdd(&s_nlive, __popcll(bal));
                base = __builtin_amdgcn_readfirstlane(base);
                if (live) s_clist[base + __popcll(bal & below)] = c;
            }
            __syncthreads();
            const int n_live = __builtin_amdgcn_readfirstlane(s_nlive);
            // ---- scan the bounding boxes of SCAN_K live chunks per iteration (independent loads in flight) ----
            for (int ci = 0, it = 0; ci < n_live; ci += SCAN_K, ++it) {
                bool hit[SCAN_K];
                int tt[SCAN_K];
#pragma unroll
                for (int k = 0; k < SCAN_K; ++k) {
                    hit[k] = false;
                    tt[k] = (ci + k < n_live) ? s_clist[ci + k] * 256 + tid : slot_end;
                    if (tt[k] >= T && tt[k] < Tp) tt[k] = slot_end;      // (padding of the last triangle chunk)
                }
                TriBox qk[SCAN_K];
#pragma unroll
                for (int k = 0; k < SCAN_K; ++k) qk[k] = ld32(bx, min(tt[k], slot_end - 1));
                unsigned long long bal[SCAN_K];
                int total = 0;
#pragma unroll
                for (int k = 0; k < SCAN_K; ++k) {
                    const TriBox q = qk[k];
                    hit[k] = tt[k] < slot_end && (q.x0 <= q.x1) && !(q.x1 < bin_x0 || q.x0 > bin_x1 || q.y1 < bin_y0 || q.y0 > bin_y1);
                    bal[k] = __ballot(hit[k]);
                    total += __popcll(bal[k]);
                }
                int base = 0;
                if (lane == 0 && total) base = atomicAdd(&s_pending, total);
                if (lane == 0) s_wtot[it & 1][wave] = total;
                base = __builtin_amdgcn_readfirstlane(base);
#pragma unroll
                for (int k = 0; k < SCAN_K; ++k) {
                    if (hit[k]) s_list[base + __popcll(bal[k] & below)] = tt[k];
                    base += __popcll(bal[k]);
                }
                __syncthreads();
                // block-uniform by construction: the four waves' counts of THIS iteration (a fast wave may already be
                // adding the next iteration's hits to s_pending, so that counter must not be re-read here)
                const int added = __builtin_amdgcn_readfirstlane(s_wtot[it & 1][0] + s_wtot[it & 1][1] + s_wtot[it & 1][2] + s_wtot[it & 1][3]);
                pending += added;
                total_hits += added;
                while (pending >= BATCH) process_batch(BATCH);
            }
            if (seg + 256 < n_chunks) {   // another segment follows: recycle the chunk list
                __syncthreads();
                if (tid == 0) s_nlive = 0;
                __syncthreads();
            }
        }
        while (pending > 0) process_batch(min(pending, BATCH));
    }

    if (sparse) {
        if (tid == 0) sh.occ[bin_lin] = total_hits > 0 ? 1 : 0;
        if (total_hits == 0) return;
    }
    if (!SHADE && sh.op_hint && tid == 0) sh.op_hint[bin_lin] = total_hits > 0 ? 1 : 0;
    if (total_hits == 0) {
        // ---- nothing touches this bin (most bins of a dense call): stream the empty result, two full 512-byte rows of
        // rast per wave instruction instead of the shading loop's 8x8 quadrant pattern ----
        float ecol[4] = {0.f, 0.f, 0.f, 0.f};
        if (SHADE) {
            const Taps tp0 = make_taps(0.0f, 0.0f, sh.Ht, sh.Wt, sh.C, sh.boundary);
            for (int c = 0; c < min(sh.C, 4); ++c) ecol[c] = bilerp(sh.tex, tp0, c, sh.C);
        }
        for (int i = tid; i < BIN * BIN; i += 256) {
            const int px = bin_x0 + (i % BIN), py = bin_y0 + (i / BIN);
            if (px >= W || py >= H) continue;
            const size_t off = ((size_t)b * H + py) * W + px;
            rast[off] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (WRITE_DB) rast_db[off] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (SHADE) {
                if (sh.C <= 4) { for (int c = 0; c < sh.C; ++c) sh.color[off * sh.C + c] = ecol[c]; }
                else {
                    const Taps tp0 = make_taps(0.0f, 0.0f, sh.Ht, sh.Wt, sh.C, sh.boundary);
                    for (int c = 0; c < sh.C; ++c) sh.color[off * sh.C + c] = bilerp(sh.tex, tp0, c, sh.C);
                }
            }
        }
        return;
    }
    const int C = CS > 0 ? CS : sh.C, boundary = BMODE >= 0 ? BMODE : sh.boundary;
    // ---- fold the tile path's register winners into the depth buffer, then read every pixel's winner ----
    // From here on a thread owns the four pixels (tid & 31, (tid >> 5) + 8 k) of the bin: the 32 lanes of a half wave write
    // one 512-byte row segment of rast per instruction (the raster's 8 x 8 quadrant pattern gave 128-byte pieces).
    int win[4];
    __shared__ float s_fy[BIN];      // NDC y of the bin's 32 rows: one IEEE division per row instead of one per pixel
    if (tid < BIN) s_fy[tid] = (2.0f * (float)(bin_y0 + tid) + 1.0f) / (float)H - 1.0f;
    if (bin_live) {
#pragma unroll
        for (int k = 0; k < TILES_PER_WAVE; ++k) {
            const int tile = wave * TILES_PER_WAVE + k;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int zx = (tile % TILES_X) * TILE + (q & 1) * QUAD + lx, zy = (tile / TILES_X) * TILE + (q >> 1) * QUAD + ly;
                if (best_id[k][q] >= 0) atomicMin(&s_z[zy * BIN + zx], zpack(best_d[k][q], best_id[k][q]));
            }
        }
        __syncthreads();
    }
    if (IDS) {
        // one-pass objective: the plane of this bin's winners, 16 bytes per thread (four adjacent pixels of row tid >> 3); the
        // silhouette bits of a pixel's triangle ride above its id, so that the shading kernel classifies pixel pairs from ids alone
        const int r = tid >> 3, c4 = (tid & 7) * 4;
        unsigned int e[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned long long z = s_z[r * BIN + c4 + j];
            // (the tile path resolves whole 16 x 16 tiles: a winner beyond the image's right / top border is not a pixel)
            const unsigned int key = (unsigned int)z;      // (triangle << 3) | silhouette bits
            const bool none = z == Z_EMPTY || bin_x0 + c4 + j >= W || bin_y0 + r >= H;
            e[j] = none ? 0u : (((key & 7u) << 24) | ((key >> 3) + 1u));
        }
        reinterpret_cast<uint4 *>(sh.idp + bin_lin * (BIN * BIN))[tid] = make_uint4(e[0], e[1], e[2], e[3]);
        return;
    }
    const int zx = tid & 31, zy0 = tid >> 5;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const unsigned long long z = bin_live ? s_z[(zy0 + 8 * k) * BIN + zx] : Z_EMPTY;
        win[k] = z == Z_EMPTY ? -1 : (int)(unsigned int)z;
    }

    // ---- shade + write (every pixel of the bin is written exactly once) ----
    // One-channel colour and its loss gradient are 4 bytes per pixel: stores of 4 bytes per lane reach a quarter of the rate of
    // 16-byte ones, so both are staged in LDS (the raster phase's lists are dead by now) and leave as float4 at the end.
    static_assert(sizeof(s_list) >= BIN * BIN * sizeof(float) && sizeof(s_tri) >= BIN * BIN * sizeof(float), "staging aliases the raster lists");
    float *s_col = reinterpret_cast<float *>(s_list);
    float *s_gaa = reinterpret_cast<float *>(s_tri);
    const bool stage = SHADE && C == 1;
    float col0[4] = {0.f, 0.f, 0.f, 0.f};   // LOSS: channel 0 of this thread's pixels (re-reading a just-written line stalls)
    unsigned int any_sil = 0u;              // LOSS: silhouette bits of the triangles this thread's pixels show, OR-ed
    const float sx = 2.0f / (float)W, sy = 2.0f / (float)H;
    const float4 *p = pos + (size_t)b * V;
    Taps empty_tp = {};
    float empty_col[4] = {0.f, 0.f, 0.f, 0.f};
    if (SHADE && QUEUE && C <= 4) {
        // list form: the colour of an empty pixel was computed once for the call (k_list_count, block 0) -- four scalar loads
        // instead of a tap set and four texel loads in front of every bin's shading
        for (int c = 0; c < C; ++c) empty_col[c] = sh.empty_out[c];
    } else if (SHADE) {
        empty_tp = make_taps(0.0f, 0.0f, sh.Ht, sh.Wt, C, boundary);
        for (int c = 0; c < min(C, 4); ++c) empty_col[c] = bilerp(sh.tex, empty_tp, c, C);
    }
    const int px = bin_x0 + zx;
    // Every gather below goes through a 32-bit byte offset from a wave-uniform base (common.h ld32 / at32): per-image vertex, index
    // and silhouette arrays, and the bin's own pixels at bin_off + zy * W + zx.
    struct I3 { int a, b, c; };
    struct UV3 { float2 q0, q1, q2; };
    const size_t bin_off = ((size_t)b * H + bin_y0) * W + bin_x0;
    float4 *const rast_bin = rast + bin_off;
    const uint8_t *const sil_img = LOSS ? sh.sil + (size_t)b * T : nullptr;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int zy = zy0 + 8 * k, py = bin_y0 + zy;
        if (px >= W || py >= H) continue;
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f), d = make_float4(0.f, 0.f, 0.f, 0.f);
        const int t = win[k];
        if (t >= 0) {
            const I3 ti = ld32(reinterpret_cast<const I3 *>(tri), t);
            const float fx = (2.0f * (float)px + 1.0f) / (float)W - 1.0f;
            const float fy = s_fy[zy];
            Shade sd = shade_pixel(ld32(p, ti.a), ld32(p, ti.b), ld32(p, ti.c), fx, fy, sx, sy);
            o = make_float4(sd.u, sd.v, sd.zw, (float)(t + 1));
            d = make_float4(sd.dudx, sd.dudy, sd.dvdx, sd.dvdy);
        }
        const unsigned int poff = (unsigned int)(zy * W + zx);      // pixel offset inside the bin's window of the image
        const size_t off = bin_off + poff;
        at32(rast_bin, poff) = o;
        if (WRITE_DB) rast_db[off] = d;
        if (LOSS) {   // (z/w, id) of the bin's pixels for the neighbour tests below; this thread owns the entry
            // id + 1 in 24 bits (ids are exact in rast's float anyway), the triangle's silhouette bits above them
            const unsigned int sb = t >= 0 ? (unsigned int)ld32(sil_img, t) : 0u;
            any_sil |= sb;
            s_z[zy * BIN + zx] = ((unsigned long long)__float_as_uint(o.z) << 32) | (sb << 24) | (unsigned int)(t + 1);
        }
        if (SHADE) {
            // interpolate (reference fit.py:157) + texture 'linear' (fit.py:158), same arithmetic as the stand-alone
            // kernels; an empty pixel samples uv = (0,0) exactly as they do (one tap set, hoisted out of the loops)
            if (t >= 0) {
                float2 q0, q1, q2;
                if (sh.tri_uv) { const UV3 tq = ld32(reinterpret_cast<const UV3 *>(sh.tri_uv), t); q0 = tq.q0; q1 = tq.q1; q2 = tq.q2; }
                else uv_indirect(sh.uv, sh.uv_tri, t, q0, q1, q2);      // (out of line: merged with the branch above, its loads would drag
                                                                       //  64-bit address arithmetic into the common path)
                const float w = 1.0f - o.x - o.y;
                const float tu = o.x * q0.x + o.y * q1.x + w * q2.x;
                const float tv = o.x * q0.y + o.y * q1.y + w * q2.y;
                auto put = [&](int c, float v) {
                    if (stage) s_col[zy * BIN + zx] = v;
                    else sh.color[off * C + c] = v;
                    if (LOSS && c == 0) col0[k] = v;
                };
                if (MIP) {
                    // interpolate(..., rast_db, diff_attrs='all') + texture('linear-mipmap-linear') (fit.py:153-155), same arithmetic as
                    // the stand-alone kernels: the footprint of the texture coordinate from the barycentrics' screen derivatives
                    const float e0x = q0.x - q2.x, e0y = q0.y - q2.y, e1x = q1.x - q2.x, e1y = q1.y - q2.y;
                    const float4 da = make_float4(d.x * e0x + d.z * e1x, d.y * e0x + d.w * e1x, d.x * e0y + d.z * e1y, d.y * e0y + d.w * e1y);
                    TexLevels lv;
                    lv.tex[0] = sh.tex;
                    for (int l = 1; l <= FPCDR_MAX_MIP; ++l) lv.tex[l] = sh.mip[l - 1];
                    mip_sample_fwd(lv, 0, sh.n_levels, make_float2(tu, tv), true, da, 0.0f, sh.Ht, sh.Wt, C, true, boundary, put);
                } else {
                // ('zero' takes the general tap routine: its taps carry validity bits, which bilerp masks by)
                const Taps tp = boundary == FPCDR_BOUNDARY_ZERO ? make_taps(tu, tv, sh.Ht, sh.Wt, C, boundary)
                                                                : make_taps_fast(tu, tv, sh.Ht, sh.Wt, C, boundary);
                for (int c = 0; c < C; ++c) put(c, bilerp<true>(sh.tex, tp, c, C));      // (the fused entry points require < 2^30 texel values)
                }
            } else if (stage) {
                s_col[zy * BIN + zx] = empty_col[0];
            } else {
                for (int c = 0; c < C; ++c) sh.color[off * C + c] = c < 4 ? empty_col[c] : bilerp(sh.tex, empty_tp, c, C);
            }
        }
    }
    // the staged planes leave as 16-byte stores: thread -> four adjacent pixels of row tid >> 3
    auto flush_plane = [&](const float *s_plane, float *__restrict__ dst) {
        const int r = tid >> 3, c4 = (tid & 7) * 4;
        const int fx0 = bin_x0 + c4, fy = bin_y0 + r;
        if (fy >= H || fx0 >= W) return;
        float *const dst_bin = dst + bin_off;      // (uniform)
        const unsigned int poff = (unsigned int)(r * W + c4);
        const float4 v = *reinterpret_cast<const float4 *>(s_plane + r * BIN + c4);
        if ((W & 3) == 0 && (((size_t)dst) & 15) == 0) *reinterpret_cast<float4 *>(&at32(dst_bin, poff)) = v;
        else {
            const float e[4] = {v.x, v.y, v.z, v.w};
            for (int j = 0; j < 4; ++j)
                if (fx0 + j < W) at32(dst_bin, poff + j) = e[j];
        }
    };
    if (SHADE && !LOSS) {
        if (stage) {     // (uniform)
            __syncthreads();
            flush_plane(s_col, sh.color);
        }
    }
    if (LOSS) {
        // ---- every pixel gets the loss term and gradient of its UN-antialiased colour; CANDIDATES are marked ----
        // A pixel's antialiased colour differs from its colour only if one of its four pixel pairs has different ids AND
        // the nearer triangle of the pair has a silhouette edge (the early-outs of for_active_edges, aa_pairs.h).  The
        // pairs inside the bin are classified here from the ids in LDS; for the pairs across the bin's border the four
        // border lines (z/w, id) go to a compact edge buffer.  k_aa_fix (fused.hip) classifies those, runs the full
        // antialias on the candidates only and corrects their loss / gradient: no dense antialias pass over the image.
        __shared__ unsigned int s_cmask[BIN];
        __shared__ float s_lpart[4];
        if (tid < BIN) s_cmask[tid] = 0u;
        // (barrier: (z/w, id) entries, the staged colour and the cleared masks are visible.)  A pair inside the bin can only be a
        // candidate if one of its two triangles owns a silhouette edge: most bins of a face show none at all -- the interior of the
        // mesh -- and skip the neighbour tests of their 1024 pixels altogether (the pairs across the border are k_aa_fix's)
        const bool bin_has_sil = __builtin_amdgcn_readfirstlane(__syncthreads_or(any_sil != 0u ? 1 : 0)) != 0;
        const size_t bin_id = bin_lin;
        unsigned long long *edge = sh.edges + bin_id * (4 * BIN);   // [left col | right col | bottom row | top row][32]
        if (tid < 4 * BIN) {      // the four border lines: one entry per thread (pixels beyond the image export 0)
            const int side = tid >> 5, i = tid & 31;
            const int ex = side == 0 ? 0 : (side == 1 ? BIN - 1 : i), ey = side == 2 ? 0 : (side == 3 ? BIN - 1 : i);
            edge[tid] = (bin_x0 + ex < W && bin_y0 + ey < H) ? s_z[ey * BIN + ex] : 0ull;
        }
        float lsum = 0.0f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int zy = zy0 + 8 * k, py = bin_y0 + zy;
            const bool inimg = px < W && py < H;
            const int idx = zy * BIN + zx;
            const unsigned long long me = inimg ? s_z[idx] : 0ull;
            if (!inimg) continue;
            const int id = (int)((unsigned int)me & 0xffffffu);
            const float z = __uint_as_float((unsigned int)(me >> 32));
            // pair_select keeps the FIRST pixel's triangle on a depth tie: right / upper pairs are (me, n), left / lower
            // pairs (n, me), as for_active_edges is called; the chosen triangle's silhouette bits ride in the entries.
            // First the cheap part for all four pairs: ids differ AND one of the two triangles owns a silhouette edge at all
            // (interior triangles of a closed mesh own none, so most id discontinuities end here); only such pairs decide
            // which of the two triangles is analysed.
            bool cand = false;
            if (bin_has_sil) {      // (uniform)
                const bool hR = zx < BIN - 1 && px + 1 < W, hU = zy < BIN - 1 && py + 1 < H, hL = zx > 0, hD = zy > 0;
                const unsigned long long nR = hR ? s_z[idx + 1] : me, nU = hU ? s_z[idx + BIN] : me;
                const unsigned long long nL = hL ? s_z[idx - 1] : me, nD = hD ? s_z[idx - BIN] : me;
                auto maybe = [&](unsigned long long n) {
                    return ((unsigned int)n & 0xffffffu) != (unsigned int)id && ((((unsigned int)n | (unsigned int)me) >> 24) & 0xffu) != 0;
                };
                if (((int)maybe(nR) | (int)maybe(nU) | (int)maybe(nL) | (int)maybe(nD))) {
                    auto pair = [&](unsigned long long n, bool me_first) {
                        const int nid = (int)((unsigned int)n & 0xffffffu);
                        if (nid == id) return false;
                        const float nz = __uint_as_float((unsigned int)(n >> 32));
                        const PairSel ps = me_first ? pair_select(id, z, nid, nz, T) : pair_select(nid, nz, id, z, T);
                        const bool takes_n = me_first ? ps.use1 : !ps.use1;
                        return ps.tau >= 0 && (((unsigned int)(takes_n ? n : me) >> 24) & 0xffu) != 0;
                    };
                    cand = (hR && pair(nR, true)) || (hU && pair(nU, true)) || (hL && pair(nL, false)) || (hD && pair(nD, false));
                }
            }
            if (cand) atomicOr(&s_cmask[zy], 1u << zx);
            const unsigned int poff = (unsigned int)(zy * W + zx);
            const size_t off = bin_off + poff;
            if (id > 0) {
                const float rf = (float)ld32(sh.ref + bin_off, poff);
                const float d0 = rf - sh.bg * sh.color_scale;
                for (int c = 0; c < C; ++c) {
                    const float cv = c == 0 ? col0[k] : sh.color[off * C + c];   // this thread wrote it above
                    const float dd = rf - cv * sh.color_scale;
                    lsum += dd * dd - d0 * d0;
                    const float gq = (-2.0f * sh.color_scale * sh.grad_scale) * dd;
                    if (stage) s_gaa[idx] = gq;
                    else sh.g_aa[off * C + c] = gq;
                }
            } else if (stage) {
                s_gaa[idx] = 0.0f;
            } else {
                for (int c = 0; c < C; ++c) sh.g_aa[off * C + c] = 0.0f;
            }
        }
        lsum = wave_sum_dpp(lsum);
        if (lane == 0) s_lpart[wave] = lsum;
        __syncthreads();
        if (stage) {
            flush_plane(s_col, sh.color);
            flush_plane(s_gaa, sh.g_aa);
        }
        if (tid < BIN) sh.cmask[bin_id * BIN + tid] = s_cmask[tid];
        if (tid == 0) {
            const double tot = (double)s_lpart[0] + (double)s_lpart[1] + (double)s_lpart[2] + (double)s_lpart[3];
            const unsigned int slot = ((unsigned int)bxi + 31u * (unsigned int)byi + 977u * (unsigned int)b) % FPCDR_LOSS_SLOTS;
            if (tot != 0.0) atomicAdd(sh.loss_sum + slot, tot);
        }
    }
}

// grid form: one workgroup per bin, grid (OX, OY, B)
template <bool WRITE_DB, bool SHADE, bool LOSS = false>
__global__ void __launch_bounds__(256) FPCDR_BINS_WPE k_bins(const float4 *__restrict__ pos, const int32_t *__restrict__ tri,
                                              int V, int T, int H, int W, const TriRec *__restrict__ recs,
                                              const TriBox *__restrict__ boxes, const TriBox *__restrict__ cboxes,
                                              const ImgBox *__restrict__ ibox, float4 *__restrict__ rast,
                                              float4 *__restrict__ rast_db, ShadeArgs sh) {
    bins_body<WRITE_DB, SHADE, LOSS, false>(blockIdx.z, blockIdx.x, blockIdx.y, gridDim.x, gridDim.y, pos, tri, V, T, H, W, recs, boxes, cboxes,
                                     ibox, rast, rast_db, sh);
}

// list form (sparse objective): one workgroup per entry of the list k_worklist built (the bins some chunk box touches,
// ~20 % of a face-rig frame).  Dispatching one workgroup per bin of the whole batch costs 0.21 ms for 588 k workgroups
// that mostly leave at once; the caller sizes this launch from the count of an EARLIER call (`cap`), and whatever lies
// beyond it is swept up by the strided form below, so the result never depends on the hint.
#define FPCDR_BINSQ_WPE
template <bool WRITE_DB, bool SHADE, bool LOSS, int CS = 0, int BMODE = -1, bool MIP = false, bool IDS = false>
__global__ void __launch_bounds__(256) FPCDR_BINS_WPE k_bins_list(const int32_t *__restrict__ list, const int32_t *__restrict__ count,
                                              int cap, int OX, int OY, fpcdr_bin_decode dc,
                                              const float4 *__restrict__ pos, const int32_t *__restrict__ tri,
                                              int V, int T, int H, int W, const TriRec *__restrict__ recs,
                                              const TriBox *__restrict__ boxes, const TriBox *__restrict__ cboxes,
                                              const ImgBox *__restrict__ ibox, float4 *__restrict__ rast,
                                              float4 *__restrict__ rast_db, ShadeArgs sh) {
    const int item = fpcdr_list_item(*count, cap);      // (XCD x takes the x-th eighth of the entries: common.h)
    if (item < 0) return;
    const int lin = __builtin_amdgcn_readfirstlane(list[item]);
    int b, byi, bxi;
    fpcdr_decode_bin(lin, dc, b, byi, bxi);
    bins_body<WRITE_DB, SHADE, LOSS, true, CS, BMODE, MIP, IDS>(b, bxi, byi, OX, OY, pos, tri, V, T, H, W, recs, boxes, cboxes, ibox, rast, rast_db, sh);
}

// strided form: entries first, first + gridDim.x, ... of the list.  The loop variable is scalar by construction, so the loop
// and the body's barriers stay uniform for the compiler.  (A dynamic pop -- thread 0's atomic handed round through LDS
// between two barriers -- made LLVM's structurizer wrap the barrier pair in a second loop level, and waves repeated
// barriers out of step; and the body inlined into a loop runs ~25 % slower than stand-alone -- 150 spilled SGPRs --, which
// is why this form only sweeps up what the hinted launch above did not reach.)
template <bool WRITE_DB, bool SHADE, bool LOSS, bool MIP = false, bool IDS = false>
__global__ void __launch_bounds__(256) FPCDR_BINSQ_WPE k_bins_queue(const int32_t *__restrict__ list, const int32_t *__restrict__ count,
                                              int first, int OX, int OY, fpcdr_bin_decode dc,
                                              const float4 *__restrict__ pos, const int32_t *__restrict__ tri,
                                              int V, int T, int H, int W, const TriRec *__restrict__ recs,
                                              const TriBox *__restrict__ boxes, const TriBox *__restrict__ cboxes,
                                              const ImgBox *__restrict__ ibox, float4 *__restrict__ rast,
                                              float4 *__restrict__ rast_db, ShadeArgs sh) {
    const int n = *count;
    for (int item = first + blockIdx.x; item < n; item += gridDim.x) {
        const int lin = __builtin_amdgcn_readfirstlane(list[item]);
        int b, byi, bxi;
        fpcdr_decode_bin(lin, dc, b, byi, bxi);
        bins_body<WRITE_DB, SHADE, LOSS, true, 0, -1, MIP, IDS>(b, bxi, byi, OX, OY, pos, tri, V, T, H, W, recs, boxes, cboxes, ibox, rast, rast_db, sh);
        __syncthreads();     // the next bin's first LDS writes must not overtake this bin's last LDS reads
    }
}

// per-bin window masks of the occupancy map (include/fpcdr.h, fpcdr_render_fwd_params.occ)
__global__ void __launch_bounds__(256) k_occ_window(const uint8_t *__restrict__ raw, int B, int OY, int OX, uint16_t *__restrict__ win) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)B * OY * OX) return;
    win[i] = (uint16_t)window_mask(raw, i, OY, OX);
}

// region hint, plane 1: plane 0 OR-ed over the 3 x 3 neighbourhood
__global__ void __launch_bounds__(256) k_hint_dilate(const uint8_t *__restrict__ raw, int B, int OY, int OX, uint8_t *__restrict__ out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)B * OY * OX) return;
    const int x = (int)(i % OX), y = (int)((i / OX) % OY);
    const uint8_t *img = raw + (i - (long long)y * OX - x);
    uint8_t m = 0;
    for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
            const int cx = x + dx, cy = y + dy;
            if (cx >= 0 && cx < OX && cy >= 0 && cy < OY) m |= img[cy * OX + cx];
        }
    out[i] = m ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------
template <bool HAS_DDB>
__global__ void __launch_bounds__(256) k_grad(const float4 *__restrict__ pos, const int32_t *__restrict__ tri,
                                              const float4 *__restrict__ rast, const float4 *__restrict__ dy,
                                              const float4 *__restrict__ ddb, int B, int V, int T, int H, int W,
                                              float *__restrict__ grad_pos, const uint8_t *__restrict__ hint) {
    // One workgroup per 32 x 32-pixel bin, four pixels per thread.  A wave pass covers two adjacent 32-pixel rows, the second
    // one right to left, so that the pixels of one triangle sit next to each other in lane order: ONE segmented scan sums the
    // nine gradient components of every run of equal triangle (common.h wave_segment_reduce9), the run tails add into the
    // workgroup's LDS vertex table, which is flushed once.  (The per-vertex loop with shuffle butterflies this replaces made
    // the kernel LDS-bound at 3.9 ms for cfg3; nine global atomics per run tail instead of the table: 7.4 ms.)
    if (hint && !fpcdr_hint_on(hint, 0, B, H, W, blockIdx.z, blockIdx.y * 32, blockIdx.x * 32)) return;   // empty bin: nothing to read
    __shared__ int s_vkey[FPCDR_VT_SLOTS];
        __shared__ double s_vacc[FPCDR_VT_SLOTS][3];
    const VTable vt = {s_vkey, s_vacc};
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = (lane & 32) ? 63 - lane : lane;
    const int px = blockIdx.x * 32 + col;
    const int b = blockIdx.z;
    // pass 1: which of this thread's four pixels carry a gradient (most bins of an image: none at all)
    float4 rr[4], gg[4], gdd[4];
    bool any = false;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int py = blockIdx.y * 32 + wave * 8 + 2 * k + (lane >> 5);
        rr[k] = make_float4(0.f, 0.f, 0.f, 0.f); gg[k] = rr[k]; gdd[k] = rr[k];
        if (px < W && py < H) {
            const size_t off = ((size_t)b * H + py) * W + px;
            rr[k] = rast[off];
            const int t = (int)rr[k].w - 1;
            if (t >= 0 && t < T) {
                gg[k] = dy[off];
                if (HAS_DDB) gdd[k] = ddb[off];
                any |= gg[k].x != 0.f || gg[k].y != 0.f || gdd[k].x != 0.f || gdd[k].y != 0.f || gdd[k].z != 0.f || gdd[k].w != 0.f;
            } else {
                rr[k].w = 0.f;
            }
        }
    }
    if (!__builtin_amdgcn_readfirstlane(__syncthreads_or(any ? 1 : 0))) return;
    vtable_init(vt, tid, 256);
    __syncthreads();
    float *gp = grad_pos + (size_t)b * V * 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int py = blockIdx.y * 32 + wave * 8 + 2 * k + (lane >> 5);
        int tkey = -1;
        int vk[3] = {0, 0, 0};
        float gv9[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        const float4 g = gg[k], gd = gdd[k];
        if (rr[k].w > 0.f && (g.x != 0.f || g.y != 0.f || gd.x != 0.f || gd.y != 0.f || gd.z != 0.f || gd.w != 0.f)) {
            const int t = (int)rr[k].w - 1;
            vk[0] = tri[3 * t]; vk[1] = tri[3 * t + 1]; vk[2] = tri[3 * t + 2];
            const float4 *p = pos + (size_t)b * V;
            const float fx = (2.0f * (float)px + 1.0f) / (float)W - 1.0f;
            const float fy = (2.0f * (float)py + 1.0f) / (float)H - 1.0f;
            float g0[3], g1[3], g2[3];
            shade_pixel_bwd<HAS_DDB>(p[vk[0]], p[vk[1]], p[vk[2]], fx, fy, 2.0f / (float)W, 2.0f / (float)H, g, gd, g0, g1, g2);
            gv9[0] = g0[0]; gv9[1] = g0[1]; gv9[2] = g0[2];
            gv9[3] = g1[0]; gv9[4] = g1[1]; gv9[5] = g1[2];
            gv9[6] = g2[0]; gv9[7] = g2[1]; gv9[8] = g2[2];
            tkey = t;
        }
        wave_segment_reduce9(tkey, gv9, [&](int, const float (&sm)[9]) { vtable_add(vt, gp, vk, sm); });
    }
    __syncthreads();
    vtable_flush(vt, gp, tid, 256);
}

#if FPCDR_TWOCALL
// ---------------------------------------------------------------------------------------------
// Fused backward of texture('linear') -> interpolate -> rasterize (reference fit.py:158,157,151) for the render
// path: reads dL/d colour (4C B/px) and rast (16 B/px), writes nothing dense.  A covered pixel with a non-zero
// gradient re-derives its texture coordinate, scatters into grad_tex (4C atomics), chains d colour / d uv through
// the barycentrics to the three clip-space vertices and scatters into grad_pos (pre-reduced per wave by vertex).
__global__ void __launch_bounds__(256) k_render_bwd(const float4 *__restrict__ pos, const int32_t *__restrict__ tri,
                                                    const float2 *__restrict__ uv, const int32_t *__restrict__ uv_tri,
                                                    const float *__restrict__ tex, const float4 *__restrict__ rast,
                                                    const float *__restrict__ dy, int V, int T, int H, int W, int Ht, int Wt,
                                                    int C, int boundary, float *__restrict__ grad_pos,
                                                    float *__restrict__ grad_tex, const float2 *__restrict__ tri_uv) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int px = blockIdx.x * 16 + (wave & 1) * 8 + (lane & 7);
    const int py = blockIdx.y * 16 + (wave >> 1) * 8 + (lane >> 3);
    const int b = blockIdx.z;
    int key0 = -1, key1 = -1, key2 = -1;
    float g0[3] = {0, 0, 0}, g1[3] = {0, 0, 0}, g2[3] = {0, 0, 0};
    if (px < W && py < H) {
        const size_t off = ((size_t)b * H + py) * W + px;
        const float *g = dy + off * C;
        bool any = false;
        for (int c = 0; c < C; ++c) any |= (g[c] != 0.0f);
        if (any) {
            const float4 r = rast[off];
            int t = (int)r.w - 1;
            if (t >= T) t = -1;
            // texture backward; an empty pixel sampled uv = (0,0) in the forward pass and scatters there
            float2 q0 = make_float2(0.f, 0.f), q1 = q0, q2 = q0;
            float tu = 0.0f, tv = 0.0f;
            if (t >= 0) {
                if (tri_uv) { q0 = tri_uv[3 * t]; q1 = tri_uv[3 * t + 1]; q2 = tri_uv[3 * t + 2]; }
                else { q0 = uv[uv_tri[3 * t]]; q1 = uv[uv_tri[3 * t + 1]]; q2 = uv[uv_tri[3 * t + 2]]; }
                const float w = 1.0f - r.x - r.y;
                tu = r.x * q0.x + r.y * q1.x + w * q2.x;
                tv = r.x * q0.y + r.y * q1.y + w * q2.y;
            }
            const Taps tp = make_taps(tu, tv, Ht, Wt, C, boundary);
            const float w00 = (1.0f - tp.fx) * (1.0f - tp.fy), w10 = tp.fx * (1.0f - tp.fy);
            const float w01 = (1.0f - tp.fx) * tp.fy, w11 = tp.fx * tp.fy;
            float gfx = 0.f, gfy = 0.f;
            for (int c = 0; c < C; ++c) {
                const float gc = g[c];
                float t00, t10, t01, t11;
                load_taps(tex, tp, c, C, t00, t10, t01, t11);
                mask_taps(tp, t00, t10, t01, t11);
                gfx += gc * ((t10 - t00) * (1.0f - tp.fy) + (t11 - t01) * tp.fy);
                gfy += gc * ((t01 + (t11 - t01) * tp.fx) - (t00 + (t10 - t00) * tp.fx));
                if (grad_tex && gc != 0.0f) {      // (boundary mode 'zero': the padding receives no gradient)
                    if (tp.valid & 1u) atomicAdd(grad_tex + tp.i00 + c, gc * w00);
                    if (tp.valid & 2u) atomicAdd(grad_tex + tp.i10 + c, gc * w10);
                    if (tp.valid & 4u) atomicAdd(grad_tex + tp.i01 + c, gc * w01);
                    if (tp.valid & 8u) atomicAdd(grad_tex + tp.i11 + c, gc * w11);
                }
            }
            if (t >= 0 && grad_pos) {
                const float mu = (boundary == FPCDR_BOUNDARY_CLAMP && !(tu >= 0.0f && tu <= 1.0f)) ? 0.0f : 1.0f;
                const float mv = (boundary == FPCDR_BOUNDARY_CLAMP && !(tv >= 0.0f && tv <= 1.0f)) ? 0.0f : 1.0f;
                const float gtu = gfx * (float)Wt * mu, gtv = gfy * (float)Ht * mv;
                // interpolate backward: d uv / d (u, v)
                const float gu = gtu * (q0.x - q2.x) + gtv * (q0.y - q2.y);
                const float gv = gtu * (q1.x - q2.x) + gtv * (q1.y - q2.y);
                if (gu != 0.0f || gv != 0.0f) {
                    const int i0 = tri[3 * t], i1 = tri[3 * t + 1], i2 = tri[3 * t + 2];
                    const float4 *p = pos + (size_t)b * V;
                    const float fx = (2.0f * (float)px + 1.0f) / (float)W - 1.0f;
                    const float fy = (2.0f * (float)py + 1.0f) / (float)H - 1.0f;
                    shade_pixel_bwd<false>(p[i0], p[i1], p[i2], fx, fy, 2.0f / (float)W, 2.0f / (float)H,
                                           make_float4(gu, gv, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f), g0, g1, g2);
                    key0 = i0; key1 = i1; key2 = i2;
                }
            }
        }
    }
    if (!grad_pos || __ballot(key0 >= 0) == 0ull) return;
    float *gp = grad_pos + (size_t)b * V * 4;
    {
        float *const d[3] = {gp + 4 * (size_t)max(key0, 0), gp + 4 * (size_t)max(key0, 0) + 1, gp + 4 * (size_t)max(key0, 0) + 3};
        wave_group_atomic_add<3>(key0, d, g0);
    }
    {
        float *const d[3] = {gp + 4 * (size_t)max(key1, 0), gp + 4 * (size_t)max(key1, 0) + 1, gp + 4 * (size_t)max(key1, 0) + 3};
        wave_group_atomic_add<3>(key1, d, g1);
    }
    {
        float *const d[3] = {gp + 4 * (size_t)max(key2, 0), gp + 4 * (size_t)max(key2, 0) + 1, gp + 4 * (size_t)max(key2, 0) + 3};
        wave_group_atomic_add<3>(key2, d, g2);
    }
}
#endif  // FPCDR_TWOCALL

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// layout of the caller's scratch buffer: records and boxes (2 Tp slots per image: triangles + overflow), chunk boxes, image boxes, clip lists
struct RasterScratch { TriRec *recs; TriBox *boxes, *cboxes; ImgBox *ibox; int32_t *clip; size_t bytes; };
static RasterScratch raster_scratch(void *base, int B, int T) {
    const size_t n = (size_t)B * 2 * (size_t)padded_slots(T), nc = (size_t)B * (size_t)(padded_slots(T) / 256);
    char *s = (char *)base;
    RasterScratch r;
    r.recs = (TriRec *)s;
    r.boxes = (TriBox *)(s + align_up(n * sizeof(TriRec), 256));
    r.cboxes = (TriBox *)((char *)r.boxes + align_up(n * sizeof(TriBox), 256));
    r.ibox = (ImgBox *)((char *)r.cboxes + align_up(nc * sizeof(TriBox), 256));
    r.clip = (int32_t *)((char *)r.ibox + align_up((size_t)B * sizeof(ImgBox), 256));      // [B][Tp] clip lists (k_setup -> k_setup_clip)
    r.bytes = (size_t)((char *)r.clip - s) + align_up((size_t)B * (size_t)padded_slots(T) * sizeof(int32_t), 256);
    return r;
}

}  // namespace

extern "C" size_t fpcdr_rasterize_scratch_bytes(int32_t B, int32_t T) {
    if (B <= 0 || T <= 0) return 0;
    return raster_scratch(nullptr, B, T).bytes;
}

extern "C" int fpcdr_rasterize_fwd(const fpcdr_rasterize_fwd_params *p, void *stream) {
    FPCDR_REQUIRE(p != nullptr, "null params");
    FPCDR_REQUIRE(p->pos && p->tri && p->scratch && p->rast, "null pointer");
    FPCDR_REQUIRE(p->B > 0 && p->V > 0 && p->T > 0 && p->H > 0 && p->W > 0, "sizes must be positive");
    FPCDR_REQUIRE(p->H <= 32767 && p->W <= 32767, "resolution above 32767 is not supported");
    FPCDR_REQUIRE(p->B <= 65535, "more than 65535 images per call");
    FPCDR_REQUIRE(p->T < (1 << 24), "more than 2^24 triangles (rast stores triangle index + 1 as a float)");
    hipStream_t st = (hipStream_t)stream;
    const RasterScratch rs = raster_scratch(p->scratch, p->B, p->T);
    TriRec *recs = rs.recs;
    TriBox *boxes = rs.boxes, *cboxes = rs.cboxes;
    ImgBox *ibox = rs.ibox;
    hipLaunchKernelGGL(k_init_ibox, dim3(fpcdr_cdiv(p->B, 256)), dim3(256), 0, st, ibox, p->B);
    dim3 grid(fpcdr_cdiv(p->W, BIN), fpcdr_cdiv(p->H, BIN), p->B);
    const size_t nbins = (size_t)p->B * grid.y * grid.x;
    // (Tried: the empty result of the bins no chunk box touches streamed by a row-wise fill kernel, k_bins on the rest -- 1.5 ms of
    // fill + 1.3 ms for the occupied fifth of the bins, one after the other, against 2.33 ms here: in ONE kernel the memory-bound
    // empty bins overlap the compute-bound occupied ones.  r3: a stride permutation of each image's bins in dispatch order, so that
    // occupied and empty bins are in flight together at every moment, changes nothing -- 2.25 ms for strides 0 / n/55 / n/7 / n/1.6:
    // the long-lived occupied workgroups pile up on the CUs by themselves and the fill shares their 7 slots per CU.)
    hipLaunchKernelGGL(k_setup<false>, dim3(fpcdr_cdiv(p->T, 256), p->B), dim3(256), 0, st, (const float4 *)p->pos, p->tri,
                       p->B, p->V, p->T, p->H, p->W, recs, boxes, cboxes, ibox, (uint8_t *)nullptr, p->ranges, (int32_t *)nullptr, (int32_t *)nullptr,
                       (const int32_t *)nullptr, (uint8_t *)nullptr, rs.clip);
    hipLaunchKernelGGL(k_setup_clip<false>, dim3(p->B), dim3(256), 0, st, (const float4 *)p->pos, p->tri, p->V, p->T, p->H, p->W, recs, boxes, cboxes,
                       ibox, (uint8_t *)nullptr, (int32_t *)nullptr, (int32_t *)nullptr, rs.clip);
    ShadeArgs sh = {};
    sh.op_hint = p->hint;
    if (p->rast_db)
        hipLaunchKernelGGL((k_bins<true, false>), grid, dim3(256), 0, st, (const float4 *)p->pos, p->tri, p->V, p->T, p->H, p->W,
                           recs, boxes, cboxes, ibox, (float4 *)p->rast, (float4 *)p->rast_db, sh);
    else
        hipLaunchKernelGGL((k_bins<false, false>), grid, dim3(256), 0, st, (const float4 *)p->pos, p->tri, p->V, p->T, p->H, p->W,
                           recs, boxes, cboxes, ibox, (float4 *)p->rast, (float4 *)nullptr, sh);
    if (p->hint)
        hipLaunchKernelGGL(k_hint_dilate, dim3(fpcdr_cdiv((long long)nbins, 256)), dim3(256), 0, st, p->hint, p->B, (int)grid.y, (int)grid.x,
                           p->hint + nbins);
    FPCDR_CHECK_LAUNCH();
    return FPCDR_OK;
}

extern "C" int fpcdr_rasterize_bwd(const fpcdr_rasterize_bwd_params *p, void *stream) {
    FPCDR_REQUIRE(p != nullptr, "null params");
    FPCDR_REQUIRE(p->pos && p->tri && p->rast && p->dy && p->grad_pos, "null pointer");
    FPCDR_REQUIRE(p->B > 0 && p->V > 0 && p->T > 0 && p->H > 0 && p->W > 0, "sizes must be positive");
    FPCDR_REQUIRE(p->B <= 65535, "more than 65535 images per call");
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(fpcdr_cdiv(p->W, 32), fpcdr_cdiv(p->H, 32), p->B);
    if (p->ddb)
        hipLaunchKernelGGL(k_grad<true>, grid, dim3(256), 0, st, (const float4 *)p->pos, p->tri, (const float4 *)p->rast,
                           (const float4 *)p->dy, (const float4 *)p->ddb, p->B, p->V, p->T, p->H, p->W, p->grad_pos, p->hint);
    else
        hipLaunchKernelGGL(k_grad<false>, grid, dim3(256), 0, st, (const float4 *)p->pos, p->tri, (const float4 *)p->rast,
                           (const float4 *)p->dy, (const float4 *)nullptr, p->B, p->V, p->T, p->H, p->W, p->grad_pos, p->hint);
    FPCDR_CHECK_LAUNCH();
    return FPCDR_OK;
}

// ---- the two-call form of the pixel objective and the fused render pair (include/fpcdr_twocall.h): superseded by fpcdr_objective_fwd in the
// fit loop, kept as a parity partner and for the dense mode; compiled into libfpcdr_twocall.so only (csrc/Makefile) ----
#if FPCDR_TWOCALL
extern "C" int fpcdr_render_fwd(const fpcdr_render_fwd_params *p, void *stream) {
    FPCDR_REQUIRE(p != nullptr, "null params");
    FPCDR_REQUIRE(p->pos && p->tri && p->scratch && p->rast && p->uv && p->uv_tri && p->tex && p->color, "null pointer");
    FPCDR_REQUIRE(p->B > 0 && p->V > 0 && p->T > 0 && p->H > 0 && p->W > 0 && p->Ht > 0 && p->Wt > 0 && p->C > 0,
                  "sizes must be positive");
    FPCDR_REQUIRE(p->H <= 32767 && p->W <= 32767 && p->B <= 65535, "resolution / batch too large");
    FPCDR_REQUIRE(p->T < (1 << 24), "more than 2^24 triangles (rast stores triangle index + 1 as a float)");
    FPCDR_REQUIRE(p->boundary_mode == FPCDR_BOUNDARY_WRAP || p->boundary_mode == FPCDR_BOUNDARY_CLAMP || p->boundary_mode == FPCDR_BOUNDARY_ZERO,
                  "bad boundary mode");
    FPCDR_REQUIRE((long long)p->Ht * p->Wt * p->C < (1ll << 30), "texture too large (the fused paths take < 2^30 texel values)");
    FPCDR_REQUIRE(!p->mip, "the mip-mapped lookup is part of fpcdr_render_loss_fwd only");
    hipStream_t st = (hipStream_t)stream;
    const RasterScratch rs = raster_scratch(p->scratch, p->B, p->T);
    TriRec *recs = rs.recs;
    TriBox *boxes = rs.boxes, *cboxes = rs.cboxes;
    ImgBox *ibox = rs.ibox;
    hipLaunchKernelGGL(k_init_ibox, dim3(fpcdr_cdiv(p->B, 256)), dim3(256), 0, st, ibox, p->B);
    hipLaunchKernelGGL(k_setup<false>, dim3(fpcdr_cdiv(p->T, 256), p->B), dim3(256), 0, st, (const float4 *)p->pos, p->tri,
                       p->B, p->V, p->T, p->H, p->W, recs, boxes, cboxes, ibox, (uint8_t *)nullptr, (const int32_t *)nullptr, (int32_t *)nullptr, (int32_t *)nullptr,
                       (const int32_t *)nullptr, (uint8_t *)nullptr, rs.clip);
    hipLaunchKernelGGL(k_setup_clip<false>, dim3(p->B), dim3(256), 0, st, (const float4 *)p->pos, p->tri, p->V, p->T, p->H, p->W, recs, boxes, cboxes,
                       ibox, (uint8_t *)nullptr, (int32_t *)nullptr, (int32_t *)nullptr, rs.clip);
    dim3 grid(fpcdr_cdiv(p->W, BIN), fpcdr_cdiv(p->H, BIN), p->B);
    ShadeArgs sh = {(const float2 *)p->uv, p->uv_tri, p->tex, p->color, p->Ht, p->Wt, p->C, p->boundary_mode,
                    nullptr, p->empty_color, (const float2 *)p->tri_uv};
    const size_t nbins = (size_t)p->B * grid.y * grid.x;
    if (p->occ) {
        FPCDR_REQUIRE(p->empty_color != nullptr, "sparse mode needs empty_color");
        sh.occ = (uint8_t *)p->occ + fpcdr_queue_layout_of(p->B, p->H, p->W).occ_raw;   // raw byte map behind the window masks
    }
    hipLaunchKernelGGL((k_bins<false, true>), grid, dim3(256), 0, st, (const float4 *)p->pos, p->tri, p->V, p->T, p->H, p->W,
                       recs, boxes, cboxes, ibox, (float4 *)p->rast, (float4 *)nullptr, sh);
    if (p->occ)
        hipLaunchKernelGGL(k_occ_window, dim3(fpcdr_cdiv((long long)nbins, 256)), dim3(256), 0, st, sh.occ, p->B, (int)grid.y,
                           (int)grid.x, p->occ);
    FPCDR_CHECK_LAUNCH();
    return FPCDR_OK;
}

extern "C" int fpcdr_render_bwd(const fpcdr_render_bwd_params *p, void *stream) {
    FPCDR_REQUIRE(p != nullptr, "null params");
    FPCDR_REQUIRE(p->pos && p->tri && p->uv && p->uv_tri && p->tex && p->rast && p->dy, "null pointer");
    FPCDR_REQUIRE(p->grad_pos || p->grad_tex, "nothing to compute");
    FPCDR_REQUIRE(p->B > 0 && p->V > 0 && p->T > 0 && p->H > 0 && p->W > 0 && p->Ht > 0 && p->Wt > 0 && p->C > 0,
                  "sizes must be positive");
    FPCDR_REQUIRE(p->B <= 65535, "more than 65535 images per call");
    dim3 grid(fpcdr_cdiv(p->W, 16), fpcdr_cdiv(p->H, 16), p->B);
    hipLaunchKernelGGL(k_render_bwd, grid, dim3(256), 0, (hipStream_t)stream, (const float4 *)p->pos, p->tri,
                       (const float2 *)p->uv, p->uv_tri, p->tex, (const float4 *)p->rast, p->dy, p->V, p->T, p->H, p->W, p->Ht,
                       p->Wt, p->C, p->boundary_mode, p->grad_pos, p->grad_tex, (const float2 *)p->tri_uv);
    FPCDR_CHECK_LAUNCH();
    return FPCDR_OK;
}

extern "C" int fpcdr_render_loss_fwd(const fpcdr_render_fwd_params *p, const fpcdr_aa_loss_fwd_params *l, uint32_t *cmask,
                                     void *stream) {
    FPCDR_REQUIRE(p != nullptr && l != nullptr && cmask != nullptr, "null params");
    FPCDR_REQUIRE(p->pos && p->tri && p->scratch && p->uv && p->uv_tri && p->tex && p->rast && p->color, "null pointer");
    FPCDR_REQUIRE(p->occ && p->empty_color, "fpcdr_render_loss_fwd works in sparse mode only (occ, empty_color)");
    FPCDR_REQUIRE(((size_t)p->occ & 3) == 0 && ((size_t)cmask & 7) == 0, "occ must be 4-byte and cmask 8-byte aligned");
    FPCDR_REQUIRE(l->adj && l->ref && l->sil && l->flags && l->grad_aa && l->loss_sum, "null pointer");
    FPCDR_REQUIRE(l->color == p->color && l->rast == p->rast && l->pos == p->pos && l->tri == p->tri && l->occ == p->occ &&
                      l->empty_color == p->empty_color && l->B == p->B && l->H == p->H && l->W == p->W && l->C == p->C &&
                      l->V == p->V && l->T == p->T,
                  "the two parameter blocks describe different batches");
    FPCDR_REQUIRE(p->B > 0 && p->V > 0 && p->T > 0 && p->H > 0 && p->W > 0 && p->Vt > 0 && p->Ht > 0 && p->Wt > 0 && p->C > 0,
                  "sizes must be positive");
    FPCDR_REQUIRE(p->C == 1 || p->C == 3 || p->C == 4, "fused objective supports C = 1, 3, 4");
    FPCDR_REQUIRE(p->H <= 32767 && p->W <= 32767 && p->B <= 65535, "resolution / batch too large");
    FPCDR_REQUIRE(p->T < (1 << 24), "more than 2^24 triangles");
    FPCDR_REQUIRE(p->boundary_mode == FPCDR_BOUNDARY_WRAP || p->boundary_mode == FPCDR_BOUNDARY_CLAMP || p->boundary_mode == FPCDR_BOUNDARY_ZERO,
                  "bad boundary mode");
    FPCDR_REQUIRE((long long)p->Ht * p->Wt * p->C < (1ll << 30), "texture too large (the fused paths take < 2^30 texel values)");
    if (p->mip) {
        FPCDR_REQUIRE(p->n_levels >= 0 && p->n_levels <= FPCDR_MAX_MIP, "bad n_levels");
        for (int lvl = 1; lvl <= p->n_levels; ++lvl) {
            FPCDR_REQUIRE(p->tex_mip[lvl - 1] != nullptr, "missing mip level");
            FPCDR_REQUIRE(!((p->Ht >> (lvl - 1)) & 1) && !((p->Wt >> (lvl - 1)) & 1), "mip levels need even sizes");
        }
    }
    hipStream_t st = (hipStream_t)stream;
    // (the silhouette kernel also zeroes the call's flag planes: the caller need not)
    int rc = fpcdr_launch_sil(p->pos, p->tri, l->adj, p->B, p->V, p->T, p->H, p->W, l->sil, l->flags,
                              (size_t)2 * p->B * p->H * FPCDR_AA_ROW_WORDS(p->W) * sizeof(uint64_t), st);
    if (rc) return rc;
    const RasterScratch rs = raster_scratch(p->scratch, p->B, p->T);
    TriRec *recs = rs.recs;
    TriBox *boxes = rs.boxes, *cboxes = rs.cboxes;
    ImgBox *ibox = rs.ibox;
    const int OX = fpcdr_cdiv(p->W, BIN), OY = fpcdr_cdiv(p->H, BIN);
    const size_t nbins = (size_t)p->B * OY * OX;
    FPCDR_REQUIRE(nbins < 0x7ffffff0ULL, "too many bins for one call");      // (list launches are rounded up to a multiple of 8 workgroups)
    const fpcdr_queue_layout q = fpcdr_queue_layout_of(p->B, p->H, p->W);
    char *cm = (char *)cmask, *oc = (char *)p->occ;
    int32_t *hdr = (int32_t *)(cm + q.cm_hdr), *bin_list = (int32_t *)(cm + q.cm_bin_list), *fix_list = (int32_t *)(cm + q.cm_fix_list);
    uint8_t *live = (uint8_t *)(cm + q.cm_live);
    int32_t *hdr_bwd = (int32_t *)(oc + q.occ_hdr), *bwd_list = (int32_t *)(oc + q.occ_bwd_list);
    uint8_t *occ_raw = (uint8_t *)(oc + q.occ_raw);
    // header in occ (survives the call; include/fpcdr.h FPCDR_OCC_COUNTS): [0] backward bins, [2] live bins, [3] antialias-fix bins
    hipLaunchKernelGGL(k_init_queue, dim3(256), dim3(256), 0, st, ibox, p->B, (uint32_t *)live, (long long)(align_up(nbins, 4) / 4),
                       (uint32_t *)oc, (long long)(q.occ_hdr / 4), hdr, hdr_bwd);
    hipLaunchKernelGGL(k_setup<false>, dim3(fpcdr_cdiv(p->T, 256), p->B), dim3(256), 0, st, (const float4 *)p->pos, p->tri,
                       p->B, p->V, p->T, p->H, p->W, recs, boxes, cboxes, ibox, live, (const int32_t *)nullptr, (int32_t *)nullptr, (int32_t *)nullptr,
                       (const int32_t *)nullptr, (uint8_t *)nullptr, rs.clip);
    hipLaunchKernelGGL(k_setup_clip<false>, dim3(p->B), dim3(256), 0, st, (const float4 *)p->pos, p->tri, p->V, p->T, p->H, p->W, recs, boxes, cboxes,
                       ibox, live, (int32_t *)nullptr, (int32_t *)nullptr, rs.clip);
    int32_t *n_bins = hdr_bwd + 2, *n_fix = hdr_bwd + 3;     // all three counts live in the occ header, where the caller finds them
    const int nblk = fpcdr_cdiv((long long)nbins, 256);
    int32_t *blk = (int32_t *)(cm + q.cm_blk);          // [2][nblk] per-block counts, then offsets
    hipLaunchKernelGGL(k_list_count<0>, dim3(nblk), dim3(256), 0, st, live, (long long)nbins, OY, OX, blk, (uint16_t *)nullptr,
                       p->tex, p->Ht, p->Wt, p->C, p->boundary_mode, p->empty_color);
    hipLaunchKernelGGL(k_list_scan, dim3(1), dim3(1024), 0, st, blk, nblk, n_bins, (int32_t *)nullptr);
    hipLaunchKernelGGL(k_list_write<0>, dim3(nblk), dim3(256), 0, st, live, (long long)nbins, OY, OX, blk, bin_list, (int32_t *)nullptr);
    ShadeArgs sh = {(const float2 *)p->uv, p->uv_tri, p->tex, p->color, p->Ht, p->Wt, p->C, p->boundary_mode,
                    occ_raw, p->empty_color, (const float2 *)p->tri_uv,
                    l->sil, l->ref, l->grad_aa, cmask, (unsigned long long *)(cm + q.cm_edges), l->loss_sum, l->bg, l->color_scale,
                    l->grad_scale};
    if (p->mip) {
        for (int lvl = 0; lvl < FPCDR_MAX_MIP; ++lvl) sh.mip[lvl] = lvl < p->n_levels ? p->tex_mip[lvl] : nullptr;
        sh.n_levels = p->n_levels;
    }
    // hinted single-shot launch + strided sweep of the rest (l->cap_bins <= 0: no hint, one workgroup per possible entry)
    const fpcdr_bin_decode dc = fpcdr_make_bin_decode(OX, OY);
    const int cap_bins = (l->cap_bins > 0 && (size_t)l->cap_bins < nbins) ? l->cap_bins : (int)nbins;
    if (p->mip && p->C != 1)
        hipLaunchKernelGGL((k_bins_list<false, true, true, 0, -1, true>), dim3(fpcdr_list_grid(cap_bins)), dim3(256), 0, st, bin_list, n_bins, cap_bins, OX, OY, dc,
                       (const float4 *)p->pos, p->tri, p->V, p->T, p->H, p->W, recs, boxes, cboxes, ibox, (float4 *)p->rast,
                       (float4 *)nullptr, sh);
    else if (p->mip)
        hipLaunchKernelGGL((k_bins_list<false, true, true, 1, -1, true>), dim3(fpcdr_list_grid(cap_bins)), dim3(256), 0, st, bin_list, n_bins, cap_bins, OX, OY, dc,
                       (const float4 *)p->pos, p->tri, p->V, p->T, p->H, p->W, recs, boxes, cboxes, ibox, (float4 *)p->rast,
                       (float4 *)nullptr, sh);
    else if (p->C == 1 && p->boundary_mode == FPCDR_BOUNDARY_WRAP)     // the reference's case, with both as compile-time constants
        hipLaunchKernelGGL((k_bins_list<false, true, true, 1, FPCDR_BOUNDARY_WRAP>), dim3(fpcdr_list_grid(cap_bins)), dim3(256), 0, st, bin_list, n_bins, cap_bins, OX, OY, dc,
                       (const float4 *)p->pos, p->tri, p->V, p->T, p->H, p->W, recs, boxes, cboxes, ibox, (float4 *)p->rast,
                       (float4 *)nullptr, sh);
    else
        hipLaunchKernelGGL((k_bins_list<false, true, true>), dim3(fpcdr_list_grid(cap_bins)), dim3(256), 0, st, bin_list, n_bins, cap_bins, OX, OY, dc,
                       (const float4 *)p->pos, p->tri, p->V, p->T, p->H, p->W, recs, boxes, cboxes, ibox, (float4 *)p->rast,
                       (float4 *)nullptr, sh);
    if ((size_t)cap_bins < nbins) {
        if (p->mip)
            hipLaunchKernelGGL((k_bins_queue<false, true, true, true>), dim3(FPCDR_SWEEP_WGS), dim3(256), 0, st, bin_list, n_bins, cap_bins, OX, OY, dc,
                               (const float4 *)p->pos, p->tri, p->V, p->T, p->H, p->W, recs, boxes, cboxes, ibox, (float4 *)p->rast,
                               (float4 *)nullptr, sh);
        else
            hipLaunchKernelGGL((k_bins_queue<false, true, true>), dim3(FPCDR_SWEEP_WGS), dim3(256), 0, st, bin_list, n_bins, cap_bins, OX, OY, dc,
                               (const float4 *)p->pos, p->tri, p->V, p->T, p->H, p->W, recs, boxes, cboxes, ibox, (float4 *)p->rast,
                               (float4 *)nullptr, sh);
    }
    hipLaunchKernelGGL(k_list_count<1>, dim3(nblk), dim3(256), 0, st, occ_raw, (long long)nbins, OY, OX, blk, p->occ,
                       (const float *)nullptr, 0, 0, 0, 0, (float *)nullptr);
    hipLaunchKernelGGL(k_list_scan, dim3(1), dim3(1024), 0, st, blk, nblk, n_fix, hdr_bwd);
    hipLaunchKernelGGL(k_list_write<1>, dim3(nblk), dim3(256), 0, st, occ_raw, (long long)nbins, OY, OX, blk, fix_list, bwd_list);
    FPCDR_CHECK_LAUNCH();
    return fpcdr_launch_aa_fix(l, cmask, (const unsigned long long *)(cm + q.cm_edges), fix_list, n_fix, (int)nbins, st);
}

#endif  // FPCDR_TWOCALL

// First half of fpcdr_objective_fwd (objective.hip; not part of the C ABI): set-up, the list of live bins, the rasteriser in its IDS form
// (id planes only) and the ordered list of OCCUPIED bins + window masks for the shading kernels.  The caller has run k_sil2.
int fpcdr_launch_raster_ids(const fpcdr_objective_params *p, hipStream_t st, const int32_t **occ_list, const int32_t **n_occ_dev,
                            const FpcdrZeroList &zl_in, bool sil_in_setup) {
    const RasterScratch rs = raster_scratch(p->scratch, p->B, p->T);
    const int OX = fpcdr_cdiv(p->W, BIN), OY = fpcdr_cdiv(p->H, BIN);
    const size_t nbins = (size_t)p->B * OY * OX;
    FPCDR_REQUIRE(nbins < 0x7ffffff0ULL, "too many bins for one call");
    const fpcdr_queue_layout q = fpcdr_queue_layout_of(p->B, p->H, p->W, true);
    char *cm = (char *)p->cmask, *oc = (char *)p->occ;
    int32_t *hdr = (int32_t *)(cm + q.cm_hdr), *bin_list = (int32_t *)(cm + q.cm_bin_list), *olist = (int32_t *)(cm + q.cm_fix_list);
    uint8_t *live = (uint8_t *)(cm + q.cm_live);
    int32_t *hdr_occ = (int32_t *)(oc + q.occ_hdr);
    uint8_t *occ_raw = (uint8_t *)(oc + q.occ_raw);
    // per-bin triangle lists (p->binlist: counts, then BL_CAP slots per bin), or the chunk scan for every bin when the caller gave none
    int32_t *bin_cnt = (int32_t *)p->binlist, *tri_lists = bin_cnt ? bin_cnt + align_up(nbins, 64) : nullptr;
    // list building: two launches per round (k_list_count, k_list_write_sum) up to LW_MAX_BLOCKS workgroups, three beyond
    const int nblk = fpcdr_cdiv((long long)nbins, 256);
    const bool two_launch_lists = nblk <= LW_MAX_BLOCKS;
    int32_t *blk = (int32_t *)(cm + q.cm_blk);
    // the largest buffer of the list, if it is 16-byte aligned, is left to the set-up kernel (see k_setup)
    FpcdrZeroList zl = zl_in;
    uint4 *big16 = nullptr;
    long long big_n16 = 0;
    {
        int big = -1;
        for (int r = 0; r < zl.count; ++r)
            if (zl.n[r] >= (1ll << 20) && ((size_t)zl.p[r] & 15) == 0 && (zl.n[r] & 3) == 0 && (big < 0 || zl.n[r] > zl.n[big])) big = r;
        if (big >= 0) {
            big16 = reinterpret_cast<uint4 *>(zl.p[big]);
            big_n16 = zl.n[big] >> 2;
            zl.p[big] = zl.p[zl.count - 1]; zl.n[big] = zl.n[zl.count - 1]; --zl.count;
        }
    }
    long long zero_words = 0;
    for (int r = 0; r < zl.count; ++r) zero_words += zl.n[r];
    const int init_grid = (int)std::min<long long>(2048, std::max<long long>(256, zero_words / 4096));
    hipLaunchKernelGGL(k_init_objective, dim3(init_grid), dim3(256), 0, st, rs.ibox, p->B, (uint32_t *)live, (long long)(align_up(nbins, 4) / 4),
                       (uint32_t *)oc, (long long)(q.occ_hdr / 4), hdr, hdr_occ, bin_cnt, (long long)nbins, zl);
    // (sil_in_setup: the caller has not computed the silhouette bits -- the set-up kernel does, one walk of the index -> position chain)
#define SETUP(BL, SL)                                                                                                              \
    hipLaunchKernelGGL((k_setup<BL, SL>), dim3(fpcdr_cdiv(p->T, 256), p->B), dim3(256), 0, st, (const float4 *)p->pos, p->tri,     \
                       p->B, p->V, p->T, p->H, p->W, rs.recs, rs.boxes, rs.cboxes, rs.ibox, live, (const int32_t *)nullptr, bin_cnt, tri_lists, \
                       p->adj, p->sil, rs.clip, big16, big_n16)
    if (bin_cnt) { if (sil_in_setup) SETUP(true, true); else SETUP(true, false); }
    else { if (sil_in_setup) SETUP(false, true); else SETUP(false, false); }
#undef SETUP
    // (the triangles with a vertex at w <= 0, normally none: one workgroup per image looks at its list's count and leaves)
    if (bin_cnt)
        hipLaunchKernelGGL(k_setup_clip<true>, dim3(p->B), dim3(256), 0, st, (const float4 *)p->pos, p->tri, p->V, p->T, p->H, p->W, rs.recs, rs.boxes,
                           rs.cboxes, rs.ibox, live, bin_cnt, tri_lists, rs.clip);
    else
        hipLaunchKernelGGL(k_setup_clip<false>, dim3(p->B), dim3(256), 0, st, (const float4 *)p->pos, p->tri, p->V, p->T, p->H, p->W, rs.recs, rs.boxes,
                           rs.cboxes, rs.ibox, live, (int32_t *)nullptr, (int32_t *)nullptr, rs.clip);
    int32_t *n_bins = hdr_occ + 2, *n_occ = hdr_occ + 3;      // (include/fpcdr.h FPCDR_OCC_COUNTS_OFFSET)
    hipLaunchKernelGGL(k_list_count<0>, dim3(nblk), dim3(256), 0, st, live, (long long)nbins, OY, OX, blk, (uint16_t *)nullptr,
                       p->tex, p->Ht, p->Wt, p->C, p->boundary_mode, p->empty_color);
    if (two_launch_lists)
        hipLaunchKernelGGL(k_list_write_sum<0>, dim3(nblk), dim3(256), 0, st, live, (long long)nbins, OY, OX, blk, bin_list, (int32_t *)nullptr,
                           n_bins, (int32_t *)nullptr);
    else {
        hipLaunchKernelGGL(k_list_scan, dim3(1), dim3(1024), 0, st, blk, nblk, n_bins, (int32_t *)nullptr);
        hipLaunchKernelGGL(k_list_write<0>, dim3(nblk), dim3(256), 0, st, live, (long long)nbins, OY, OX, blk, bin_list, (int32_t *)nullptr);
    }
    ShadeArgs sh = {};
    sh.occ = occ_raw;
    sh.sil = p->sil;
    sh.idp = p->idp;
    sh.bin_cnt = bin_cnt;
    sh.bin_list = tri_lists;
    const fpcdr_bin_decode dc = fpcdr_make_bin_decode(OX, OY);
    const int cap_bins = (p->cap_bins > 0 && (size_t)p->cap_bins < nbins) ? p->cap_bins : (int)nbins;
    // the silhouette bits computed by the caller on another stream (sil_ready): the rasteriser kernel is their first reader
    if (p->sil_ready && p->sil_event)
        FPCDR_REQUIRE(hipStreamWaitEvent(st, (hipEvent_t)p->sil_event, 0) == hipSuccess, "hipStreamWaitEvent(sil_event) failed");
    hipLaunchKernelGGL((k_bins_list<false, false, false, 0, -1, false, true>), dim3(fpcdr_list_grid(cap_bins)), dim3(256), 0, st, bin_list, n_bins,
                       cap_bins, OX, OY, dc, (const float4 *)p->pos, p->tri, p->V, p->T, p->H, p->W, rs.recs, rs.boxes, rs.cboxes, rs.ibox,
                       (float4 *)nullptr, (float4 *)nullptr, sh);
    if ((size_t)cap_bins < nbins)
        hipLaunchKernelGGL((k_bins_queue<false, false, false, false, true>), dim3(FPCDR_SWEEP_WGS), dim3(256), 0, st, bin_list, n_bins, cap_bins,
                           OX, OY, dc, (const float4 *)p->pos, p->tri, p->V, p->T, p->H, p->W, rs.recs, rs.boxes, rs.cboxes, rs.ibox,
                           (float4 *)nullptr, (float4 *)nullptr, sh);
    hipLaunchKernelGGL(k_list_count<2>, dim3(nblk), dim3(256), 0, st, occ_raw, (long long)nbins, OY, OX, blk, p->occ,
                       (const float *)nullptr, 0, 0, 0, 0, (float *)nullptr);
    if (two_launch_lists)
        hipLaunchKernelGGL(k_list_write_sum<2>, dim3(nblk), dim3(256), 0, st, occ_raw, (long long)nbins, OY, OX, blk, olist, (int32_t *)nullptr,
                           n_occ, (int32_t *)nullptr);
    else {
        hipLaunchKernelGGL(k_list_scan, dim3(1), dim3(1024), 0, st, blk, nblk, n_occ, (int32_t *)nullptr);
        hipLaunchKernelGGL(k_list_write<2>, dim3(nblk), dim3(256), 0, st, occ_raw, (long long)nbins, OY, OX, blk, olist, (int32_t *)nullptr);
    }
    FPCDR_CHECK_LAUNCH();
    *occ_list = olist;
    *n_occ_dev = n_occ;
    return FPCDR_OK;
}

extern "C" size_t fpcdr_binlist_bytes(int32_t B, int32_t H, int32_t W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    const size_t nbins = (size_t)B * FPCDR_OCC_DIM(H) * FPCDR_OCC_DIM(W);
    return (align_up(nbins, 64) + nbins * BL_CAP) * sizeof(int32_t);
}

extern "C" size_t fpcdr_idplane_bytes(int32_t B, int32_t H, int32_t W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    return (size_t)B * FPCDR_OCC_DIM(H) * FPCDR_OCC_DIM(W) * (BIN * BIN) * sizeof(uint32_t);
}

extern "C" size_t fpcdr_occ_bytes(int32_t B, int32_t H, int32_t W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    return fpcdr_queue_layout_of(B, H, W).occ_bytes;
}

extern "C" size_t fpcdr_cmask_bytes(int32_t B, int32_t H, int32_t W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    return fpcdr_queue_layout_of(B, H, W).cm_bytes;
}

extern "C" size_t fpcdr_objective_cmask_bytes(int32_t B, int32_t H, int32_t W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    return fpcdr_queue_layout_of(B, H, W, true).cm_bytes;
}
