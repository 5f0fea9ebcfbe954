// Shared helpers for the gfx950 kernels (wave64, CDNA4).  Not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/fpcdr.h"
#include "../../include/fpcdr_twocall.h"

#define FPCDR_WAVE 64

void fpcdr_set_error(const char *fmt, ...);

#define FPCDR_REQUIRE(cond, msg)                                   \
    do {                                                           \
        if (!(cond)) {                                             \
            fpcdr_set_error("%s: %s", __func__, msg);              \
            return FPCDR_EINVAL;                                   \
        }                                                          \
    } while (0)

#define FPCDR_CHECK_LAUNCH()                                                        \
    do {                                                                            \
        hipError_t e_ = hipGetLastError();                                          \
        if (e_ != hipSuccess) {                                                     \
            fpcdr_set_error("%s: launch failed: %s", __func__, hipGetErrorString(e_)); \
            return FPCDR_ELAUNCH;                                                   \
        }                                                                           \
    } while (0)

static inline int fpcdr_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// internal cross-file launchers (fused.hip), used by fpcdr_render_loss_fwd (rasterize.hip)
int fpcdr_launch_sil(const float *pos, const int32_t *tri, const int32_t *adj, int B, int V, int T, int H, int W, uint8_t *sil,
                     void *zero_dst, size_t zero_bytes, hipStream_t st);
int fpcdr_launch_aa_fix(const fpcdr_aa_loss_fwd_params *p, const uint32_t *cmask, const unsigned long long *edges,
                        const int32_t *fix_list, const int32_t *fix_count, int nbins, hipStream_t st);

// buffers the first kernel of fpcdr_objective_fwd zero-fills beside its own maps (fpcdr_objective_params.zero_outputs): 4-byte words
struct FpcdrZeroList {
    static constexpr int MAXR = 24;
    uint32_t *p[MAXR];
    long long n[MAXR];
    int count;
    bool overflow;
    // (a dropped entry would be an un-zeroed accumulator, i.e. a wrong gradient: the entry points FPCDR_REQUIRE(!overflow))
    __host__ void add(void *ptr, long long words) {
        if (!ptr || words <= 0) return;
        if (count >= MAXR) { overflow = true; return; }
        p[count] = (uint32_t *)ptr; n[count] = words; ++count;
    }
};
static_assert(FpcdrZeroList::MAXR >= 5 + FPCDR_MAX_MIP + 2, "esum, loss, grad_pos, grad_tex, zero_extra, every mip level's gradient (+ 2 spare)");
int fpcdr_launch_raster_ids(const fpcdr_objective_params *p, hipStream_t st, const int32_t **occ_list, const int32_t **n_occ_dev,
                            const FpcdrZeroList &zl, bool sil_in_setup);

// workgroups of the strided sweep behind a hinted single-shot launch (normally they find nothing to do)
#define FPCDR_SWEEP_WGS 256

// Byte offsets inside the two buffers of the sparse objective (include/fpcdr.h: fpcdr_occ_bytes / fpcdr_cmask_bytes).
//   occ   (saved for the backward call): window masks u16[nb] | raw occupancy u8[nb] | "bin holds antialias flags" u8[nb] |
//         header i32[16] | backward bin list i32[nb]
//   cmask (forward scratch): candidate row masks u32[nb*32] | border lines u64[nb*128] (one-pass: hit masks u32[nb*32] | f32[128]) | header i32[16] | live bin list i32[nb] |
//         antialias-fix bin list i32[nb] | live map u8[nb] | per-block counts i32[2][ceil(nb / 256)]
struct fpcdr_queue_layout {
    size_t occ_raw, occ_binflag, occ_hdr, occ_bwd_list, occ_bytes;
    size_t cm_edges, cm_esum, cm_hdr, cm_bin_list, cm_fix_list, cm_live, cm_blk, cm_bytes;
};
// (onepass: fpcdr_objective_fwd keeps 32 row masks per bin where the two-call form keeps 128 border lines of 8 bytes: 128 B instead of 1 KB)
static inline fpcdr_queue_layout fpcdr_queue_layout_of(int B, int H, int W, bool onepass = false) {
    const size_t nb = (size_t)B * FPCDR_OCC_DIM(H) * FPCDR_OCC_DIM(W);
    const size_t nb4 = (nb + 3) / 4 * 4;
    fpcdr_queue_layout q;
    q.occ_raw = 2 * nb;
    q.occ_binflag = 3 * nb;
    q.occ_hdr = (4 * nb + 3) / 4 * 4;
    q.occ_bwd_list = q.occ_hdr + 64;
    q.occ_bytes = q.occ_bwd_list + 4 * nb;
    q.cm_edges = nb * 128;
    q.cm_esum = q.cm_edges + nb * 128;      // (onepass: 512 B of gradient slots behind the hit masks)
    q.cm_hdr = q.cm_edges + nb * (onepass ? 128 : 1024) + (onepass ? 512 : 0);
    q.cm_bin_list = q.cm_hdr + 64;
    q.cm_fix_list = q.cm_bin_list + 4 * nb;
    q.cm_live = q.cm_fix_list + 4 * nb;
    q.cm_blk = q.cm_live + nb4;
    q.cm_bytes = q.cm_blk + 2 * 4 * ((nb + 255) / 256);
    return q;
}

// ---- exact division of a list entry by a launch constant ----------------------------------------------------------
// The list kernels turn a linear bin index into (image, bin row, bin column).  The hardware has no integer division: the
// compiler's expansion costs ~25 vector instructions (several of them quarter rate) per division and wave, three per
// workgroup.  With a host-made reciprocal it is one multiply-high and a shift, on the scalar unit when the index is uniform.
// q = (n * m) >> (31 + l), l = ceil(log2 d), m = floor(2^(31+l) / d) + 1: exact for every 0 <= n < 2^31, 1 <= d < 2^31.
struct fpcdr_div { uint32_t m, sh, d; };
static inline fpcdr_div fpcdr_make_div(uint32_t d) {
    uint32_t l = 0;
    while ((1ull << l) < d) ++l;
    fpcdr_div r;
    r.m = (uint32_t)(((1ull << (31 + l)) / d) + 1ull);
    r.sh = 31 + l;
    r.d = d;
    return r;
}
__device__ __forceinline__ uint32_t fpcdr_divide(uint32_t n, const fpcdr_div &v) { return (uint32_t)(((unsigned long long)n * v.m) >> v.sh); }
// linear bin index -> (image, bin row, bin column) for OX x OY bins per image
struct fpcdr_bin_decode { fpcdr_div per_image, per_row; };
static inline fpcdr_bin_decode fpcdr_make_bin_decode(int OX, int OY) { return {fpcdr_make_div((uint32_t)OX * (uint32_t)OY), fpcdr_make_div((uint32_t)OX)}; }
__device__ __forceinline__ void fpcdr_decode_bin(int lin, const fpcdr_bin_decode &dc, int &b, int &byi, int &bxi) {
    const uint32_t q = fpcdr_divide((uint32_t)lin, dc.per_image), rem = (uint32_t)lin - q * dc.per_image.d;
    const uint32_t y = fpcdr_divide(rem, dc.per_row);
    b = (int)q; byi = (int)y; bxi = (int)(rem - y * dc.per_row.d);
}

// tuning override for the number of resident workgroups per CU of a work-queue kernel (experiments; default = dflt)
static inline int fpcdr_env_int(const char *name, int dflt) {
    const char *v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}

// compute units of the current device (work-queue kernels launch a fixed number of resident workgroups per CU)
static inline int fpcdr_cu_count() {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
        n = 256;   // MI355X
    return n;
}

// ---- gathers through a 32-bit byte offset ----------------------------------------------------------------------------
// base[i] with an int index costs a sign extension, a 64-bit shift-add and two address registers per gather; with the byte offset
// formed in 32 bits the compiler uses the scalar-base addressing mode (global_load v, v_offset, s[base:base+1]): one shift.  A
// shaded pixel issues ~20 gathers, so this is ~15 % of its vector instructions.  Valid while base is uniform over the wave and
// i * sizeof(T) < 2^32: per-image vertex / triangle / record arrays, the pixels of one bin, textures below 2^30 texel values.
template <typename T> __device__ __forceinline__ const T &ld32(const T *base, unsigned int i) {
    return *reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + i * (unsigned int)sizeof(T));
}
template <typename T> __device__ __forceinline__ T &at32(T *base, unsigned int i) {
    return *reinterpret_cast<T *>(reinterpret_cast<char *>(base) + i * (unsigned int)sizeof(T));
}

// ---- list kernels: which entry a workgroup takes ------------------------------------------------
// Workgroup i of a launch runs on XCD i mod 8 (scripts/micro/atomic_scope_bench.hip), and each XCD has its own L2.  Taking entry i
// would deal NEIGHBOURING bins -- which share vertices, triangle records and texels -- to eight different L2s; instead XCD x takes
// the x-th eighth of the entries [0, m), m = min(entries, launch size), in order.  The launch is rounded up to a multiple of 8
// workgroups (fpcdr_list_grid) so that every entry below m has a workgroup.  Returns -1 for a workgroup without an entry.
#define FPCDR_XCD_LISTS 1
__host__ __device__ inline int fpcdr_list_grid(int cap) { return FPCDR_XCD_LISTS ? (cap + 7) / 8 * 8 : cap; }
__device__ __forceinline__ int fpcdr_list_item(int n_entries, int cap) {
    const int m = min(n_entries, cap);
    if (!FPCDR_XCD_LISTS) return (int)blockIdx.x < m ? (int)blockIdx.x : -1;
    const int chunk = (m + 7) >> 3;
    const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
    if (j >= chunk) return -1;
    // (r5, measured and dropped: XCD x starting its eighth a fraction x * 0.25 / 0.382 / 0.5 / 0.618 / 0.75 of the way in, so that the eight
    //  XCDs do not shade the same screen bin of the same camera in eight different frames at the same moment -- whose texel gradients go
    //  to the same addresses: k_shade 1 577-1 613 us at cfg3 for every fraction, 1 611 without: the flush atomics are bound by the number
    //  of 32-byte sector operations, not by collisions; profiles/r05_flush_experiments.txt)
    const int idx = x * chunk + j;
    return idx < m ? idx : -1;
}

// ---- region hints (include/fpcdr.h) ----------------------------------------------------------
// plane: 0 = the bin itself is occupied, 1 = the bin or one of its eight neighbours is
__device__ __forceinline__ bool fpcdr_hint_on(const uint8_t *__restrict__ hint, int plane, int B, int H, int W, int b, int y, int x) {
    const int OX = FPCDR_OCC_DIM(W), OY = FPCDR_OCC_DIM(H);
    return hint[(((size_t)plane * B + b) * OY + (y >> 5)) * OX + (x >> 5)] != 0;
}

// ---- LDS float accumulation ------------------------------------------------------------------
// ds_add_f32 costs 3 cycles PER ACTIVE LANE on gfx950 (192 cycles for a full wave, whatever the addresses), while integer LDS
// atomics run at 6-11 cycles per wave-instruction (scripts/micro/lds_atomic_bench.hip: int add 5.7, u64 min 7.9, 32-bit
// compare-and-swap 11.4).  A float add built from a read and a compare-and-swap loop is therefore several times faster; lanes
// that share an address simply go round again.
__device__ __forceinline__ void lds_add_f32(float *addr, float v) {
    unsigned int *a = reinterpret_cast<unsigned int *>(addr);
    unsigned int old = *a, assumed;
    do {
        assumed = old;
        old = atomicCAS(a, assumed, __float_as_uint(__uint_as_float(assumed) + v));
    } while (old != assumed);
}
// ds_add_f64, on the other hand, is a native 8-cycle instruction (same microbenchmark): an LDS accumulator kept in DOUBLE takes a
// float term with one conversion and one non-returning atomic -- no read, no compare-and-swap loop -- and the sum no longer
// depends (to float precision) on the order in which lanes arrive.  Costs twice the LDS space.
__device__ __forceinline__ void lds_add_f64(double *addr, float v) { atomicAdd(addr, (double)v); }
// two adjacent floats (8-byte aligned) in one 64-bit compare-and-swap
__device__ __forceinline__ void lds_add_f32x2(float *addr, float vx, float vy) {
    unsigned long long *a = reinterpret_cast<unsigned long long *>(addr);
    unsigned long long old = *a, assumed;
    do {
        assumed = old;
        const float x = __uint_as_float((unsigned int)assumed) + vx, y = __uint_as_float((unsigned int)(assumed >> 32)) + vy;
        old = atomicCAS(a, assumed, ((unsigned long long)__float_as_uint(y) << 32) | __float_as_uint(x));
    } while (old != assumed);
}

// ---- wave-level helpers ---------------------------------------------------------------------

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

// Sum over the lanes whose bit is set in `mask` (mask must contain the calling lane and be the
// same in all participating lanes).  Result valid in every participating lane.
__device__ __forceinline__ float wave_sum_masked(float v, unsigned long long mask) {
    // butterfly over all 64 lanes with non-members contributing 0
    int l = lane_id();
    float x = ((mask >> l) & 1ull) ? v : 0.0f;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) x += __shfl_xor(x, o, 64);
    return x;
}

// Full-wave (64 lanes) f32 sum with DPP row operations (VALU rate; __shfl_xor lowers to ds_bpermute, several
// times slower).  gfx9 family: row_bcast15 / row_bcast31 exist.  The total is returned in every lane.
__device__ __forceinline__ float wave_sum_dpp(float v) {
    // quad_perm [1,0,3,2] = 0xB1, [2,3,0,1] = 0x4E, row_half_mirror = 0x141, row_mirror = 0x140,
    // row_bcast15 = 0x142 (row_mask 0xA), row_bcast31 = 0x143 (row_mask 0xC)
    int x = __float_as_int(v);
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, false)); x = __float_as_int(v);
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, false)); x = __float_as_int(v);
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x141, 0xF, 0xF, false)); x = __float_as_int(v);
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x140, 0xF, 0xF, false)); x = __float_as_int(v);
    // after the four steps every lane holds the sum of its 16-lane row
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false)); x = __float_as_int(v);
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// Group-reduce-then-atomic: lanes of a wave that share `key` (>= 0) sum their `n` values and the
// group leader issues the atomics.  Lanes with key < 0 do not participate.  All lanes of the wave
// must call this (convergent).  Sums with DPP row operations (the __shfl_xor butterflies this used to run lower to
// ds_bpermute, several times slower, and made the operator backward kernels LDS-bound).
template <int N>
__device__ __forceinline__ void wave_group_atomic_add(int key, float *const (&dst)[N], const float (&val)[N]) {
    unsigned long long todo = __ballot(key >= 0);
    const int l = lane_id();
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int k = __builtin_amdgcn_readlane(key, leader);
        const unsigned long long grp = __ballot(key == k) & todo;
        const bool mine = (grp >> l) & 1ull;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const float x = wave_sum_dpp(mine ? val[i] : 0.0f);
            if (l == leader && x != 0.0f) atomicAdd(dst[i], x);
        }
        todo &= ~grp;
    }
}

// Lanes of a wave that share `key` (>= 0) sum their N values; the group's first lane calls emit(key, sums).
// Lanes with key < 0 do not participate.  Convergent: every lane of the wave must call it.
// The sums are parked in the leader lanes' registers while the groups are walked, and ALL leaders emit together
// afterwards: emitting inside the loop would run the (LDS / memory) latency of every group one after the other in a
// single lane (measured: 2.5 ms of a 5.2 ms kernel).
template <int N, typename Emit>
__device__ __forceinline__ void wave_group_reduce(int key, const float (&val)[N], Emit &&emit) {
    unsigned long long todo = __ballot(key >= 0);
    const int l = lane_id();
    float out[N];
    bool is_leader = false;
#pragma unroll
    for (int i = 0; i < N; ++i) out[i] = 0.0f;
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int k = __builtin_amdgcn_readlane(key, leader);
        const unsigned long long grp = __ballot(key == k) & todo;
        const bool mine = (grp >> l) & 1ull;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const float sm = wave_sum_dpp(mine ? val[i] : 0.0f);
            if (l == leader) out[i] = sm;
        }
        if (l == leader) is_leader = true;
        todo &= ~grp;
    }
    if (is_leader) emit(key, out);
}

// Segmented sums over runs of equal `key` in LANE ORDER: consecutive lanes holding the same key (>= 0) form a segment, and
// the last lane of every segment calls emit(key, sums) with the segment's totals; all of them at once, after a single
// 6-step segmented scan (DPP row_shr 1,2,4,8 + row_bcast15/31) whose cost does not depend on the number of segments.
// Lanes with key < 0 are skipped.  Convergent: every lane of the wave must call it.  Callers order their lanes so that
// pixels of one triangle sit next to each other (rows of a tile, serpentine).
// ROW16: segments are additionally cut at every 16-lane DPP row: four scan steps instead of six (the two cross-row broadcast
// steps go), at the price of one more tail wherever a run crosses a row boundary.
template <int N, bool ROW16 = false, typename Emit>
__device__ __forceinline__ void wave_segment_reduce(int key, const float (&val)[N], Emit &&emit) {
    const int l = lane_id();
    const int prev = __builtin_amdgcn_update_dpp(key, key, 0x138, 0xF, 0xF, false);   // wave_shr:1 (lane 0 keeps its own)
    int f = ((ROW16 ? (l & 15) == 0 : l == 0) || prev != key) ? 1 : 0;                  // segment start
    const int next = __builtin_amdgcn_update_dpp(key, key, 0x130, 0xF, 0xF, false);   // wave_shl:1 (lane 63 keeps its own)
    const bool tail = ((ROW16 ? (l & 15) == 15 : l == 63) || next != key);
    float v[N];
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = val[i];
    // row_shr:n = 0x110 + n; lanes without a source inside their row of 16 read `old` = 0
#define FPCDR_SEG_STEP(CTRL, ROWMASK)                                                                                   \
    {                                                                                                                    \
        const int fp = __builtin_amdgcn_update_dpp(0, f, CTRL, ROWMASK, 0xF, false);                                     \
        _Pragma("unroll") for (int i = 0; i < N; ++i) {                                                                  \
            const float vp = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v[i]), CTRL, ROWMASK, 0xF, false)); \
            v[i] += f ? 0.0f : vp;                                                                                       \
        }                                                                                                                \
        f |= fp;                                                                                                         \
    }
    FPCDR_SEG_STEP(0x111, 0xF)
    FPCDR_SEG_STEP(0x112, 0xF)
    FPCDR_SEG_STEP(0x114, 0xF)
    FPCDR_SEG_STEP(0x118, 0xF)
    if (!ROW16) {
        FPCDR_SEG_STEP(0x142, 0xA)   // row_bcast15: lane 15 of rows 0, 2 into rows 1, 3
        FPCDR_SEG_STEP(0x143, 0xC)   // row_bcast31: lane 31 into rows 2, 3
    }
#undef FPCDR_SEG_STEP
    if (tail && key >= 0) emit(key, v);
}

// The same segmented sums for N = 9 with the scan written out as v_fmac_f32_dpp (one instruction per value and step: the
// shifted neighbour times the lane's 0 / 1 "no segment start since" mask is added in place; v_mul_f32_dpp carries the mask).
// 6 x 10 instructions instead of ~250 through update_dpp + select + add: the backward kernel is vector-issue bound
// (profiles/r02: 80 % of the issue slots) and a quarter of its instructions were this scan.  Lanes without a DPP source are
// disabled for that instruction (no bound_ctrl), which leaves value and mask unchanged -- the neutral element of both.
// Finite values only: the mask is a FACTOR (0 / 1), so a NaN or infinite value of one run reaches the sums of the runs after it in the
// wave (0 * NaN), where the select-based wave_segment_reduce keeps it inside its own run.  The callers' values are gradients of a
// finite loss; a step whose gradients are not finite is lost either way (Adam's moments keep the NaN).
template <typename Emit>
__device__ __forceinline__ void wave_segment_reduce9(int key, const float (&val)[9], Emit &&emit) {
    const int l = lane_id();
    const int prev = __builtin_amdgcn_update_dpp(key, key, 0x138, 0xF, 0xF, false);   // wave_shr:1 (lane 0 keeps its own)
    const int next = __builtin_amdgcn_update_dpp(key, key, 0x130, 0xF, 0xF, false);   // wave_shl:1 (lane 63 keeps its own)
    const bool tail = (l == 63 || next != key);
    float m = (l == 0 || prev != key) ? 0.0f : 1.0f;      // 1 while no segment start lies between the source lane and this one
    float v0 = val[0], v1 = val[1], v2 = val[2], v3 = val[3], v4 = val[4], v5 = val[5], v6 = val[6], v7 = val[7], v8 = val[8];
#define FPCDR_SCAN_STEP(CTRL)                           \
    "v_fmac_f32_dpp %0, %0, %9 " CTRL "\n\t"            \
    "v_fmac_f32_dpp %1, %1, %9 " CTRL "\n\t"            \
    "v_fmac_f32_dpp %2, %2, %9 " CTRL "\n\t"            \
    "v_fmac_f32_dpp %3, %3, %9 " CTRL "\n\t"            \
    "v_fmac_f32_dpp %4, %4, %9 " CTRL "\n\t"            \
    "v_fmac_f32_dpp %5, %5, %9 " CTRL "\n\t"            \
    "v_fmac_f32_dpp %6, %6, %9 " CTRL "\n\t"            \
    "v_fmac_f32_dpp %7, %7, %9 " CTRL "\n\t"            \
    "v_fmac_f32_dpp %8, %8, %9 " CTRL "\n\t"            \
    "v_mul_f32_dpp %9, %9, %9 " CTRL "\n\t"             \
    "s_nop 1\n\t"
    asm volatile("s_nop 1\n\t"        // (a DPP read needs two wait states after the VALU write of its source)
                 FPCDR_SCAN_STEP("row_shr:1 row_mask:0xf bank_mask:0xf")
                 FPCDR_SCAN_STEP("row_shr:2 row_mask:0xf bank_mask:0xf")
                 FPCDR_SCAN_STEP("row_shr:4 row_mask:0xf bank_mask:0xf")
                 FPCDR_SCAN_STEP("row_shr:8 row_mask:0xf bank_mask:0xf")
                 FPCDR_SCAN_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf")
                 FPCDR_SCAN_STEP("row_bcast:31 row_mask:0xc bank_mask:0xf")
                 : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7), "+v"(v8), "+v"(m));
#undef FPCDR_SCAN_STEP
    if (tail && key >= 0) {
        const float v[9] = {v0, v1, v2, v3, v4, v5, v6, v7, v8};
        emit(key, v);
    }
}

// ---- per-workgroup vertex-gradient table in LDS (shared by the backward kernels) ------------------------------------
// Scattered global f32 atomics retire slowly (one per run tail tripled the time of rasterize backward), so a workgroup sums
// the (x, y, w) gradients of its vertices in a small open-addressed LDS table first and flushes every slot once.
#define FPCDR_VT_SLOTS_N 256
constexpr int FPCDR_VT_SLOTS = FPCDR_VT_SLOTS_N;      // (a power of two)
struct VTable {
    int *key;             // [FPCDR_VT_SLOTS], -1 = free
    double (*acc)[3];     // [FPCDR_VT_SLOTS][3] = (x, y, w) sums in double (lds_add_f64)
};
__device__ __forceinline__ void vtable_init(const VTable &t, int tid, int nthreads) {
    for (int k = tid; k < FPCDR_VT_SLOTS; k += nthreads) {
        t.key[k] = -1;
        t.acc[k][0] = 0.0; t.acc[k][1] = 0.0; t.acc[k][2] = 0.0;
    }
}
// add the nine sums of one triangle run (three vertices x (x, y, w)); gp = the image's grad_pos rows
__device__ __forceinline__ void vtable_add(const VTable &t, float *gp, const int (&vk)[3], const float (&sm)[9]) {
    unsigned int slot[3];
    int old[3];
#pragma unroll
    // (r5, measured and dropped: slot = key & (SLOTS - 1), so that neighbouring vertex indices flush as neighbouring 16-byte pieces of
    //  one 32-byte sector: k_shade 1 591 -> 1 651 us at cfg3, 1 698 with 512 slots -- the clustered keys cost the adds more probes than
    //  the flush saves sectors)
    for (int kk = 0; kk < 3; ++kk) slot[kk] = (((unsigned int)vk[kk] * 2654435761u) >> 16) & (FPCDR_VT_SLOTS - 1);
#pragma unroll
    for (int kk = 0; kk < 3; ++kk) old[kk] = atomicCAS(&t.key[slot[kk]], -1, vk[kk]);     // three claims in flight
#pragma unroll
    for (int kk = 0; kk < 3; ++kk) {
        const int key = vk[kk];
        bool done = (old[kk] == -1 || old[kk] == key);
        for (int probe = 1; probe < FPCDR_VT_SLOTS && !done; ++probe) {
            slot[kk] = (slot[kk] + 1) & (FPCDR_VT_SLOTS - 1);
            const int o = atomicCAS(&t.key[slot[kk]], -1, key);
            done = (o == -1 || o == key);
        }
        if (done) {
            lds_add_f64(&t.acc[slot[kk]][0], sm[3 * kk]);
            lds_add_f64(&t.acc[slot[kk]][1], sm[3 * kk + 1]);
            lds_add_f64(&t.acc[slot[kk]][2], sm[3 * kk + 2]);
        } else {   // table full: straight to memory
            atomicAdd(gp + 4 * (size_t)key + 0, sm[3 * kk]); atomicAdd(gp + 4 * (size_t)key + 1, sm[3 * kk + 1]);
            atomicAdd(gp + 4 * (size_t)key + 3, sm[3 * kk + 2]);
        }
    }
}
// lane = (slot, component): the four dwords of a vertex are one contiguous 16-byte access
__device__ __forceinline__ void vtable_flush(const VTable &t, float *gp, int tid, int nthreads) {
    for (int k = tid; k < FPCDR_VT_SLOTS * 4; k += nthreads) {
        const int slot = k >> 2, comp = k & 3;
        const int key = t.key[slot];
        if (key >= 0 && comp != 2) {      // (x, y, -, w): z receives no gradient
            const float v = (float)t.acc[slot][comp == 3 ? 2 : comp];
            if (v != 0.0f) atomicAdd(gp + 4 * (size_t)key + comp, v);
        }
    }
}

// Full-wave integer min / max with DPP (same structure as wave_sum_dpp); result in every lane.
__device__ __forceinline__ int wave_min_dpp(int v) {
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xF, 0xF, false));
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xF, 0xF, false));
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x141, 0xF, 0xF, false));
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x140, 0xF, 0xF, false));
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x142, 0xA, 0xF, false));
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x143, 0xC, 0xF, false));
    return __builtin_amdgcn_readlane(v, 63);
}

// __syncthreads_or for four-wave workgroups without its DPP reduction (the library's is ~27 vector instructions): every wave votes
// (one v_cmp into a scalar pair), its first lane stores the verdict in the wave's OWN slot -- nothing to initialise --, and behind the
// barrier everybody reads the four slots.  `slots`: four ints of LDS that nobody else touches between this call and the next barrier.
__device__ __forceinline__ bool block4_any(int *slots, bool pred) {
    const unsigned long long bal = __builtin_amdgcn_ballot_w64(pred);
    if ((threadIdx.x & 63) == 0) slots[threadIdx.x >> 6] = bal != 0ull ? 1 : 0;
    __syncthreads();
    const int4 v = *reinterpret_cast<const int4 *>(slots);
    return __builtin_amdgcn_readfirstlane(v.x | v.y | v.z | v.w) != 0;
}

// ds_min_i32 / ds_max_i32 on an LDS word from whichever lanes are active, as ONE instruction each.  (atomicMin / atomicMax on LDS go
// through the compiler's atomic optimiser, which first reduces over the wave -- ten vector instructions per call -- although the
// callers here have already reduced and a single lane publishes.)  The trailing wait makes the update visible to a following barrier.
__device__ __forceinline__ void lds_minmax4(int *p_min0, int v_min0, int *p_max0, int v_max0, int *p_min1, int v_min1, int *p_max1, int v_max1) {
    typedef __attribute__((address_space(3))) int lds_int;
    asm volatile("ds_min_i32 %0, %1\n\tds_max_i32 %2, %3\n\tds_min_i32 %4, %5\n\tds_max_i32 %6, %7\n\ts_waitcnt lgkmcnt(0)"
                 :: "v"((lds_int *)p_min0), "v"(v_min0), "v"((lds_int *)p_max0), "v"(v_max0), "v"((lds_int *)p_min1), "v"(v_min1),
                    "v"((lds_int *)p_max1), "v"(v_max1) : "memory");
}

// Four full-wave integer minima at once, written out as v_min_i32_dpp (one instruction per value and step: the permuted neighbour is
// an operand of the minimum itself, where update_dpp + min is a v_mov_b32_dpp and a v_min_i32).  The four chains are interleaved, so
// the two wait states a DPP read needs after the VALU write of its source are filled with the other three values' instructions: 24
// vector instructions for four reductions instead of ~53.  The results are valid in LANE 63 ONLY (the last row of the last broadcast
// step): callers let that lane publish them.  Maxima: pass ~x and invert the result.
__device__ __forceinline__ void wave_min4_dpp_lane63(int &a, int &b, int &c, int &d) {
#define FPCDR_MIN4_STEP(CTRL)                          \
    "v_min_i32_dpp %0, %0, %0 " CTRL "\n\t"            \
    "v_min_i32_dpp %1, %1, %1 " CTRL "\n\t"            \
    "v_min_i32_dpp %2, %2, %2 " CTRL "\n\t"            \
    "v_min_i32_dpp %3, %3, %3 " CTRL "\n\t"
    asm volatile("s_nop 1\n\t"
                 FPCDR_MIN4_STEP("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
                 FPCDR_MIN4_STEP("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
                 FPCDR_MIN4_STEP("row_half_mirror row_mask:0xf bank_mask:0xf")
                 FPCDR_MIN4_STEP("row_mirror row_mask:0xf bank_mask:0xf")
                 FPCDR_MIN4_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf")
                 FPCDR_MIN4_STEP("row_bcast:31 row_mask:0xc bank_mask:0xf")
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
#undef FPCDR_MIN4_STEP
}

// (An LDS-atomic variant of the per-group sums -- leaders store, members ds_add_f32 into a wave-private scratch --
// was measured 75 % SLOWER than the DPP reductions above: same-address LDS float atomics serialise badly.)
