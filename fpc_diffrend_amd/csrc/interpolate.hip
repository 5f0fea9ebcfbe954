// interpolate forward / backward for gfx950 (MI355X).
//
// Performs the work of `dr.interpolate(uv[None], rast_out, uv_idx[, rast_db, diff_attrs='all'])` at
// reference src/torch/fit.py:154 and :157 (nvdiffrast op, absent from the reference tree).
//
// Pure streaming kernels, one pixel per lane, bounded by HBM:
//   fwd  reads rast (16 B/px, one float4 per lane, fully coalesced), gathers 3 x A attribute floats
//        from an L2-resident vertex table and writes A floats (+ 2*n_diff differentials);
//   bwd  reads dy + rast, writes grad_rast as one float4 per lane; the attribute gradient (only when
//        the caller asks for it -- the reference's uv carries no gradient, fit.py:431) is scattered
//        with wave-level pre-reduction by vertex before the f32 atomics.
#include "common.h"

namespace {

struct DiffIdx { int32_t v[FPCDR_MAX_ATTR]; };

constexpr int SROWS = 8;   // rows per thread of the streaming kernels (divides the 32-row hint bin)

template <int A_STATIC>
__global__ void __launch_bounds__(256) k_interp_fwd(const float *__restrict__ attr, const float4 *__restrict__ rast,
                                                    const int32_t *__restrict__ tri, const float4 *__restrict__ rast_db,
                                                    int H, int W, int B, int Ba, int Vt, int A_dyn, int T, int n_diff,
                                                    DiffIdx didx, float *__restrict__ out, float *__restrict__ out_da,
                                                    const uint8_t *__restrict__ hint) {
    // 256 x SROWS pixels per workgroup, grid (W / 256, H / SROWS, B): a thread owns SROWS vertically adjacent pixels and has all
    // their rast loads in flight at once (one pixel per thread made 2.5 M tiny workgroups whose lifetime -- one memory round
    // trip -- bounded the kernel at ~3.3 ms whatever it read).  The SROWS rows lie in one 32-row bin: one hint lookup.
    const int A = A_STATIC > 0 ? A_STATIC : A_dyn;
    const int px = blockIdx.x * 256 + threadIdx.x, y0 = blockIdx.y * SROWS, b = blockIdx.z;
    if (px >= W) return;
    // region hint: an empty bin holds rast = 0 -- its result is known without reading it
    const bool known_empty = hint && !fpcdr_hint_on(hint, 0, B, H, W, b, y0, px);
    float4 rr[SROWS];
#pragma unroll
    for (int k = 0; k < SROWS; ++k)
        rr[k] = (known_empty || y0 + k >= H) ? make_float4(0.f, 0.f, 0.f, 0.f) : rast[((long long)b * H + y0 + k) * W + px];
#pragma unroll
    for (int k = 0; k < SROWS; ++k) {
        const int py = y0 + k;
        if (py >= H) break;
        const long long i = ((long long)b * H + py) * W + px;
        const float4 r = rr[k];
        const int t = (int)r.w - 1;
        float *o = out + i * A;
        if (t < 0 || t >= T) {
            if (A_STATIC == 2) {
                *reinterpret_cast<float2 *>(o) = make_float2(0.f, 0.f);
            } else {
                for (int k2 = 0; k2 < A; ++k2) o[k2] = 0.f;
            }
            if (n_diff > 0) {
                float *od = out_da + i * 2 * n_diff;
                if (A_STATIC == 2 && n_diff == 2) {
                    *reinterpret_cast<float4 *>(od) = make_float4(0.f, 0.f, 0.f, 0.f);
                } else {
                    for (int k2 = 0; k2 < 2 * n_diff; ++k2) od[k2] = 0.f;
                }
            }
            continue;
        }
        const float *ab = attr + (Ba > 1 ? (size_t)b * Vt * A : 0);
        const int i0 = tri[3 * t], i1 = tri[3 * t + 1], i2 = tri[3 * t + 2];
        const float *a0 = ab + (size_t)i0 * A, *a1 = ab + (size_t)i1 * A, *a2 = ab + (size_t)i2 * A;
        const float u = r.x, v = r.y, w = 1.0f - u - v;
        if (A_STATIC == 2) {
            const float2 q0 = *reinterpret_cast<const float2 *>(a0);
            const float2 q1 = *reinterpret_cast<const float2 *>(a1);
            const float2 q2 = *reinterpret_cast<const float2 *>(a2);
            *reinterpret_cast<float2 *>(o) = make_float2(u * q0.x + v * q1.x + w * q2.x, u * q0.y + v * q1.y + w * q2.y);
            if (n_diff == 2) {
                const float4 d = rast_db[i];
                float e0[2] = {q0.x - q2.x, q0.y - q2.y}, e1[2] = {q1.x - q2.x, q1.y - q2.y};
                float r_[4];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int kd = didx.v[j];
                    const float f0 = kd == 0 ? e0[0] : e0[1], f1 = kd == 0 ? e1[0] : e1[1];
                    r_[2 * j] = d.x * f0 + d.z * f1;
                    r_[2 * j + 1] = d.y * f0 + d.w * f1;
                }
                *reinterpret_cast<float4 *>(out_da + i * 4) = make_float4(r_[0], r_[1], r_[2], r_[3]);
                continue;
            }
        } else {
            for (int k2 = 0; k2 < A; ++k2) o[k2] = u * a0[k2] + v * a1[k2] + w * a2[k2];
        }
        if (n_diff > 0) {
            const float4 d = rast_db[i];
            float *od = out_da + i * 2 * n_diff;
            for (int j = 0; j < n_diff; ++j) {
                const int kd = didx.v[j];
                const float e0 = a0[kd] - a2[kd], e1 = a1[kd] - a2[kd];
                od[2 * j] = d.x * e0 + d.z * e1;
                od[2 * j + 1] = d.y * e0 + d.w * e1;
            }
        }
    }
}

// Backward.  2-D launch (8x8 pixel tile per wave) so lanes of a wave tend to share triangles when
// the attribute gradient is scattered.
template <bool GRAD_ATTR>
__global__ void __launch_bounds__(256) k_interp_bwd(const float *__restrict__ attr, const float4 *__restrict__ rast,
                                                    const int32_t *__restrict__ tri, const float4 *__restrict__ rast_db,
                                                    const float *__restrict__ dy, const float *__restrict__ dda, int B,
                                                    int H, int W, int Ba, int Vt, int A, int T, int n_diff, DiffIdx didx,
                                                    float *__restrict__ grad_attr, float4 *__restrict__ grad_rast,
                                                    float4 *__restrict__ grad_rast_db, const uint8_t *__restrict__ hint) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // with an attribute gradient: 8x8 pixel tile per wave; without: 256 consecutive pixels of a row per block
    const int px = GRAD_ATTR ? blockIdx.x * 16 + (wave & 1) * 8 + (lane & 7) : blockIdx.x * 256 + threadIdx.x;
    const int py = GRAD_ATTR ? blockIdx.y * 16 + (wave >> 1) * 8 + (lane >> 3) : blockIdx.y;
    const int b = blockIdx.z;
    const bool inside = px < W && py < H;
    const size_t i = ((size_t)b * H + (inside ? py : 0)) * W + (inside ? px : 0);
    int t = -1;
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (inside && !(hint && !fpcdr_hint_on(hint, 0, B, H, W, b, py, px))) {   // (an empty bin: rast = 0, zero gradients)
        r = rast[i];
        t = (int)r.w - 1;
        if (t >= T) t = -1;
    }
    float4 gr = make_float4(0.f, 0.f, 0.f, 0.f), gdb = make_float4(0.f, 0.f, 0.f, 0.f);
    int i0 = -1, i1 = -1, i2 = -1;
    const float *ab = attr + (Ba > 1 ? (size_t)b * Vt * A : 0);
    float *gab = GRAD_ATTR ? grad_attr + (Ba > 1 ? (size_t)b * Vt * A : 0) : nullptr;
    float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
    if (t >= 0) {
        i0 = tri[3 * t]; i1 = tri[3 * t + 1]; i2 = tri[3 * t + 2];
        const float *a0 = ab + (size_t)i0 * A, *a1 = ab + (size_t)i1 * A, *a2 = ab + (size_t)i2 * A;
        const float *g = dy + i * A;
        float gu = 0.f, gv = 0.f;
        for (int k = 0; k < A; ++k) {
            const float gk = g[k], c2 = a2[k];
            gu += gk * (a0[k] - c2);
            gv += gk * (a1[k] - c2);
        }
        gr.x = gu; gr.y = gv;
        if (n_diff > 0) {
            d = rast_db[i];
            const float *gd = dda + i * 2 * n_diff;
            for (int j = 0; j < n_diff; ++j) {
                const int k = didx.v[j];
                const float e0 = a0[k] - a2[k], e1 = a1[k] - a2[k];
                const float gx = gd[2 * j], gy = gd[2 * j + 1];
                gdb.x += gx * e0; gdb.y += gy * e0; gdb.z += gx * e1; gdb.w += gy * e1;
            }
        }
    }
    if (inside) {
        grad_rast[i] = gr;
        if (n_diff > 0) grad_rast_db[i] = gdb;
    }
    if (GRAD_ATTR) {
        if (__ballot(t >= 0) == 0ull) return;
        const float u = r.x, v = r.y, w = 1.0f - u - v;
        // one attribute channel at a time: three group-reduced atomics per channel
        for (int k = 0; k < A; ++k) {
            float gk = (t >= 0) ? dy[i * A + k] : 0.f;
            float ge0 = 0.f, ge1 = 0.f;
            if (t >= 0 && n_diff > 0) {
                for (int j = 0; j < n_diff; ++j)
                    if (didx.v[j] == k) {
                        const float gx = dda[i * 2 * n_diff + 2 * j], gy = dda[i * 2 * n_diff + 2 * j + 1];
                        ge0 += gx * d.x + gy * d.y;
                        ge1 += gx * d.z + gy * d.w;
                    }
            }
            {
                float *const dst[1] = {gab + (size_t)max(i0, 0) * A + k};
                const float val[1] = {u * gk + ge0};
                wave_group_atomic_add<1>(i0, dst, val);
            }
            {
                float *const dst[1] = {gab + (size_t)max(i1, 0) * A + k};
                const float val[1] = {v * gk + ge1};
                wave_group_atomic_add<1>(i1, dst, val);
            }
            {
                float *const dst[1] = {gab + (size_t)max(i2, 0) * A + k};
                const float val[1] = {w * gk - ge0 - ge1};
                wave_group_atomic_add<1>(i2, dst, val);
            }
        }
    }
}

// Backward without an attribute gradient (the fit loop's case: uv carries none, fit.py:431): pure streaming, 256 x SROWS
// pixels per workgroup like the forward kernel.
__global__ void __launch_bounds__(256) k_interp_bwd_rows(const float *__restrict__ attr, const float4 *__restrict__ rast,
                                                         const int32_t *__restrict__ tri, const float4 *__restrict__ rast_db,
                                                         const float *__restrict__ dy, const float *__restrict__ dda, int B,
                                                         int H, int W, int Ba, int Vt, int A, int T, int n_diff, DiffIdx didx,
                                                         float4 *__restrict__ grad_rast, float4 *__restrict__ grad_rast_db,
                                                         const uint8_t *__restrict__ hint) {
    const int px = blockIdx.x * 256 + threadIdx.x, y0 = blockIdx.y * SROWS, b = blockIdx.z;
    if (px >= W) return;
    const bool known_empty = hint && !fpcdr_hint_on(hint, 0, B, H, W, b, y0, px);   // rast = 0 there: zero gradients, nothing read
    float4 rr[SROWS];
#pragma unroll
    for (int k = 0; k < SROWS; ++k)
        rr[k] = (known_empty || y0 + k >= H) ? make_float4(0.f, 0.f, 0.f, 0.f) : rast[((size_t)b * H + y0 + k) * W + px];
    const float *ab = attr + (Ba > 1 ? (size_t)b * Vt * A : 0);
#pragma unroll
    for (int k = 0; k < SROWS; ++k) {
        const int py = y0 + k;
        if (py >= H) break;
        const size_t i = ((size_t)b * H + py) * W + px;
        int t = (int)rr[k].w - 1;
        if (t >= T) t = -1;
        float4 gr = make_float4(0.f, 0.f, 0.f, 0.f), gdb = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t >= 0) {
            const int i0 = tri[3 * t], i1 = tri[3 * t + 1], i2 = tri[3 * t + 2];
            const float *a0 = ab + (size_t)i0 * A, *a1 = ab + (size_t)i1 * A, *a2 = ab + (size_t)i2 * A;
            const float *g = dy + i * A;
            float gu = 0.f, gv = 0.f;
            for (int c = 0; c < A; ++c) {
                const float gk = g[c], c2 = a2[c];
                gu += gk * (a0[c] - c2);
                gv += gk * (a1[c] - c2);
            }
            gr.x = gu; gr.y = gv;
            if (n_diff > 0) {
                const float *gd = dda + i * 2 * n_diff;
                for (int j = 0; j < n_diff; ++j) {
                    const int c = didx.v[j];
                    const float e0 = a0[c] - a2[c], e1 = a1[c] - a2[c];
                    const float gx = gd[2 * j], gy = gd[2 * j + 1];
                    gdb.x += gx * e0; gdb.y += gy * e0; gdb.z += gx * e1; gdb.w += gy * e1;
                }
            }
        }
        grad_rast[i] = gr;
        if (n_diff > 0) grad_rast_db[i] = gdb;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Bin-shaped fast paths for the shapes of the reference's render() (A = 2 texture coordinates, no pixel differentials,
// W % 4 == 0): one workgroup per 32 x 32-pixel bin, a thread owns FOUR horizontally adjacent pixels, so every store is 16
// bytes per lane (4- and 8-byte-per-lane stores reach ~1.2-1.7 TB/s on this chip, 16-byte ones 5 TB/s).  A bin the region
// hint calls empty is written without reading anything.
__device__ __forceinline__ float2 interp2(const float2 *__restrict__ ab, const int32_t *__restrict__ tri, float4 r, int T) {
    const int t = (int)r.w - 1;
    if (t < 0 || t >= T) return make_float2(0.f, 0.f);
    const float2 q0 = ab[tri[3 * t]], q1 = ab[tri[3 * t + 1]], q2 = ab[tri[3 * t + 2]];
    const float u = r.x, v = r.y, w = 1.0f - u - v;
    return make_float2(u * q0.x + v * q1.x + w * q2.x, u * q0.y + v * q1.y + w * q2.y);
}

__global__ void __launch_bounds__(256) k_interp_fwd_bin2(const float2 *__restrict__ attr, const float4 *__restrict__ rast,
                                                         const int32_t *__restrict__ tri, int H, int W, int B, int Ba, int Vt, int T,
                                                         float4 *__restrict__ out4, const uint8_t *__restrict__ hint) {
    const int tid = threadIdx.x, b = blockIdx.z;
    const int px = blockIdx.x * 32 + (tid & 7) * 4, py = blockIdx.y * 32 + (tid >> 3);
    if (px >= W || py >= H) return;
    const size_t i = ((size_t)b * H + py) * W + px;       // first of the thread's four pixels
    if (hint && !fpcdr_hint_on(hint, 0, B, H, W, b, py, px)) {
        out4[i / 2] = make_float4(0.f, 0.f, 0.f, 0.f);
        out4[i / 2 + 1] = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    const float4 r0 = rast[i], r1 = rast[i + 1], r2 = rast[i + 2], r3 = rast[i + 3];
    const float2 *ab = attr + (Ba > 1 ? (size_t)b * Vt : 0);
    const float2 o0 = interp2(ab, tri, r0, T), o1 = interp2(ab, tri, r1, T), o2 = interp2(ab, tri, r2, T), o3 = interp2(ab, tri, r3, T);
    out4[i / 2] = make_float4(o0.x, o0.y, o1.x, o1.y);
    out4[i / 2 + 1] = make_float4(o2.x, o2.y, o3.x, o3.y);
}

__global__ void __launch_bounds__(256) k_interp_bwd_bin2(const float2 *__restrict__ attr, const float4 *__restrict__ rast,
                                                         const int32_t *__restrict__ tri, const float2 *__restrict__ dy2, int H, int W,
                                                         int B, int Ba, int Vt, int T, float4 *__restrict__ grad_rast,
                                                         const uint8_t *__restrict__ hint) {
    // rast and grad_rast are 16 bytes per PIXEL: one pixel per lane and instruction is already the widest access, and 32 lanes
    // of a row segment make it 512 contiguous bytes.  A thread owns pixels (col, rowgroup + 8 k), k = 0..3.
    const int tid = threadIdx.x, b = blockIdx.z;
    const int px = blockIdx.x * 32 + (tid & 31), py0 = blockIdx.y * 32 + (tid >> 5);
    if (px >= W) return;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    if (hint && !fpcdr_hint_on(hint, 0, B, H, W, b, blockIdx.y * 32, blockIdx.x * 32)) {   // rast = 0 there: zero gradients, nothing read
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (py0 + 8 * k < H) grad_rast[((size_t)b * H + py0 + 8 * k) * W + px] = z;
        return;
    }
    // four pixels per thread, every stage of the dependent chain (rast -> triangle -> vertex indices -> attributes) issued for
    // all four before the next stage is consumed
    const float2 *ab = attr + (Ba > 1 ? (size_t)b * Vt : 0);
    size_t idx[4];
    bool in[4];
    float4 rr[4];
    float2 g[4];
    int t[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        in[k] = py0 + 8 * k < H;
        // (rows beyond the image re-read a row inside it: with H % 32 in 1..7 even py0 itself lies outside in the last bin row)
        idx[k] = ((size_t)b * H + (in[k] ? py0 + 8 * k : min(py0, H - 1))) * W + px;
        rr[k] = rast[idx[k]];
        g[k] = dy2[idx[k]];
    }
    int v0[4], v1[4], v2[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        t[k] = (int)rr[k].w - 1;
        if (t[k] < 0 || t[k] >= T) t[k] = -1;
    }
    if (!__builtin_amdgcn_readfirstlane(__ballot((t[0] & t[1] & t[2] & t[3]) >= 0) != 0ull)) {     // a wave of empty pixels (no hint given)
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (in[k]) grad_rast[idx[k]] = z;
        return;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int tt = max(t[k], 0);
        v0[k] = tri[3 * tt]; v1[k] = tri[3 * tt + 1]; v2[k] = tri[3 * tt + 2];
    }
    float2 q0[4], q1[4], q2[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { q0[k] = ab[v0[k]]; q1[k] = ab[v1[k]]; q2[k] = ab[v2[k]]; }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        // (same order of operations as k_interp_bwd: sum over the two channels of g_k (a_k - a2_k))
        float gu = 0.f, gv = 0.f;
        gu += g[k].x * (q0[k].x - q2[k].x); gv += g[k].x * (q1[k].x - q2[k].x);
        gu += g[k].y * (q0[k].y - q2[k].y); gv += g[k].y * (q1[k].y - q2[k].y);
        if (in[k]) grad_rast[idx[k]] = t[k] >= 0 ? make_float4(gu, gv, 0.f, 0.f) : z;
    }
}

}  // namespace

extern "C" int fpcdr_interpolate_fwd(const fpcdr_interpolate_fwd_params *p, void *stream) {
    FPCDR_REQUIRE(p != nullptr, "null params");
    FPCDR_REQUIRE(p->attr && p->rast && p->tri && p->out, "null pointer");
    FPCDR_REQUIRE(p->B > 0 && p->H > 0 && p->W > 0 && p->Vt > 0 && p->A > 0 && p->T > 0, "sizes must be positive");
    FPCDR_REQUIRE(p->Ba == 1 || p->Ba == p->B, "attr batch must be 1 or B");
    FPCDR_REQUIRE(p->n_diff >= 0 && p->n_diff <= FPCDR_MAX_ATTR, "n_diff out of range");
    FPCDR_REQUIRE(p->n_diff == 0 || (p->rast_db && p->out_da), "differentials need rast_db and out_da");
    DiffIdx di;
    for (int j = 0; j < FPCDR_MAX_ATTR; ++j) di.v[j] = j < p->n_diff ? p->diff_idx[j] : 0;
    for (int j = 0; j < p->n_diff; ++j) FPCDR_REQUIRE(di.v[j] >= 0 && di.v[j] < p->A, "diff_idx out of range");
    FPCDR_REQUIRE(p->B <= 65535 && p->H <= 65535, "image batch / height too large for one launch");
    dim3 grid(fpcdr_cdiv(p->W, 256), fpcdr_cdiv(p->H, SROWS), p->B);
    hipStream_t st = (hipStream_t)stream;
    if (p->A == 2 && p->n_diff == 0 && (p->W & 3) == 0 && (((size_t)p->out | (size_t)p->attr) & 15) == 0) {
        hipLaunchKernelGGL(k_interp_fwd_bin2, dim3(fpcdr_cdiv(p->W, 32), fpcdr_cdiv(p->H, 32), p->B), dim3(256), 0, st,
                           (const float2 *)p->attr, (const float4 *)p->rast, p->tri, p->H, p->W, p->B, p->Ba, p->Vt, p->T, (float4 *)p->out,
                           p->hint);
        FPCDR_CHECK_LAUNCH();
        return FPCDR_OK;
    }
    if (p->A == 2)
        hipLaunchKernelGGL(k_interp_fwd<2>, grid, dim3(256), 0, st, p->attr, (const float4 *)p->rast, p->tri,
                           (const float4 *)p->rast_db, p->H, p->W, p->B, p->Ba, p->Vt, p->A, p->T, p->n_diff, di, p->out, p->out_da,
                           p->hint);
    else
        hipLaunchKernelGGL(k_interp_fwd<0>, grid, dim3(256), 0, st, p->attr, (const float4 *)p->rast, p->tri,
                           (const float4 *)p->rast_db, p->H, p->W, p->B, p->Ba, p->Vt, p->A, p->T, p->n_diff, di, p->out, p->out_da,
                           p->hint);
    FPCDR_CHECK_LAUNCH();
    return FPCDR_OK;
}

extern "C" int fpcdr_interpolate_bwd(const fpcdr_interpolate_bwd_params *p, void *stream) {
    FPCDR_REQUIRE(p != nullptr, "null params");
    FPCDR_REQUIRE(p->attr && p->rast && p->tri && p->dy && p->grad_rast, "null pointer");
    FPCDR_REQUIRE(p->B > 0 && p->H > 0 && p->W > 0 && p->Vt > 0 && p->A > 0 && p->T > 0, "sizes must be positive");
    FPCDR_REQUIRE(p->B <= 65535, "more than 65535 images per call");
    FPCDR_REQUIRE(p->Ba == 1 || p->Ba == p->B, "attr batch must be 1 or B");
    FPCDR_REQUIRE(p->n_diff >= 0 && p->n_diff <= FPCDR_MAX_ATTR, "n_diff out of range");
    FPCDR_REQUIRE(p->n_diff == 0 || (p->rast_db && p->dda && p->grad_rast_db), "differentials need rast_db, dda, grad_rast_db");
    DiffIdx di;
    for (int j = 0; j < FPCDR_MAX_ATTR; ++j) di.v[j] = j < p->n_diff ? p->diff_idx[j] : 0;
    for (int j = 0; j < p->n_diff; ++j) FPCDR_REQUIRE(di.v[j] >= 0 && di.v[j] < p->A, "diff_idx out of range");
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(fpcdr_cdiv(p->W, 16), fpcdr_cdiv(p->H, 16), p->B);
    dim3 grid_rows(fpcdr_cdiv(p->W, 256), fpcdr_cdiv(p->H, SROWS), p->B);
    if (!p->grad_attr && p->A == 2 && p->n_diff == 0 && (p->W & 3) == 0 && (((size_t)p->dy | (size_t)p->attr) & 15) == 0) {
        hipLaunchKernelGGL(k_interp_bwd_bin2, dim3(fpcdr_cdiv(p->W, 32), fpcdr_cdiv(p->H, 32), p->B), dim3(256), 0, st,
                           (const float2 *)p->attr, (const float4 *)p->rast, p->tri, (const float2 *)p->dy, p->H, p->W, p->B, p->Ba, p->Vt,
                           p->T, (float4 *)p->grad_rast, p->hint);
        FPCDR_CHECK_LAUNCH();
        return FPCDR_OK;
    }
    if (p->grad_attr)
        hipLaunchKernelGGL(k_interp_bwd<true>, grid, dim3(256), 0, st, p->attr, (const float4 *)p->rast, p->tri,
                           (const float4 *)p->rast_db, p->dy, p->dda, p->B, p->H, p->W, p->Ba, p->Vt, p->A, p->T, p->n_diff, di,
                           p->grad_attr, (float4 *)p->grad_rast, (float4 *)p->grad_rast_db, p->hint);
    else
        hipLaunchKernelGGL(k_interp_bwd_rows, grid_rows, dim3(256), 0, st, p->attr, (const float4 *)p->rast, p->tri,
                           (const float4 *)p->rast_db, p->dy, p->dda, p->B, p->H, p->W, p->Ba, p->Vt, p->A, p->T, p->n_diff, di,
                           (float4 *)p->grad_rast, (float4 *)p->grad_rast_db, p->hint);
    FPCDR_CHECK_LAUNCH();
    return FPCDR_OK;
}
