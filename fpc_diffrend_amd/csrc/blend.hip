// Blendshape combine on the f32 matrix cores of gfx950 (MI355X).
//
// Performs the contraction of reference src/torch/fit.py:115-122 (`blend`, prior mode:
// V = v_base + B (M2 (M1 e_f))), :58-62 (`blend_free`: V = v_base + m3 (m2 (m1 e_f))) and :88-99
// (`blend_combined`), batched over the frames of one optimisation step: the reference evaluates one
// frame per iteration with a one-hot e_f; here all F frames of the step go through the matrix cores at
// once,   out[F,M] = v_base[M] + w[F,K] . Bmat[M,K]^T     (M = 3V).
//
// v_mfma_f32_32x32x2_f32 (exact f32 products and accumulation, MI355X_MICROARCH: 64 FLOP/clk/SIMD).
// The contraction is tiny next to the pixel work (2 M K F = 0.43 GFLOP at M = 45k, K = 150, F = 32)
// and is bounded by ONE read of Bmat (27 MB), not by MFMA rate -- utilisation is reported honestly
// as low.  Operand roles are chosen so that the D tile has vertex coordinates on the lane axis: every
// accumulator register is stored as two 128-byte row segments.
//
// k index trick: the sum over k is order independent, so lane half h (= lane >> 5) of each MFMA
// k-step takes k from its own contiguous range; each lane then walks a contiguous run of Bmat / w.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int KC = 64;  // k values per lane half per chunk (register-resident Bmat fragment)

// D[i = frame][j = coord]: reg r of lane l holds frame (r&3) + 8 (r>>2) + 4 (l>>5), coord l&31
__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// out[f][i] = v_base[i] + sum_k Bmat[i][k] w[f][k];  one wave per 32-coordinate tile, loops over frame tiles
__global__ void __launch_bounds__(256) k_blend_fwd(const float *__restrict__ v_base, const float *__restrict__ Bmat,
                                                   const float *__restrict__ w, float *__restrict__ out, int M, int K, int F) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tile = blockIdx.x * 4 + wave;
    const int i0 = tile * 32;
    if (i0 >= M) return;
    const int col = lane & 31, h = lane >> 5;
    const int row = min(i0 + col, M - 1);  // clamped rows are computed but never stored
    const float *brow = Bmat + (size_t)row * K;
    for (int f0 = 0; f0 < F; f0 += 32) {
        const int fr = min(f0 + col, F - 1);
        const float *wrow = w + (size_t)fr * K;
        f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        if ((K & 1) == 0) {
            // even K: rows are 8-byte aligned, two k per load (the lanes of a wave walk 64 different rows, so the number
            // of load instructions, not bytes, is what this loop waits for)
            for (int kc = 0; kc < K; kc += 2 * KC) {
                const int kb = kc + h * KC;
#pragma unroll 8
                for (int s = 0; s < KC; s += 2) {
                    const int k = kb + s;
                    const bool ok = k < K;
                    const float2 a = ok ? *reinterpret_cast<const float2 *>(wrow + k) : make_float2(0.f, 0.f);
                    const float2 b = ok ? *reinterpret_cast<const float2 *>(brow + k) : make_float2(0.f, 0.f);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
                }
            }
        } else {
            for (int kc = 0; kc < K; kc += 2 * KC) {
                const int kb = kc + h * KC;
#pragma unroll 8
                for (int s = 0; s < KC; ++s) {
                    const int k = kb + s;
                    const bool ok = k < K;
                    const float a = ok ? wrow[k] : 0.0f;   // A[i = frame][k]
                    const float b = ok ? brow[k] : 0.0f;   // B[k][j = coord]
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
                }
            }
        }
        const int i = i0 + col;
        if (i < M) {
            const float vb = v_base ? v_base[i] : 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int f = f0 + acc_row(r, lane);
                if (f < F) out[(size_t)f * M + i] = acc[r] + vb;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// LDS-staged forward (K <= LDS_KMAX): one workgroup per 32-coordinate tile.  The tile's rows are CONTIGUOUS in Bmat
// (32 K floats), so the workgroup streams them with aligned 16-byte loads -- every 128-byte line once, whole -- instead of
// 64 lanes walking 64 different rows 8 bytes at a time (52 us for the 27 MB of the 30k rig; this form: 32 us by HIP events, of
// which 20 us remain with the global loads compiled out: LDS scatter, barriers and 75 MFMAs of 64 cycles per tile).  Rows go to LDS with
// a stride of KQ + 4 floats (== 4 mod 32: 16-byte reads of 8 consecutive rows cover all 32 banks); the four waves split the
// k range, each lane half walks its own contiguous run with ds_read_b128 feeding four MFMA k-steps, and the four partial
// 32 x 32 tiles are summed through LDS.
constexpr int LDS_KMAX = 224;
constexpr int TM = 32;

__global__ void __launch_bounds__(256) k_blend_fwd_lds(const float *__restrict__ v_base, const float *__restrict__ Bmat,
                                                       const float *__restrict__ w, float *__restrict__ out, int M, int K, int F,
                                                       int KQ) {
    extern __shared__ float smem[];
    const int KP = KQ + 4;
    float *s_b = smem;                    // [32][KP] Bmat tile, zero beyond K
    float *s_w = smem + TM * KP;          // [32][KP] weights of the current 32 frames
    float *s_red = s_w;                   // [3][16][64] partial accumulators of waves 1..3 (aliases s_w once it is consumed)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 31, h = lane >> 5;
    const int i0 = blockIdx.x * TM;
    // (row, k) of a flat element index advance by a fixed step per trip: two divisions per thread instead of one per element
    const int step4_r = 1024 / K, step4_k = 1024 - step4_r * K;     // +1024 elements (256 float4)
    // zero the pad columns [K, KP) of both tiles (the copies below never touch them)
    {
        const int np = KP - K;
        for (int e = tid; e < 32 * np; e += 256) {
            const int r = e / np, k = K + e - r * np;
            s_b[r * KP + k] = 0.0f;
            s_w[r * KP + k] = 0.0f;
        }
    }
    // Both tiles are contiguous runs of 32 K floats, 16-byte aligned (tile bytes = 128 K).  ALL of a thread's loads are issued
    // before the first one is consumed (NQ4 = 7 >= 32 K / 4 / 256 for K <= 224): one memory latency per tile, not one per trip
    // (a load -> LDS-store loop of 19 trips made this kernel latency-bound at 35 us).
    constexpr int NQ4 = (LDS_KMAX * TM / 4 + 255) / 256;
    const int n4 = (TM * K) / 4;          // 32 K is a multiple of 4
    auto load_tile = [&](const float *__restrict__ src, long long avail, float4 (&v)[NQ4]) {
        // src: tile start; avail: floats that may be read from src (the rest of the tile is zero)
#pragma unroll
        for (int j = 0; j < NQ4; ++j) {
            const int q = tid + 256 * j;
            const long long e0 = 4ll * q;
            v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (q < n4) {
                if (e0 + 3 < avail) v[j] = *reinterpret_cast<const float4 *>(src + e0);
                else {
                    if (e0 < avail) v[j].x = src[e0];
                    if (e0 + 1 < avail) v[j].y = src[e0 + 1];
                    if (e0 + 2 < avail) v[j].z = src[e0 + 2];
                }
            }
        }
    };
    auto store_tile = [&](float *__restrict__ dst, const float4 (&v)[NQ4]) {
        int r = (4 * tid) / K, k = 4 * tid - r * K;
#pragma unroll
        for (int j = 0; j < NQ4; ++j) {
            if (tid + 256 * j < n4) {
                const float vv[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
                int rr = r, kk = k;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    dst[rr * KP + kk] = vv[c];
                    if (++kk == K) { kk = 0; ++rr; }
                }
            }
            r += step4_r; k += step4_k;
            if (k >= K) { k -= K; ++r; }
        }
    };
    float4 vb4[NQ4], vw4[NQ4];
    load_tile(Bmat + (size_t)i0 * K, (long long)(M - i0) * K, vb4);
    load_tile(w, (long long)F * K, vw4);
    store_tile(s_b, vb4);
    const int kw = KQ / 4, kh = KQ / 8;               // k per wave, per lane half (a multiple of 4)
    const int kbase = wave * kw + h * kh;
    for (int f0 = 0; f0 < F; f0 += 32) {
        if (f0 > 0) {
            __syncthreads();                          // previous frame tile (and its s_red) consumed
            load_tile(w + (size_t)f0 * K, (long long)(F - f0) * K, vw4);
            // s_red overwrote the pad columns of s_w with partial sums: zero them again (they meet the zero pads of s_b, but
            // 0 * x is 0 only for finite x)
            const int np = KP - K;
            for (int e = tid; e < 32 * np; e += 256) {
                const int r = e / np;
                s_w[r * KP + K + e - r * np] = 0.0f;
            }
        }
        store_tile(s_w, vw4);
        __syncthreads();
        f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        const float *pa = s_w + col * KP + kbase, *pb = s_b + col * KP + kbase;
        for (int s = 0; s < kh; s += 4) {
            const float4 a = *reinterpret_cast<const float4 *>(pa + s);
            const float4 b = *reinterpret_cast<const float4 *>(pb + s);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
        }
        __syncthreads();                              // every wave has read its part of s_w: s_red may overwrite it
        if (wave > 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s_red[((wave - 1) * 16 + r) * 64 + lane] = acc[r];
        }
        __syncthreads();
        if (wave == 0) {
            const int i = i0 + col;
            const float vb = (v_base && i < M) ? v_base[i] : 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = acc[r] + s_red[r * 64 + lane] + s_red[(16 + r) * 64 + lane] + s_red[(32 + r) * 64 + lane];
                const int f = f0 + acc_row(r, lane);
                if (f < F && i < M) out[(size_t)f * M + i] = v + vb;
            }
        }
    }
}

// grad_w[f][k] += sum_i gout[f][i] Bmat[i][k];  D[i = frame][j = k]; the long i reduction is split over
// blockIdx.y slabs of SLAB rows and finished with f32 atomics (few adders per address).
#define FPCDR_BLEND_SLAB 512
constexpr int SLAB = FPCDR_BLEND_SLAB;
__global__ void __launch_bounds__(256) k_blend_bwd_w(const float *__restrict__ Bmat, const float *__restrict__ gout,
                                                     float *__restrict__ grad_w, int M, int K, int F) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 31, h = lane >> 5;
    const int k0 = blockIdx.x * 32, f0 = blockIdx.z * 32;
    const int kk = k0 + col, fr = min(f0 + col, F - 1);
    // each wave of the block takes a quarter of the slab; lane half h a contiguous half of that
    const int rows_per_wave = SLAB / 4, rows_per_half = rows_per_wave / 2;
    const int base = blockIdx.y * SLAB + wave * rows_per_wave + h * rows_per_half;
    const float *grow = gout + (size_t)fr * M;
    f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if ((M & 1) == 0) {
        // even M: the frame's gradient row is 8-byte aligned at every even i -- two i per (uncoalesced) load of A
#pragma unroll 8
        for (int s = 0; s < rows_per_half; s += 2) {
            const int i = base + s;
            const bool ok = i < M;      // i even, M even: i + 1 < M as well
            const float2 a = ok ? *reinterpret_cast<const float2 *>(grow + i) : make_float2(0.f, 0.f);   // A[frame][i], [i + 1]
            const float b0 = (ok && kk < K) ? Bmat[(size_t)i * K + kk] : 0.0f;                           // B[i][k]
            const float b1 = (ok && kk < K) ? Bmat[(size_t)(i + 1) * K + kk] : 0.0f;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b1, acc, 0, 0, 0);
        }
    } else {
#pragma unroll 8
        for (int s = 0; s < rows_per_half; ++s) {
            const int i = base + s;
            const bool ok = i < M;
            const float a = ok ? grow[i] : 0.0f;                                  // A[frame][i]
            const float b = (ok && kk < K) ? Bmat[(size_t)i * K + kk] : 0.0f;     // B[i][k]
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
    }
    // the four waves' partial tiles summed through LDS first: one atomic per entry and WORKGROUP (the adders per address are what this
    // kernel waits for: 375 per address with one atomic per wave, 94 so)
    __shared__ float s_red[3][16][64];
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s_red[wave - 1][r][lane] = acc[r];
    }
    __syncthreads();
    if (wave == 0 && kk < K) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int f = f0 + acc_row(r, lane);
            const float v = acc[r] + s_red[0][r][lane] + s_red[1][r][lane] + s_red[2][r][lane];
            if (f < F && v != 0.0f) atomicAdd(grad_w + (size_t)f * K + kk, v);
        }
    }
}

// grad_B[i][k] = sum_f gout[f][i] w[f][k];  D[i = coord][j = k]
__global__ void __launch_bounds__(256) k_blend_bwd_basis(const float *__restrict__ w, const float *__restrict__ gout,
                                                         float *__restrict__ grad_B, int M, int K, int F) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 31, h = lane >> 5;
    const int i0 = (blockIdx.x * 4 + wave) * 32, k0 = blockIdx.y * 32;
    if (i0 >= M) return;
    const int ii = min(i0 + col, M - 1), kk = min(k0 + col, K - 1);
    f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const int half = (F + 1) / 2;
    for (int s = 0; s < half; ++s) {
        const int f = h * half + s;
        const bool ok = f < F && (h == 0 ? s < half : true);
        const float a = ok ? gout[(size_t)f * M + ii] : 0.0f;  // A[i = coord][f]
        const float b = ok ? w[(size_t)f * K + kk] : 0.0f;     // B[f][j = k]
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    const int k = k0 + col;
    if (k < K) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = i0 + acc_row(r, lane);
            if (i < M) grad_B[(size_t)i * K + k] = acc[r];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The rig's weight algebra (reference fit.py:115-116: maps_intermediate . (maps . one-hot frame); a batch selects columns):
//   w[fb][k] = sum_f mi[k][f] * maps[f][col(fb)],   col(fb) = cols ? cols[fb] : col0 + fb
// written in the [Fb, K] layout the blend kernel reads.  150 x 32 x 32 multiply-adds: as torch ops it is a GEMM, a transposing copy and --
// backward -- two GEMMs and two adds, 35 us of 5-9 us launches in the serial tail of a 2.8 ms step; here one launch each way.
// column of batch entry fb: negative indices count from the end (as maps[:, ids] does); -1 = outside maps (the torch form raises; a kernel
// cannot: the forward writes NaN into that row -- loud downstream, no out-of-bounds read -- and the backward gives it no gradient;
// fit.rig_weights(validate=True) checks on the host first)
__device__ __forceinline__ int rig_col(const int64_t *__restrict__ cols, int col0, int fb, int Fc) {
    if (!cols) return col0 + fb;
    long long c = cols[fb];
    if (c < 0) c += Fc;
    return (c >= 0 && c < Fc) ? (int)c : -1;
}

__global__ void __launch_bounds__(256) k_rig_weights_fwd(const float *__restrict__ mi, const float *__restrict__ maps,
                                                         const int64_t *__restrict__ cols, int col0, int K, int Fr, int Fc, int Fb,
                                                         float *__restrict__ w) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= Fb * K) return;
    const int fb = idx / K, k = idx - fb * K;
    const int c = rig_col(cols, col0, fb, Fc);
    if (c < 0) { w[idx] = __int_as_float(0x7fc00000); return; }
    float s = 0.0f;
#pragma unroll 8
    for (int f = 0; f < Fr; ++f) s += mi[(size_t)k * Fr + f] * maps[(size_t)f * Fc + c];
    w[idx] = s;
}

// g_mi[k][f] = sum_fb g_w[fb][k] * maps[f][col(fb)];   g_maps[f][c] = sum over the fb with col(fb) = c of sum_k mi[k][f] * g_w[fb][k]
// (both overwritten; columns no frame of the batch selects get zero).  ONE WAVE per output entry, the lanes over the sum's terms: a
// thread per entry walks up to Fb x K dependent load + multiply-add trips on its own (340 us at K = 150, Fb = 32; this form: a few).
__global__ void __launch_bounds__(256) k_rig_weights_bwd(const float *__restrict__ mi, const float *__restrict__ maps,
                                                         const int64_t *__restrict__ cols, int col0, const float *__restrict__ g_w,
                                                         int K, int Fr, int Fc, int Fb, float *__restrict__ g_mi, float *__restrict__ g_maps) {
    const int lane = threadIdx.x & 63;
    int idx = blockIdx.x * 4 + (threadIdx.x >> 6);      // (wave-uniform)
    if (idx < K * Fr) {
        if (!g_mi) return;
        const int k = idx / Fr, f = idx - k * Fr;
        float s = 0.0f;
        for (int fb = lane; fb < Fb; fb += 64) {
            const int c = rig_col(cols, col0, fb, Fc);
            if (c >= 0) s += g_w[(size_t)fb * K + k] * maps[(size_t)f * Fc + c];
        }
        s = wave_sum_dpp(s);
        if (lane == 0) g_mi[idx] = s;
        return;
    }
    idx -= K * Fr;
    if (idx >= Fr * Fc || !g_maps) return;
    const int f = idx / Fc, c = idx - f * Fc;
    float s = 0.0f;
    for (int fb = 0; fb < Fb; ++fb) {
        if (rig_col(cols, col0, fb, Fc) != c) continue;      // (uniform over the wave)
        for (int k = lane; k < K; k += 64) s += mi[(size_t)k * Fr + f] * g_w[(size_t)fb * K + k];
    }
    s = wave_sum_dpp(s);
    if (lane == 0) g_maps[idx] = s;
}

}  // namespace

extern "C" int fpcdr_blend_fwd(const float *v_base, const float *Bmat, const float *w, float *out, int32_t M, int32_t K,
                               int32_t F, void *stream) {
    FPCDR_REQUIRE(Bmat && w && out, "null pointer");
    FPCDR_REQUIRE(M > 0 && K > 0 && F > 0, "sizes must be positive");
    if (K <= LDS_KMAX && ((size_t)Bmat & 15) == 0 && ((size_t)w & 15) == 0) {
        const int KQ = (K + 31) / 32 * 32;
        const size_t tile = (size_t)32 * (KQ + 4);
        const size_t lds = (tile + (tile > 3 * 16 * 64 ? tile : (size_t)3 * 16 * 64)) * sizeof(float);   // s_b + max(s_w, s_red): <= 58 KB
        hipLaunchKernelGGL(k_blend_fwd_lds, dim3(fpcdr_cdiv(M, TM)), dim3(256), lds, (hipStream_t)stream, v_base, Bmat, w, out, M, K,
                           F, KQ);
        FPCDR_CHECK_LAUNCH();
        return FPCDR_OK;
    }
    hipLaunchKernelGGL(k_blend_fwd, dim3(fpcdr_cdiv(fpcdr_cdiv(M, 32), 4)), dim3(256), 0, (hipStream_t)stream, v_base, Bmat, w,
                       out, M, K, F);
    FPCDR_CHECK_LAUNCH();
    return FPCDR_OK;
}

extern "C" int fpcdr_blend_bwd_w(const float *Bmat, const float *grad_out, float *grad_w, int32_t M, int32_t K, int32_t F,
                                 void *stream) {
    FPCDR_REQUIRE(Bmat && grad_out && grad_w, "null pointer");
    FPCDR_REQUIRE(M > 0 && K > 0 && F > 0, "sizes must be positive");
    FPCDR_REQUIRE(fpcdr_cdiv(M, SLAB) <= 65535 && fpcdr_cdiv(F, 32) <= 65535, "problem too large for one launch");
    dim3 grid(fpcdr_cdiv(K, 32), fpcdr_cdiv(M, SLAB), fpcdr_cdiv(F, 32));
    hipLaunchKernelGGL(k_blend_bwd_w, grid, dim3(256), 0, (hipStream_t)stream, Bmat, grad_out, grad_w, M, K, F);
    FPCDR_CHECK_LAUNCH();
    return FPCDR_OK;
}

extern "C" int fpcdr_blend_bwd_basis(const float *w, const float *grad_out, float *grad_B, int32_t M, int32_t K, int32_t F,
                                     void *stream) {
    FPCDR_REQUIRE(w && grad_out && grad_B, "null pointer");
    FPCDR_REQUIRE(M > 0 && K > 0 && F > 0, "sizes must be positive");
    FPCDR_REQUIRE(fpcdr_cdiv(K, 32) <= 65535, "K too large for one launch");
    dim3 grid(fpcdr_cdiv(fpcdr_cdiv(M, 32), 4), fpcdr_cdiv(K, 32));
    hipLaunchKernelGGL(k_blend_bwd_basis, grid, dim3(256), 0, (hipStream_t)stream, w, grad_out, grad_B, M, K, F);
    FPCDR_CHECK_LAUNCH();
    return FPCDR_OK;
}

extern "C" int fpcdr_rig_weights_fwd(const float *mi, const float *maps, const int64_t *cols, int32_t col0, int32_t K, int32_t Fr, int32_t Fc,
                                     int32_t Fb, float *w, void *stream) {
    FPCDR_REQUIRE(mi && maps && w, "null pointer");
    FPCDR_REQUIRE(K > 0 && Fr > 0 && Fc > 0 && Fb > 0 && (long long)Fb * K < (1ll << 30), "bad sizes");
    FPCDR_REQUIRE(cols || (col0 >= 0 && col0 + Fb <= Fc), "column range outside maps");
    hipLaunchKernelGGL(k_rig_weights_fwd, dim3(fpcdr_cdiv(Fb * K, 256)), dim3(256), 0, (hipStream_t)stream, mi, maps, cols, col0, K, Fr, Fc, Fb, w);
    FPCDR_CHECK_LAUNCH();
    return FPCDR_OK;
}

extern "C" int fpcdr_rig_weights_bwd(const float *mi, const float *maps, const int64_t *cols, int32_t col0, const float *grad_w, int32_t K,
                                     int32_t Fr, int32_t Fc, int32_t Fb, float *grad_mi, float *grad_maps, void *stream) {
    FPCDR_REQUIRE(mi && maps && grad_w, "null pointer");
    FPCDR_REQUIRE(K > 0 && Fr > 0 && Fc > 0 && Fb > 0 && (long long)Fb * K < (1ll << 30) && (long long)K * Fr + (long long)Fr * Fc < (1ll << 30),
                  "bad sizes");
    FPCDR_REQUIRE(cols || (col0 >= 0 && col0 + Fb <= Fc), "column range outside maps");
    hipLaunchKernelGGL(k_rig_weights_bwd, dim3(fpcdr_cdiv(K * Fr + Fr * Fc, 4)), dim3(256), 0, (hipStream_t)stream, mi, maps, cols, col0,
                       grad_w, K, Fr, Fc, Fb, grad_mi, grad_maps);
    FPCDR_CHECK_LAUNCH();
    return FPCDR_OK;
}
