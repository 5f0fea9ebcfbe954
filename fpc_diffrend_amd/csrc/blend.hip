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

// grad_w[f][k] += sum_i gout[f][i] Bmat[i][k];  D[i = frame][j = k]; the long i reduction is split over
// blockIdx.y slabs of SLAB rows and finished with f32 atomics (few adders per address).
constexpr int SLAB = 512;
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
    if (kk < K) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int f = f0 + acc_row(r, lane);
            if (f < F && acc[r] != 0.0f) atomicAdd(grad_w + (size_t)f * K + kk, acc[r]);
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

}  // namespace

extern "C" int fpcdr_blend_fwd(const float *v_base, const float *Bmat, const float *w, float *out, int32_t M, int32_t K,
                               int32_t F, void *stream) {
    FPCDR_REQUIRE(Bmat && w && out, "null pointer");
    FPCDR_REQUIRE(M > 0 && K > 0 && F > 0, "sizes must be positive");
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
