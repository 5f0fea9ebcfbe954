// transform_clip for a minibatch, forward and backward, for gfx950 (MI355X).
//
// Performs the work of reference src/torch/camera.py:11-23 (`posw @ mvp.T` on homogeneous vertices) for all
// B = F x Nc images of a step at once: image b = f * Nc + c uses vertex buffer f and matrix b.  The reference (and
// fpc_diffrend_amd.camera.transform_clip) leave this to torch.matmul; at [288,15002,4] x [4,4] the batched GEMM
// picked by the BLAS library costs 0.3 ms forward and 0.6 ms backward for 69 MB of traffic, so the fit loop uses
// these streaming kernels instead (the views of a frame read the same vertices through L2).
#include "common.h"

namespace {

__global__ void __launch_bounds__(256) k_clip_fwd(const float *__restrict__ mvp, const float *__restrict__ verts, int V, int Nc,
                                                  float4 *__restrict__ out) {
    const int b = blockIdx.y, f = b / Nc;
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    const float *m = mvp + (size_t)b * 16;   // wave-uniform: scalar loads
    const float *p = verts + ((size_t)f * V + v) * 3;
    const float x = p[0], y = p[1], z = p[2];
    float4 o;
    o.x = x * m[0] + y * m[1] + z * m[2] + m[3];
    o.y = x * m[4] + y * m[5] + z * m[6] + m[7];
    o.z = x * m[8] + y * m[9] + z * m[10] + m[11];
    o.w = x * m[12] + y * m[13] + z * m[14] + m[15];
    out[(size_t)b * V + v] = o;
}

// d/d verts: thread = (frame, vertex), sums over the frame's Nc views (no atomics)
__global__ void __launch_bounds__(256) k_clip_bwd_verts(const float *__restrict__ mvp, const float4 *__restrict__ g, int V, int Nc,
                                                        float *__restrict__ g_verts) {
    const int f = blockIdx.y;
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    float gx = 0.f, gy = 0.f, gz = 0.f;
    for (int c = 0; c < Nc; ++c) {
        const int b = f * Nc + c;
        const float *m = mvp + (size_t)b * 16;
        const float4 q = g[(size_t)b * V + v];
        gx += q.x * m[0] + q.y * m[4] + q.z * m[8] + q.w * m[12];
        gy += q.x * m[1] + q.y * m[5] + q.z * m[9] + q.w * m[13];
        gz += q.x * m[2] + q.y * m[6] + q.z * m[10] + q.w * m[14];
    }
    float *o = g_verts + ((size_t)f * V + v) * 3;
    o[0] = gx; o[1] = gy; o[2] = gz;
}

// d/d mvp[b][i][j] = sum_v g[b][v][i] * (x,y,z,1)[j]: wave DPP sums, then 16 atomics per wave
__global__ void __launch_bounds__(256) k_clip_bwd_mvp(const float *__restrict__ verts, const float4 *__restrict__ g, int V, int Nc,
                                                      float *__restrict__ g_mvp) {
    const int b = blockIdx.y, f = b / Nc;
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
    float pw[4] = {0.f, 0.f, 0.f, 0.f};
    if (v < V) {
        q = g[(size_t)b * V + v];
        const float *p = verts + ((size_t)f * V + v) * 3;
        pw[0] = p[0]; pw[1] = p[1]; pw[2] = p[2]; pw[3] = 1.0f;
    }
    const float gi[4] = {q.x, q.y, q.z, q.w};
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float s = wave_sum_dpp(gi[i] * pw[j]);
            if (lane == 0 && s != 0.0f) atomicAdd(g_mvp + (size_t)b * 16 + i * 4 + j, s);
        }
}

// Uniform-Laplacian gather over a static, padded one-ring table (reference regulariser, fit.py:581 via pytorch3d):
//   mode 0 (forward):   out[f][v] = inv_deg[v] * sum_n x[f][n] - x[f][v]            (L x,   L = D^-1 A - I)
//   mode 1 (backward):  out[f][v] = sum_n inv_deg[n] * x[f][n] - x[f][v]            (L^T x, L^T = A D^-1 - I)
// nbr [V,D] holds neighbour indices, entries >= V are padding.  Both directions are gathers: no atomics.
__global__ void __launch_bounds__(256) k_lap_gather(const float *__restrict__ x, const int32_t *__restrict__ nbr,
                                                    const float *__restrict__ inv_deg, int V, int D, int mode,
                                                    float *__restrict__ out) {
    const int f = blockIdx.y;
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    const float *xf = x + (size_t)f * V * 3;
    float sx = 0.f, sy = 0.f, sz = 0.f;
    for (int d = 0; d < D; ++d) {
        const int n = nbr[(size_t)v * D + d];
        if (n < V) {
            const float w = mode ? inv_deg[n] : 1.0f;
            sx += w * xf[3 * n]; sy += w * xf[3 * n + 1]; sz += w * xf[3 * n + 2];
        }
    }
    const float s = mode ? 1.0f : inv_deg[v];
    float *o = out + ((size_t)f * V + v) * 3;
    o[0] = s * sx - xf[3 * v]; o[1] = s * sy - xf[3 * v + 1]; o[2] = s * sz - xf[3 * v + 2];
}

}  // namespace

extern "C" int fpcdr_laplacian_gather(const float *x, const int32_t *nbr, const float *inv_deg, float *out, int32_t F, int32_t V,
                                      int32_t D, int32_t transpose, void *stream) {
    FPCDR_REQUIRE(x && nbr && inv_deg && out, "null pointer");
    FPCDR_REQUIRE(F > 0 && V > 0 && D > 0 && F <= 65535, "bad sizes");
    hipLaunchKernelGGL(k_lap_gather, dim3(fpcdr_cdiv(V, 256), F), dim3(256), 0, (hipStream_t)stream, x, nbr, inv_deg, V, D,
                       transpose ? 1 : 0, out);
    FPCDR_CHECK_LAUNCH();
    return FPCDR_OK;
}

extern "C" int fpcdr_transform_clip_fwd(const float *mvp, const float *verts, float *out, int32_t F, int32_t Nc, int32_t V,
                                        void *stream) {
    FPCDR_REQUIRE(mvp && verts && out, "null pointer");
    FPCDR_REQUIRE(F > 0 && Nc > 0 && V > 0 && (long long)F * Nc <= 65535, "bad sizes");
    hipLaunchKernelGGL(k_clip_fwd, dim3(fpcdr_cdiv(V, 256), F * Nc), dim3(256), 0, (hipStream_t)stream, mvp, verts, V, Nc,
                       (float4 *)out);
    FPCDR_CHECK_LAUNCH();
    return FPCDR_OK;
}

extern "C" int fpcdr_transform_clip_bwd(const float *mvp, const float *verts, const float *grad_out, float *grad_verts,
                                        float *grad_mvp, int32_t F, int32_t Nc, int32_t V, void *stream) {
    FPCDR_REQUIRE(mvp && verts && grad_out, "null pointer");
    FPCDR_REQUIRE(F > 0 && Nc > 0 && V > 0 && (long long)F * Nc <= 65535, "bad sizes");
    hipStream_t st = (hipStream_t)stream;
    if (grad_verts)
        hipLaunchKernelGGL(k_clip_bwd_verts, dim3(fpcdr_cdiv(V, 256), F), dim3(256), 0, st, mvp, (const float4 *)grad_out, V, Nc,
                           grad_verts);
    if (grad_mvp)
        hipLaunchKernelGGL(k_clip_bwd_mvp, dim3(fpcdr_cdiv(V, 256), F * Nc), dim3(256), 0, st, verts, (const float4 *)grad_out, V,
                           Nc, grad_mvp);
    FPCDR_CHECK_LAUNCH();
    return FPCDR_OK;
}
