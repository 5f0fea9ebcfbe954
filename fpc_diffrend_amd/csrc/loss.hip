// Background composite + L2 pixel loss, forward and gradient in ONE pass, for gfx950 (MI355X).
//
// Performs the work of reference src/torch/fit.py:161 (`torch.where(rast_out[..., 3:] > 0, colour,
// 45/255)`) and the pixel term of fit.py:579 (`torch.mean((ref - colour*255) ** 2)`), which the
// reference runs as ~6 eager elementwise kernels plus their autograd mirrors (~100 B/px of HBM
// traffic).  Because the pixel loss is the root of the graph its gradient is known in closed form, so
// one streaming pass reads colour (4C B/px), the coverage channel of rast and the 8-bit reference image
// and writes d loss / d colour (4C B/px); the sum of squares is reduced wave -> block -> one f64 atomic.
#include "common.h"

#include <algorithm>

namespace {

__global__ void __launch_bounds__(256) k_pixel_loss(const float *__restrict__ color, const float4 *__restrict__ rast,
                                                    const uint8_t *__restrict__ ref, long long npix, int C, float bg,
                                                    float color_scale, float grad_scale, double *__restrict__ loss_sum,
                                                    float *__restrict__ grad_color) {
    __shared__ float s_part[4];
    float acc = 0.0f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (long long)gridDim.x * blockDim.x) {
        const bool covered = rast[i].w > 0.0f;
        const float r = (float)ref[i];
        for (int c = 0; c < C; ++c) {
            const float col = covered ? color[i * C + c] : bg;
            const float d = r - col * color_scale;
            acc += d * d;
            if (grad_color) grad_color[i * C + c] = covered ? (-2.0f * color_scale * grad_scale) * d : 0.0f;
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(loss_sum, (double)s_part[0] + (double)s_part[1] + (double)s_part[2] + (double)s_part[3]);
}

// value of the pixel objective from the loss slots of fpcdr_render_loss_fwd / fpcdr_aa_loss_fwd: one wave, one launch (the same
// arithmetic as five eager torch kernels -- sum, scale, add, divide, cast -- took 30 us between the forward and the backward call)
__global__ void __launch_bounds__(64) k_objective_value(const double *__restrict__ slots, int n, const double *__restrict__ bg_sumsq,
                                                        double bg_coeff, double n_total, float *__restrict__ out) {
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 64) acc += slots[i];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (threadIdx.x == 0) {
        if (bg_sumsq) acc = acc + bg_coeff * bg_sumsq[0];
        out[0] = (float)(acc / n_total);
    }
}

// sum over one image of (ref - bg_scaled)^2: grid (chunks, images), 16 pixels per thread and trip
__global__ void __launch_bounds__(256) k_ref_bg_sumsq(const uint8_t *__restrict__ ref, long long px, float bgs,
                                                      double *__restrict__ out) {
    __shared__ double s_part[4];
    const uint8_t *r = ref + (size_t)blockIdx.y * px;
    double acc = 0.0;
    const long long nvec = ((size_t)r % 16 == 0) ? px / 16 : 0;   // 16-byte loads when the image base is aligned
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long long)gridDim.x * 256) {
        const uint4 v = ((const uint4 *)r)[i];
        const unsigned int w[4] = {v.x, v.y, v.z, v.w};
        float s = 0.0f;
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float d = (float)((w[k] >> (8 * j)) & 255u) - bgs;
                s += d * d;
            }
        acc += (double)s;
    }
    for (long long i = nvec * 16 + (long long)blockIdx.x * 256 + threadIdx.x; i < px; i += (long long)gridDim.x * 256) {
        const float d = (float)r[i] - bgs;
        acc += (double)(d * d);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out + blockIdx.y, s_part[0] + s_part[1] + s_part[2] + s_part[3]);
}

}  // namespace

extern "C" int fpcdr_objective_value(const double *loss_slots, int32_t n_slots, const double *bg_sumsq, double bg_coeff,
                                     double n_total, float *out, void *stream) {
    FPCDR_REQUIRE(loss_slots && out, "null pointer");
    FPCDR_REQUIRE(n_slots > 0 && n_total > 0.0, "bad sizes");
    hipLaunchKernelGGL(k_objective_value, dim3(1), dim3(64), 0, (hipStream_t)stream, loss_slots, n_slots, bg_sumsq, bg_coeff, n_total, out);
    FPCDR_CHECK_LAUNCH();
    return FPCDR_OK;
}

extern "C" int fpcdr_pixel_loss(const fpcdr_pixel_loss_params *p, void *stream) {
    FPCDR_REQUIRE(p != nullptr, "null params");
    FPCDR_REQUIRE(p->color && p->rast && p->ref && p->loss_sum, "null pointer");
    FPCDR_REQUIRE(p->B > 0 && p->H > 0 && p->W > 0 && p->C > 0, "sizes must be positive");
    const long long npix = (long long)p->B * p->H * p->W;
    long long g = (npix + 255) / 256;
    const int grid = (int)(g > 8192 ? 8192 : g);
    hipLaunchKernelGGL(k_pixel_loss, dim3(grid), dim3(256), 0, (hipStream_t)stream, p->color, (const float4 *)p->rast, p->ref,
                       npix, p->C, p->bg, p->color_scale, p->grad_scale, p->loss_sum, p->grad_color);
    FPCDR_CHECK_LAUNCH();
    return FPCDR_OK;
}

extern "C" int fpcdr_ref_bg_sumsq(const uint8_t *ref, int64_t n_images, int64_t px_per_image, float bg_scaled, double *out,
                                  void *stream) {
    FPCDR_REQUIRE(ref && out, "null pointer");
    FPCDR_REQUIRE(n_images > 0 && n_images <= 65535 && px_per_image > 0, "bad sizes");
    const int chunks = (int)std::min<long long>(64, (px_per_image + 4095) / 4096);
    hipLaunchKernelGGL(k_ref_bg_sumsq, dim3(chunks, (unsigned)n_images), dim3(256), 0, (hipStream_t)stream, ref,
                       (long long)px_per_image, bg_scaled, out);
    FPCDR_CHECK_LAUNCH();
    return FPCDR_OK;
}

