// Adam update of ALL parameter tensors of the fit in one launch, with the reference's whole-tensor quaternion division folded
// in (include/fpcdr.h, fpcdr_adam_step; reference fit.py:493-505 ten parameter groups, :610-618 step + renormalisation).
// torch's fused Adam launches once per parameter group (different learning rates) plus a step-counter kernel each, and the two
// quaternion divisions are four small launches each: ~24 launches of ~5 us in the serial tail of a 5 ms step.
#include "common.h"

namespace {

// the arithmetic of torch.optim.Adam (amsgrad = False, weight_decay = 0, maximize = False):
//   m += (1 - b1) (g - m);  v = b2 v + (1 - b2) g g;  p -= step_size m / (sqrt(v) / sqrt(bc2) + eps),  step_size = lr / bc1
__device__ __forceinline__ void adam_one(float &p, float g, float &m, float &v, float omb1, float b2, float omb2, float eps,
                                         float step_size, float bc2_sqrt) {
    m = m + omb1 * (g - m);
    v = b2 * v + omb2 * (g * g);
    const float denom = sqrtf(v) / bc2_sqrt + eps;
    p = p - step_size * (m / denom);
}

__global__ void __launch_bounds__(256) k_adam_step(fpcdr_adam_params P) {
    // a step whose gradients are invalid (fpcdr_objective_params.skip_out, summed over the ranks): nothing is touched -- neither parameters
    // nor moments nor the quaternion division -- and the launch counts it.  (Every workgroup reads the counter only on the other branch.)
    if (P.skip_flag && P.skip_flag[0] != 0.0f) {
        if (P.skipped && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) P.skipped[0] = P.skipped[0] + 1;
        return;
    }
    fpcdr_adam_tensor t = P.t[blockIdx.y];
    const long long n = t.n;
    if (P.step_table) { t.step_size = P.step_table[2 * t.table_row]; t.bc2_sqrt = P.step_table[2 * t.table_row + 1]; }
    else if (P.skipped && t.grad) {
        // the host's step counters and schedule ran on through the skipped steps: the update of the run that never drew them
        const int s = P.skipped[0];      // (uniform)
        if (s > 0) {
            const double ns = (double)(t.step - s);
            t.step_size = (float)((double)t.lr * pow(P.lr_skip_gain, (double)s) / (1.0 - pow((double)P.beta1, ns)));
            t.bc2_sqrt = (float)sqrt(1.0 - pow((double)P.beta2, ns));
        }
    }
    const float step_size = t.step_size;
    if (!t.renorm) {
        if (!t.grad) return;
        const long long n4 = (n & 3) == 0 && ((((size_t)t.param | (size_t)t.grad | (size_t)t.exp_avg | (size_t)t.exp_avg_sq) & 15) == 0) ? n / 4 : 0;
        float4 *p4 = reinterpret_cast<float4 *>(t.param);
        const float4 *g4 = reinterpret_cast<const float4 *>(t.grad);
        float4 *m4 = reinterpret_cast<float4 *>(t.exp_avg), *v4 = reinterpret_cast<float4 *>(t.exp_avg_sq);
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
            float4 p = p4[i], m = m4[i], v = v4[i];
            const float4 g = g4[i];
            adam_one(p.x, g.x, m.x, v.x, P.one_minus_beta1, P.beta2, P.one_minus_beta2, P.eps, step_size, t.bc2_sqrt);
            adam_one(p.y, g.y, m.y, v.y, P.one_minus_beta1, P.beta2, P.one_minus_beta2, P.eps, step_size, t.bc2_sqrt);
            adam_one(p.z, g.z, m.z, v.z, P.one_minus_beta1, P.beta2, P.one_minus_beta2, P.eps, step_size, t.bc2_sqrt);
            adam_one(p.w, g.w, m.w, v.w, P.one_minus_beta1, P.beta2, P.one_minus_beta2, P.eps, step_size, t.bc2_sqrt);
            p4[i] = p; m4[i] = m; v4[i] = v;
        }
        for (long long i = 4 * n4 + (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
            float p = t.param[i], m = t.exp_avg[i], v = t.exp_avg_sq[i];
            adam_one(p, t.grad[i], m, v, P.one_minus_beta1, P.beta2, P.one_minus_beta2, P.eps, step_size, t.bc2_sqrt);
            t.param[i] = p; t.exp_avg[i] = m; t.exp_avg_sq[i] = v;
        }
        return;
    }
    // a quaternion tensor (quirk Q3: divided by the norm of the WHOLE tensor, fit.py:616-618): one block updates it, reduces
    // the sum of squares of the updated values and divides.  These tensors hold 36 and 4 F floats.
    if (blockIdx.x != 0) return;
    __shared__ float s_part[4];
    float ss = 0.0f;
    for (long long i = threadIdx.x; i < n; i += 256) {
        float p = t.param[i];
        if (t.grad) {
            float m = t.exp_avg[i], v = t.exp_avg_sq[i];
            adam_one(p, t.grad[i], m, v, P.one_minus_beta1, P.beta2, P.one_minus_beta2, P.eps, step_size, t.bc2_sqrt);
            t.exp_avg[i] = m; t.exp_avg_sq[i] = v;
            t.param[i] = p;
        }
        ss += p * p;
    }
    ss = wave_sum_dpp(ss);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = ss;
    __syncthreads();
    const float norm = sqrtf(s_part[0] + s_part[1] + s_part[2] + s_part[3]);
    for (long long i = threadIdx.x; i < n; i += 256) t.param[i] = t.param[i] / norm;
}

}  // namespace

extern "C" int fpcdr_adam_step(const fpcdr_adam_params *p, void *stream) {
    FPCDR_REQUIRE(p != nullptr, "null params");
    FPCDR_REQUIRE(p->n_tensors >= 0 && p->n_tensors <= FPCDR_ADAM_MAX_TENSORS, "too many tensors for one call");
    if (p->n_tensors == 0) return FPCDR_OK;
    FPCDR_REQUIRE(!p->skipped || p->lr_skip_gain > 0.0, "skipped: lr_skip_gain must be positive (1 for a constant learning rate)");
    long long nmax = 0;
    for (int i = 0; i < p->n_tensors; ++i) {
        const fpcdr_adam_tensor &t = p->t[i];
        FPCDR_REQUIRE(t.param != nullptr && t.n > 0, "null parameter tensor");
        FPCDR_REQUIRE(t.grad == nullptr || (t.exp_avg && t.exp_avg_sq), "a tensor with a gradient needs its two moment buffers");
        FPCDR_REQUIRE(t.grad != nullptr || t.renorm, "a tensor without a gradient has nothing to do");
        FPCDR_REQUIRE(t.grad == nullptr || p->step_table || t.bc2_sqrt > 0.0f, "the bias correction must be positive");
        FPCDR_REQUIRE(!p->step_table || (t.table_row >= 0 && t.table_row < FPCDR_ADAM_MAX_TENSORS), "table_row outside the table");
        FPCDR_REQUIRE(!p->skipped || p->step_table || t.grad == nullptr || (t.step >= 1 && t.lr >= 0.0f), "skipped: every tensor needs its step count and learning rate");
        if (!t.renorm && t.n > nmax) nmax = t.n;
    }
    const int bx = (int)fpcdr_cdiv(fpcdr_cdiv(nmax > 0 ? nmax : 1, 4), 256);
    hipLaunchKernelGGL(k_adam_step, dim3(bx < 1 ? 1 : (bx > 1024 ? 1024 : bx), p->n_tensors), dim3(256), 0, (hipStream_t)stream, *p);
    FPCDR_CHECK_LAUNCH();
    return FPCDR_OK;
}
