// Scattered float atomics into a 4 MB accumulator (a 1024^2 x 1 texture gradient) on gfx950: device-scope atomicAdd -- coherent across
// the eight XCDs, so performed at the memory side: 32 bytes of write traffic each -- against workgroup-scope atomics into a PRIVATE
// copy per XCD (indexed by the hardware XCC_ID the wave runs on), which stay in that XCD's L2, plus the 8-copy reduction afterwards.
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/atomic_scope_bench.hip -o scripts/micro/atomic_scope_bench && scripts/micro/atomic_scope_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

__device__ __forceinline__ int xcc_id() {
    // s_getreg_b32 HW_REG_XCC_ID (id 20 on gfx940+), bits [3:0]
    return (int)(__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 0xf);
}

// MODE 0: device scope, one copy.  MODE 1: workgroup scope, copy of the wave's XCD.  MODE 2: device scope, copy of the wave's XCD.
// pattern: every workgroup adds into a window of `span` consecutive texels (as a bin's texel window flush does) at a
// pseudo-random place; neighbouring workgroups overlap as neighbouring bins do.
template <int MODE>
__global__ void __launch_bounds__(256) k(float *acc, size_t n, int span, int per_thread, int *xcc_hist) {
    const int xc = xcc_id();
    if (threadIdx.x == 0 && xcc_hist) atomicAdd(&xcc_hist[xc * 8 + (blockIdx.x & 7)], 1);
    float *dst = acc + (MODE == 0 ? 0 : (size_t)xc * n);
    const unsigned int base = ((blockIdx.x * 2654435761u) >> 8) % (unsigned int)(n - span);
    for (int i = 0; i < per_thread; ++i) {
        const unsigned int j = base + (threadIdx.x + 256u * i) % span;
        if (MODE == 1) __hip_atomic_fetch_add(dst + j, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else atomicAdd(dst + j, 1.0f);
    }
}

__global__ void k_reduce8(const float *copies, float *out, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int c = 0; c < 8; ++c) s += copies[c * n + i];
    out[i] = s;
}

int main() {
    const size_t n = 1 << 20;
    const int wgs = 95556, span = 1156, per_thread = 4;      // ~ the occupied bins of cfg3 x ~880 touched texels each
    float *acc, *out;
    int *hist;
    hipMalloc(&acc, 8 * n * sizeof(float));
    hipMalloc(&out, n * sizeof(float));
    hipMalloc(&hist, 128 * sizeof(int));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<float> ref(n), got(n);
    for (int mode = 0; mode < 3; ++mode) {
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            hipMemset(acc, 0, 8 * n * sizeof(float));
            hipMemset(hist, 0, 128 * sizeof(int));
            hipDeviceSynchronize();
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(wgs), dim3(256), 0, 0, acc, n, span, per_thread, hist);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(wgs), dim3(256), 0, 0, acc, n, span, per_thread, hist);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(wgs), dim3(256), 0, 0, acc, n, span, per_thread, hist);
            if (mode != 0) hipLaunchKernelGGL(k_reduce8, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, acc, out, n);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        hipMemcpy(got.data(), mode == 0 ? acc : out, n * sizeof(float), hipMemcpyDeviceToHost);
        double sum = 0, diff = 0;
        for (size_t i = 0; i < n; ++i) sum += got[i];
        if (mode == 0) ref = got;
        else for (size_t i = 0; i < n; ++i) diff += fabs((double)got[i] - (double)ref[i]);
        const double natom = (double)wgs * 256 * per_thread;
        printf("mode %d (%s): %.3f ms for %.1f M atomics = %.1f G atomics/s; sum %.0f (expected %.0f), |diff to mode 0| %.0f\n", mode,
               mode == 0 ? "device scope, one copy" : (mode == 1 ? "workgroup scope, per-XCD copy + reduce" : "device scope, per-XCD copy + reduce"),
               best, natom / 1e6, natom / best / 1e6, sum, natom, diff);
    }
    int h[128];
    hipMemcpy(h, hist, sizeof(h), hipMemcpyDeviceToHost);
    printf("workgroups per XCC_ID (rows) by blockIdx & 7 (columns):\n");
    for (int x = 0; x < 16; ++x) {
        int tot = 0;
        for (int c = 0; c < 8; ++c) tot += h[x * 8 + c];
        if (!tot) continue;
        printf("  xcc %2d:", x);
        for (int c = 0; c < 8; ++c) printf(" %6d", h[x * 8 + c]);
        printf("\n");
    }
    return 0;
}
