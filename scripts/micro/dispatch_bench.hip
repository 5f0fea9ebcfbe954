// build: hipcc -O3 --offload-arch=gfx950 -Wno-unused-value dispatch_bench.hip -o /tmp/dispatch_bench
// measured on MI355X (r1): 588 k empty workgroups 0.124 ms, with one 2-byte load each 0.127 ms, 1/8 of them 0.017 ms;
// 2048 persistent workgroups pulling the same bins from ONE global atomic counter: 6.7 ms (do not do that).
// How long does the GPU take to dispatch a grid of (60, 34, 288) workgroups of 256 threads that leave at once, or after
// one 2-byte load?  (the cost of a bin-per-workgroup launch whose bins are mostly inactive)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void __launch_bounds__(256) k_empty(int *sink) { if (sink == (int *)1) *sink = 0; }
__global__ void __launch_bounds__(256) k_load(const uint16_t *occ, int *sink, int nx, int ny) {
    const uint16_t w = occ[((size_t)blockIdx.z * ny + blockIdx.y) * nx + blockIdx.x];
    if (w == 0) return;
    if (threadIdx.x == 0) atomicAdd(sink, 1);
}
__global__ void __launch_bounds__(256) k_persist(const uint16_t *occ, int *sink, int n, int *counter) {
    // persistent: each workgroup pulls bins from a counter
    __shared__ int s_i;
    for (;;) {
        if (threadIdx.x == 0) s_i = atomicAdd(counter, 1);
        __syncthreads();
        const int i = s_i;
        __syncthreads();
        if (i >= n) return;
        if (occ[i] != 0 && threadIdx.x == 0) atomicAdd(sink, 1);
    }
}
int main() {
    const int nx = 60, ny = 34, nb = 288, n = nx * ny * nb;
    uint16_t *occ; int *sink, *counter;
    hipMalloc(&occ, n * 2); hipMemset(occ, 0, n * 2); hipMalloc(&sink, 4); hipMemset(sink, 0, 4); hipMalloc(&counter, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0); for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k_empty, dim3(nx, ny, nb), dim3(256), 0, 0, sink);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); printf("empty        %.3f ms / launch\n", ms / 10);
        hipEventRecord(e0); for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k_load, dim3(nx, ny, nb), dim3(256), 0, 0, occ, sink, nx, ny);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); printf("one load     %.3f ms / launch\n", ms / 10);
        hipEventRecord(e0); for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k_empty, dim3(nx * ny * nb / 8), dim3(256), 0, 0, sink);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); printf("empty / 8    %.3f ms / launch\n", ms / 10);
        hipEventRecord(e0);
        for (int i = 0; i < 10; ++i) { hipMemsetAsync(counter, 0, 4, 0); hipLaunchKernelGGL(k_persist, dim3(2048), dim3(256), 0, 0, occ, sink, n, counter); }
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); printf("persistent   %.3f ms / launch\n", ms / 10);
    }
    return 0;
}
