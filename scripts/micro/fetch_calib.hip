// FETCH_SIZE / WRITE_SIZE calibration on gfx950: stream a buffer of known size with loads of 1, 2, 4, 8 and 16 bytes per lane
// (coalesced), with 16-byte GATHERS (random float4 from a 32 MB table, the shape of the rasteriser's vertex fetches) and 4-byte
// gathers, and compare rocprofv3's byte counts with the bytes actually touched.
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/fetch_calib.hip -o /tmp/fetch_calib
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- /tmp/fetch_calib      (and once more with WRITE_SIZE)
// MI355X_MICROARCH.md: FETCH_SIZE counts 16-byte-per-lane streams at exactly half; other widths are "uncalibrated" -- this is
// the calibration the objective kernels' traffic figures use (profiles/r02_fetch_calibration.txt).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

template <typename T>
__global__ void __launch_bounds__(256) k_stream(const T *__restrict__ src, size_t n, unsigned long long *sink) {
    unsigned long long acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const T v = src[i];
        const unsigned char *b = reinterpret_cast<const unsigned char *>(&v);
        for (int k = 0; k < (int)sizeof(T); ++k) acc += b[k];
    }
    if (acc == 0x123456789abcdefull) *sink = acc;
}
template <typename T>
__global__ void __launch_bounds__(256) k_gather(const T *__restrict__ table, size_t n_table, const uint32_t *__restrict__ idx, size_t n,
                                                unsigned long long *sink) {
    unsigned long long acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const T v = table[idx[i] % n_table];
        const unsigned char *b = reinterpret_cast<const unsigned char *>(&v);
        for (int k = 0; k < (int)sizeof(T); ++k) acc += b[k];
    }
    if (acc == 0x123456789abcdefull) *sink = acc;
}
template <typename T>
__global__ void __launch_bounds__(256) k_fill(T *__restrict__ dst, size_t n, T v) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = v;
}
__global__ void __launch_bounds__(256) k_atomic(float *__restrict__ dst, size_t n_dst, const uint32_t *__restrict__ idx, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) atomicAdd(dst + idx[i] % n_dst, 1.0f);
}

struct B2 { uint16_t a; };
struct B8 { uint32_t a, b; };
struct B16 { uint32_t a, b, c, d; };

int main() {
    const size_t bytes = (size_t)1 << 30;   // 1 GiB streamed per kernel (4x the Infinity Cache)
    void *buf, *big;
    unsigned long long *sink;
    hipMalloc(&buf, bytes);
    hipMalloc(&big, bytes);
    hipMalloc(&sink, 8);
    hipMemset(buf, 1, bytes);
    uint32_t *idx;
    const size_t n_idx = (size_t)64 << 20;   // 64 M gathers
    hipMalloc(&idx, n_idx * 4);
    {
        uint32_t *h = (uint32_t *)malloc(n_idx * 4);
        uint32_t s = 12345u;
        for (size_t i = 0; i < n_idx; ++i) { s = s * 1664525u + 1013904223u; h[i] = s >> 3; }
        hipMemcpy(idx, h, n_idx * 4, hipMemcpyHostToDevice);
        free(h);
    }
    const int grid = 256 * 16;
    for (int rep = 0; rep < 2; ++rep) {
        hipMemset(big, rep, bytes);          // evict
        hipLaunchKernelGGL(k_stream<uint8_t>, dim3(grid), dim3(256), 0, 0, (const uint8_t *)buf, bytes, sink);
        hipLaunchKernelGGL(k_stream<B2>, dim3(grid), dim3(256), 0, 0, (const B2 *)buf, bytes / 2, sink);
        hipLaunchKernelGGL(k_stream<uint32_t>, dim3(grid), dim3(256), 0, 0, (const uint32_t *)buf, bytes / 4, sink);
        hipLaunchKernelGGL(k_stream<B8>, dim3(grid), dim3(256), 0, 0, (const B8 *)buf, bytes / 8, sink);
        hipLaunchKernelGGL(k_stream<B16>, dim3(grid), dim3(256), 0, 0, (const B16 *)buf, bytes / 16, sink);
        // gathers from a 32 MB table (L2 / Infinity-Cache resident, like the vertex and texture tables) and from the whole 1 GiB
        hipLaunchKernelGGL(k_gather<B16>, dim3(grid), dim3(256), 0, 0, (const B16 *)buf, ((size_t)32 << 20) / 16, idx, n_idx, sink);
        hipLaunchKernelGGL(k_gather<uint32_t>, dim3(grid), dim3(256), 0, 0, (const uint32_t *)buf, ((size_t)32 << 20) / 4, idx, n_idx, sink);
        hipLaunchKernelGGL(k_gather<B16>, dim3(grid), dim3(256), 0, 0, (const B16 *)buf, bytes / 16, idx, n_idx, sink);
        hipLaunchKernelGGL(k_fill<uint8_t>, dim3(grid), dim3(256), 0, 0, (uint8_t *)big, bytes, (uint8_t)3);
        hipLaunchKernelGGL(k_fill<uint32_t>, dim3(grid), dim3(256), 0, 0, (uint32_t *)big, bytes / 4, 7u);
        hipLaunchKernelGGL(k_fill<B16>, dim3(grid), dim3(256), 0, 0, (B16 *)big, bytes / 16, B16{1, 2, 3, 4});
        hipLaunchKernelGGL(k_atomic, dim3(grid), dim3(256), 0, 0, (float *)big, ((size_t)4 << 20) / 4, idx, n_idx);   // 4 MB target (a texture)
    }
    hipDeviceSynchronize();
    printf("streamed %zu bytes per k_stream / k_fill launch; %zu gathers / atomics per gather launch (idx stream: %zu bytes, 4 B/lane)\n", bytes,
           n_idx, n_idx * 4);
    return 0;
}
