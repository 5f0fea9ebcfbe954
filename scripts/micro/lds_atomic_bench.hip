// LDS float-atomic throughput on gfx950: cycles per ds_add_f32 wave-instruction for the access shapes of k_render_aa_bwd
// (texel window: ~64 distinct consecutive addresses; vertex table: a dozen active lanes, some sharing an address).
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/lds_atomic_bench.hip -o scripts/micro/lds_atomic_bench && scripts/micro/lds_atomic_bench
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int MODE>
__global__ void __launch_bounds__(256) k(float *out, int iters, long long *cycles) {
    __shared__ float s[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) s[i] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int idx;
    bool active = true;
    if (MODE == 0) idx = wave * 1024 + lane;                       // 64 consecutive addresses
    else if (MODE == 1) idx = wave * 1024 + lane * 2;              // stride 2 (2-way bank conflict)
    else if (MODE == 2) { idx = wave * 1024 + lane; active = lane < 32; }          // 32 active lanes
    else if (MODE == 3) { idx = wave * 1024 + lane; active = (lane % 5) == 0; }    // 13 scattered active lanes
    else if (MODE == 4) idx = wave * 1024 + (lane >> 1);           // pairs of lanes share an address
    else if (MODE == 5) idx = wave * 1024 + (lane >> 3);           // 8 lanes share an address
    else if (MODE == 6) idx = wave * 1024 + ((lane * 37) & 1023);  // scattered
    else idx = wave * 1024;                                        // all 64 lanes one address
    const long long t0 = clock64();
    float v = 1.0f;
    for (int it = 0; it < iters; ++it) {
        if (active) atomicAdd(&s[idx], v);
        v += 1.0f;
        idx ^= (it & 1) << 6;                                      // keep the compiler from merging the adds
    }
    __syncthreads();
    const long long t1 = clock64();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    out[blockIdx.x * 256 + threadIdx.x] = s[threadIdx.x];
}

// other LDS atomics on 64 consecutive addresses: OP 0 int add, 1 u64 min, 2 float add with return, 3 32-bit compare-and-swap,
// 4 plain store (reference), 5 float add on 16 lanes only
template <int OP>
__global__ void __launch_bounds__(256) k2(float *out, int iters, long long *cycles) {
    __shared__ unsigned long long s64[1024];
    __shared__ int s32[2048];
    __shared__ float sf[2048];
    __shared__ double sd[1024];
    for (int i = threadIdx.x; i < 1024; i += 256) { s64[i] = ~0ull; sd[i] = 0.0; }
    for (int i = threadIdx.x; i < 2048; i += 256) { s32[i] = 0; sf[i] = 0.f; }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int idx = wave * 128 + lane;
    const long long t0 = clock64();
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        if (OP == 0) atomicAdd(&s32[idx], it);
        else if (OP == 1) atomicMin(&s64[idx], (unsigned long long)(0xffffffffu - it) << 8 | lane);
        else if (OP == 2) acc += atomicAdd(&sf[idx], 1.0f);
        else if (OP == 3) acc += (float)atomicCAS(&s32[idx], it, it + 1);
        else if (OP == 4) sf[idx] = (float)it;
        else if (OP == 5) { if (lane < 16) atomicAdd(&sf[idx], 1.0f); }
        else if (OP == 6) atomicAdd(&sd[idx], 1.0);                                   // double add (ds_add_f64)
        else if (OP == 7) atomicAdd(reinterpret_cast<unsigned long long *>(&sd[idx]), (unsigned long long)it);   // 64-bit integer add
        else if (OP == 8) atomicMax(&sf[idx], (float)it);                                 // float max
        else if (OP == 9) atomicAdd(&sd[wave * 128 + (lane >> 3) + ((it & 1) << 6)], 1.0);   // double add, 8 lanes share an address
        else if (OP == 10) atomicAdd(&sd[wave * 128 + ((lane * 5) & 63) * 2 % 128], 1.0);   // double add, scattered
        else if (OP == 11) { if ((lane % 7) == 0) atomicAdd(&sd[idx], 1.0); }               // double add, 10 scattered active lanes
        idx ^= (it & 1) << 6;
    }
    __syncthreads();
    const long long t1 = clock64();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    out[blockIdx.x * 256 + threadIdx.x] = acc + sf[threadIdx.x] + (float)s32[threadIdx.x] + (float)s64[threadIdx.x] + (float)sd[threadIdx.x];
}

template <int OP>
void run2(const char *name, float *out, long long *cyc, int iters) {
    hipLaunchKernelGGL(k2<OP>, dim3(256), dim3(256), 0, 0, out, iters, cyc);
    hipDeviceSynchronize();
    long long h[256];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double mean = 0;
    for (int i = 0; i < 256; ++i) mean += (double)h[i];
    mean /= 256;
    printf("%-44s %8.1f cycles per iteration of 4 waves  = %6.1f per wave-instruction\n", name, mean / iters, mean / iters / 4);
}

template <int MODE>
void run(const char *name, float *out, long long *cyc, int iters) {
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, out, iters, cyc);
    hipDeviceSynchronize();
    long long h[256];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double mean = 0;
    for (int i = 0; i < 256; ++i) mean += (double)h[i];
    mean /= 256;
    // 4 waves per workgroup issue concurrently into ONE LDS unit: cycles per wave-instruction as seen by the LDS = total / (4 iters)
    printf("%-44s %8.1f cycles per iteration of 4 waves  = %6.1f per wave-instruction\n", name, mean / iters, mean / iters / 4);
}

int main() {
    float *out; long long *cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
    const int iters = 4096;
    run<0>("64 consecutive addresses", out, cyc, iters);
    run<1>("stride 2", out, cyc, iters);
    run<2>("32 active lanes, consecutive", out, cyc, iters);
    run<3>("13 scattered active lanes", out, cyc, iters);
    run<4>("pairs of lanes share an address", out, cyc, iters);
    run<5>("8 lanes share an address", out, cyc, iters);
    run<6>("64 scattered addresses", out, cyc, iters);
    run<7>("all lanes one address", out, cyc, iters);
    run2<0>("int add, 64 consecutive", out, cyc, iters);
    run2<1>("u64 min, 64 consecutive", out, cyc, iters);
    run2<2>("float add with return", out, cyc, iters);
    run2<3>("32-bit compare-and-swap (returns)", out, cyc, iters);
    run2<4>("plain 4-byte store", out, cyc, iters);
    run2<5>("float add, 16 active lanes", out, cyc, iters);
    run2<6>("double add, 64 consecutive", out, cyc, iters);
    run2<7>("u64 add, 64 consecutive", out, cyc, iters);
    run2<8>("float max, 64 consecutive", out, cyc, iters);
    run2<9>("double add, 8 lanes share an address", out, cyc, iters);
    run2<10>("double add, scattered (stride 10 words)", out, cyc, iters);
    run2<11>("double add, 10 active lanes", out, cyc, iters);
    return 0;
}
