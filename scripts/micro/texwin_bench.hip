// LDS accumulation in the shape of the fused objective's texel window (objective.hip / fused.hip): a wave instruction adds one
// bilinear tap of 64 pixels -- lanes 0..31 a row of the bin, lanes 32..63 the next row (reversed or not) -- into a window of row
// stride S cells at `density` texels per pixel.  Cycles per wave-instruction as the LDS sees them (4 waves issue into one LDS).
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/texwin_bench.hip -o scripts/micro/texwin_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %d\n", (int)e_); return 1; } } while (0)

template <int TYPE>   // 0 double add, 1 int32 add, 2 plain int32 store (reference), 3 u64 add
__global__ void __launch_bounds__(256) k(float *out, int iters, long long *cycles, float density, int S, int reversed, int onerow) {
    __shared__ double sd[4096];
    int *si = reinterpret_cast<int *>(sd);
    for (int i = threadIdx.x; i < 4096; i += 256) sd[i] = 0.0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = onerow ? lane : ((lane & 32) ? (reversed ? 63 - lane : lane - 32) : lane);
    const int cx = (int)floorf((float)col * density);
    int cells[16];      // the 16 (pass, tap) cells of this thread, precomputed: the timed loop is loads of nothing but LDS instructions
    for (int k = 0; k < 4; ++k) {
        const int cy = (int)floorf((float)(8 * wave + 2 * k + (onerow ? 0 : (lane >> 5))) * density);
        for (int t = 0; t < 4; ++t) cells[4 * k + t] = ((cy + (t >> 1)) * S + cx + (t & 1)) & 4095;
    }
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (TYPE == 0) atomicAdd(&sd[cells[j]], 1.0);
            else if (TYPE == 1) atomicAdd(&si[cells[j]], it);
            else if (TYPE == 3) atomicAdd(reinterpret_cast<unsigned long long *>(&sd[cells[j]]), (unsigned long long)(long long)(it - 7));
            else si[cells[j]] = it;
        }
    }
    __syncthreads();
    const long long t1 = clock64();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    out[blockIdx.x * 256 + threadIdx.x] = (float)sd[threadIdx.x];
}

template <int TYPE>
int run(const char *name, float *out, long long *cyc, float density, int S, int reversed, int onerow) {
    const int iters = 512;
    hipLaunchKernelGGL(k<TYPE>, dim3(256), dim3(256), 0, 0, out, iters, cyc, density, S, reversed, onerow);
    CK(hipDeviceSynchronize());
    long long h[256];
    CK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
    double mean = 0;
    for (int i = 0; i < 256; ++i) mean += (double)h[i];
    mean /= 256;
    printf("%-12s density %.2f stride %2d %s %7.1f cycles per wave-instruction\n", name, density, S,
           onerow ? "one row of 64     " : (reversed ? "two rows, reversed" : "two rows          "), mean / iters / 16 / 4);
    return 0;
}

int main() {
    float *out; long long *cyc;
    CK(hipMalloc(&out, 256 * 256 * 4)); CK(hipMalloc(&cyc, 256 * 8));
    for (int S : {40, 32}) {
        for (int mode = 0; mode < 3; ++mode) {
            if (run<0>("double add", out, cyc, 1.0f, S, mode == 2, mode == 0)) return 1;
            if (run<1>("int32 add", out, cyc, 1.0f, S, mode == 2, mode == 0)) return 1;
            if (run<2>("int32 store", out, cyc, 1.0f, S, mode == 2, mode == 0)) return 1;
        }
    }
    for (float d : {1.0f, 0.9f, 0.7f, 0.5f, 0.25f}) {
        if (run<0>("double add", out, cyc, d, 40, 1, 0)) return 1;
        if (run<1>("int32 add", out, cyc, d, 40, 1, 0)) return 1;
        if (run<3>("u64 add", out, cyc, d, 40, 1, 0)) return 1;
    }
    return 0;
}
