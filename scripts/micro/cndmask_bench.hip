// v_cndmask_b32 on gfx950: how many cycles does a wave-instruction hold its SIMD?  (profiles/r03_valu_rate_bench.txt showed 5.6 x a
// v_fma_f32 for the VCC form inside an asm statement that declared vcc clobbered; this bench separates the encodings.)
//   hipcc -O3 --offload-arch=gfx950 scripts/micro/cndmask_bench.hip -o /tmp/cndmask_bench && /tmp/cndmask_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#define R8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define BODY64(X) R8(X) R8(X) R8(X) R8(X) R8(X) R8(X) R8(X) R8(X)
template <int OP>
__global__ void __launch_bounds__(256) k(float *out, int iters, unsigned long long mask) {
    unsigned int u[8];
    float a[8];
    for (int i = 0; i < 8; ++i) { u[i] = threadIdx.x + i; a[i] = (float)(threadIdx.x + i); }
    const unsigned int m = threadIdx.x * 3u;
    const float fm = 1.0001f;
    for (int it = 0; it < iters; ++it) {
        if (OP == 0) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(fm));
            BODY64(X)
#undef X
        } else if (OP == 1) {      // e64 encoding, condition in an SGPR pair
#define X(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(u[i]) : "v"(m), "s"(mask));
            BODY64(X)
#undef X
        } else if (OP == 2) {      // e32 encoding, condition in VCC (set once before the loop body)
            asm volatile("s_mov_b64 vcc, %0" : : "s"(mask) : "vcc");
#define X(i) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(m));
            BODY64(X)
#undef X
        } else if (OP == 3) {      // compare + select pairs, as compiled code has them
#define X(i) asm volatile("v_cmp_gt_u32_e32 vcc, %1, %0\n\tv_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(m) : "vcc");
            BODY64(X)
#undef X
        } else if (OP == 4) {      // the compare alone
#define X(i) asm volatile("v_cmp_gt_u32_e32 vcc, %1, %0" : : "v"(u[i]), "v"(m) : "vcc");
            BODY64(X)
#undef X
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += a[i] + (float)u[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP> double run(const char *name, float *out, int iters, double base) {
    const int cus = 256, wgs = cus * 8;      // 8 workgroups of 4 waves per CU: 8 waves per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(wgs), dim3(256), 0, 0, out, 10, 0x5555555555555555ull);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(wgs), dim3(256), 0, 0, out, iters, 0x5555555555555555ull);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double per = ms * 1e6 / ((double)iters * 64 * (OP == 3 ? 2 : 1) * 8);      // ns per wave-instruction and SIMD (8 waves per SIMD)
    printf("%-34s %.3f ns per wave-instruction and SIMD   %.2f x v_fma_f32\n", name, per, base > 0 ? per / base : 1.0);
    return per;
}
int main() {
    float *out; hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    const int iters = 2000;
    const double b = run<0>("v_fma_f32", out, iters, 0);
    run<1>("v_cndmask_b32_e64 (sgpr pair)", out, iters, b);
    run<2>("v_cndmask_b32_e32 (vcc)", out, iters, b);
    run<3>("v_cmp + v_cndmask (per instruction)", out, iters, b);
    run<4>("v_cmp_gt_u32_e32", out, iters, b);
    hipFree(out);
    return 0;
}
