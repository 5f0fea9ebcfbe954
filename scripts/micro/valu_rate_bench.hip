// VALU issue cost per instruction type on gfx950: which operations are full rate, which are not.  Every wave runs a loop of
// 64 independent instructions of one kind (8 accumulators, no dependency stalls), 8 waves per SIMD; the result is printed as
// cycles per wave-instruction and SIMD, relative to v_fma_f32.
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/valu_rate_bench.hip -o scripts/micro/valu_rate_bench && scripts/micro/valu_rate_bench
#include <hip/hip_runtime.h>
#include <stdio.h>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define BODY64(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X)

template <int OP>
__global__ void __launch_bounds__(512) k(float *out, int iters) {
    float a[8];
    double d[8];
    unsigned int u[8];
    unsigned long long w[8];
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f p[8];
    for (int i = 0; i < 8; ++i) {
        a[i] = 1.0f + threadIdx.x * 1e-3f + i; d[i] = a[i]; u[i] = threadIdx.x * 7 + i; w[i] = u[i]; p[i] = v2f{a[i], a[i] + 1.0f};
    }
    const float c = 1.0000001f, e = 1e-7f;
    const unsigned int m = threadIdx.x | 1;
    for (int it = 0; it < iters; ++it) {
        if (OP == 0) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(e));
            BODY64(X)
#undef X
        } else if (OP == 1) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(p[(i + 1) & 7]), "v"(p[(i + 2) & 7]));
            BODY64(X)
#undef X
        } else if (OP == 2) {
#define X(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[i]) : "v"(m));
            BODY64(X)
#undef X
        } else if (OP == 3) {
#define X(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(u[i]) : "v"(m));
            BODY64(X)
#undef X
        } else if (OP == 4) {
#define X(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w[i]) : "v"(u[i]), "v"(m) : "vcc");
            BODY64(X)
#undef X
        } else if (OP == 5) {
#define X(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            BODY64(X)
#undef X
        } else if (OP == 6) {
#define X(i) asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(a[i]) : "v"(u[i]));
            BODY64(X)
#undef X
        } else if (OP == 7) {
#define X(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(d[(i + 1) & 7]), "v"(d[(i + 2) & 7]));
            BODY64(X)
#undef X
        } else if (OP == 8) {
#define X(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(m));
            BODY64(X)
#undef X
        } else if (OP == 9) {
#define X(i) asm volatile("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(w[i]) : "v"(w[(i + 1) & 7]));
            BODY64(X)
#undef X
        } else if (OP == 10) {
#define X(i) asm volatile("v_floor_f32 %0, %0" : "+v"(a[i]));
            BODY64(X)
#undef X
        } else if (OP == 11) {
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 7]));
            BODY64(X)
#undef X
        } else if (OP == 12) {
#define X(i) asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(u[i]) : "v"(m));
            BODY64(X)
#undef X
        } else if (OP == 13) {
#define X(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(u[i]) : "v"(m));
            BODY64(X)
#undef X
        } else if (OP == 14) {
#define X(i) asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(a[i]) : "v"(c) : "vcc");
            BODY64(X)
#undef X
        } else if (OP == 15) {
#define X(i) asm volatile("v_div_fixup_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(e));
            BODY64(X)
#undef X
        } else if (OP == 16) {
#define X(i) asm volatile("v_fmac_f32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(a[(i + 1) & 7]), "v"(c));
            BODY64(X)
#undef X
        } else if (OP == 17) {
#define X(i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(a[i]));
            BODY64(X)
#undef X
        } else if (OP == 18) {
#define X(i) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(w[i]) : "v"(u[i]), "v"(m) : "vcc");
            BODY64(X)
#undef X
        } else if (OP == 19) {
#define X(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(m) : "vcc");
            BODY64(X)
#undef X
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += a[i] + (float)d[i] + (float)u[i] + (float)w[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
double run(const char *name, float *out, int iters, double base) {
    int dev = 0, cus = 256;
    hipGetDevice(&dev);
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    // 4 workgroups of 512 threads per CU = 8 waves per SIMD
    hipLaunchKernelGGL(k<OP>, dim3(cus * 4), dim3(512), 0, 0, out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(cus * 4), dim3(512), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: 8 waves x iters x 64 instructions
    const double ns_per_instr = (double)ms * 1e6 / (8.0 * iters * 64.0);
    printf("%-22s %8.3f ns per wave-instruction and SIMD   %5.2f x v_fma_f32\n", name, ns_per_instr, base > 0 ? ns_per_instr / base : 1.0);
    return ns_per_instr;
}

int main() {
    float *out;
    hipMalloc(&out, 256 * 8 * 512 * sizeof(float));
    const int iters = 4000;
    const double b = run<0>("v_fma_f32", out, iters, 0);
    run<1>("v_pk_fma_f32", out, iters, b);
    run<11>("v_pk_mul_f32", out, iters, b);
    run<2>("v_mul_lo_u32", out, iters, b);
    run<13>("v_mul_hi_u32", out, iters, b);
    run<12>("v_mul_i32_i24", out, iters, b);
    run<3>("v_mad_u32_u24", out, iters, b);
    run<4>("v_mad_u64_u32", out, iters, b);
    run<18>("v_mad_i64_i32", out, iters, b);
    run<9>("v_lshl_add_u64", out, iters, b);
    run<8>("v_add_u32", out, iters, b);
    run<19>("v_cndmask_b32", out, iters, b);
    run<5>("v_rcp_f32", out, iters, b);
    run<14>("v_div_scale_f32", out, iters, b);
    run<15>("v_div_fixup_f32", out, iters, b);
    run<10>("v_floor_f32", out, iters, b);
    run<6>("v_cvt_f32_i32", out, iters, b);
    run<17>("v_cvt_f64_f32", out, iters, b);
    run<7>("v_fma_f64", out, iters, b);
    run<16>("v_fmac_f32_dpp", out, iters, b);
    hipFree(out);
    return 0;
}
