// Shared helpers for the gfx950 kernels (wave64, CDNA4).  Not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/fpcdr.h"

#define FPCDR_WAVE 64

void fpcdr_set_error(const char *fmt, ...);

#define FPCDR_REQUIRE(cond, msg)                                   \
    do {                                                           \
        if (!(cond)) {                                             \
            fpcdr_set_error("%s: %s", __func__, msg);              \
            return FPCDR_EINVAL;                                   \
        }                                                          \
    } while (0)

#define FPCDR_CHECK_LAUNCH()                                                        \
    do {                                                                            \
        hipError_t e_ = hipGetLastError();                                          \
        if (e_ != hipSuccess) {                                                     \
            fpcdr_set_error("%s: launch failed: %s", __func__, hipGetErrorString(e_)); \
            return FPCDR_ELAUNCH;                                                   \
        }                                                                           \
    } while (0)

static inline int fpcdr_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// ---- wave-level helpers ---------------------------------------------------------------------

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

// Sum over the lanes whose bit is set in `mask` (mask must contain the calling lane and be the
// same in all participating lanes).  Result valid in every participating lane.
__device__ __forceinline__ float wave_sum_masked(float v, unsigned long long mask) {
    // butterfly over all 64 lanes with non-members contributing 0
    int l = lane_id();
    float x = ((mask >> l) & 1ull) ? v : 0.0f;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) x += __shfl_xor(x, o, 64);
    return x;
}

// Group-reduce-then-atomic: lanes of a wave that share `key` (>= 0) sum their `n` values and the
// group leader issues the atomics.  Lanes with key < 0 do not participate.  All lanes of the wave
// must call this (convergent).
template <int N>
__device__ __forceinline__ void wave_group_atomic_add(int key, float *const (&dst)[N], const float (&val)[N]) {
    unsigned long long todo = __ballot(key >= 0);
    int l = lane_id();
    while (todo) {
        int leader = __ffsll((long long)todo) - 1;
        int k = __shfl(key, leader, 64);
        unsigned long long grp = __ballot(key == k) & todo;
        bool mine = (grp >> l) & 1ull;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            float x = mine ? val[i] : 0.0f;
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) x += __shfl_xor(x, o, 64);
            if (l == leader && x != 0.0f) atomicAdd(dst[i], x);
        }
        todo &= ~grp;
    }
}
