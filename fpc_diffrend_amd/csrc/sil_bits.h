// Per-image silhouette classification shared by antialias.hip (k_sil) and fused.hip (k_sil2): 3 bits per (image, triangle), "edge e
// is a silhouette edge in this image" (boundary edge, or the two triangles' third vertices on the same side of it), on uncentred
// pixel-scaled homogeneous coordinates.  The kernels are chains of gathers with two dozen instructions behind them -- latency, not
// issue, is their cost -- so a thread classifies its triangle in SIL_NI images: the six indices (own vertices, vertices across the
// three edges) are loaded once, and the 6 x SIL_NI position gathers are all in flight before the first is used (the vertex across an
// edge used to be fetched only after the edge's line had been computed): 87 -> 69 us for 288 images of 30 000 triangles.
#pragma once
#include "common.h"

#define FPCDR_SIL_NI 2
constexpr int SIL_NI = FPCDR_SIL_NI;

// the three bits of ONE triangle in ONE image: own vertices c0..c2, the vertices across its three edges o0..o2 (adjacency ad[e]: -1 = no
// neighbour -> boundary edge; out of range = ignored), pixel-scaled by (hw, hh).  One statement of the rule for every kernel.
__device__ __forceinline__ unsigned int sil_bits_of(const float4 &c0, const float4 &c1, const float4 &c2, const float4 &o0, const float4 &o1,
                                                    const float4 &o2, const int (&ad)[3], int V, float hw, float hh) {
    const float qx[3] = {c0.x * hw, c1.x * hw, c2.x * hw}, qy[3] = {c0.y * hh, c1.y * hh, c2.y * hh}, qw[3] = {c0.w, c1.w, c2.w};
    const float ox[3] = {o0.x * hw, o1.x * hw, o2.x * hw}, oy[3] = {o0.y * hh, o1.y * hh, o2.y * hh}, ow[3] = {o0.w, o1.w, o2.w};
    unsigned int bits = 0;
#pragma unroll
    for (int e = 0; e < 3; ++e) {
        if (ad[e] == -1) { bits |= 1u << e; continue; }
        if (ad[e] < 0 || ad[e] >= V) continue;
        const int a = (e + 1) % 3, bb = (e + 2) % 3;
        const float Lx = qy[a] * qw[bb] - qw[a] * qy[bb];
        const float Ly = qw[a] * qx[bb] - qx[a] * qw[bb];
        const float Lz = qx[a] * qy[bb] - qy[a] * qx[bb];
        const float so = Lx * qx[e] + Ly * qy[e] + Lz * qw[e];
        const float sp = Lx * ox[e] + Ly * oy[e] + Lz * ow[e];
        if ((so > 0.0f && sp > 0.0f) || (so < 0.0f && sp < 0.0f)) bits |= 1u << e;
    }
    return bits;
}

// thread = triangle t of images b0 .. b0 + SIL_NI - 1 (grid: x over 256-triangle chunks, y over groups of SIL_NI images)
__device__ __forceinline__ void sil_classify(const float4 *__restrict__ pos, const int32_t *__restrict__ tri, const int32_t *__restrict__ adj,
                                             int B, int V, int T, float hw, float hh, uint8_t *__restrict__ sil, int b0, int t) {
    const int vi[3] = {tri[3 * t], tri[3 * t + 1], tri[3 * t + 2]};
    const int ad[3] = {adj[3 * t], adj[3 * t + 1], adj[3 * t + 2]};
    bool ok = true;
    for (int k = 0; k < 3; ++k) ok &= (vi[k] >= 0 && vi[k] < V);
    unsigned int idx[6];      // (an index that is not used reads vertex 0)
    for (int k = 0; k < 3; ++k) {
        idx[k] = ok ? (unsigned int)vi[k] : 0u;
        idx[3 + k] = (ad[k] >= 0 && ad[k] < V) ? (unsigned int)ad[k] : 0u;
    }
    float4 c[SIL_NI][6];
#pragma unroll
    for (int i = 0; i < SIL_NI; ++i) {
        const float4 *p = pos + (size_t)min(b0 + i, B - 1) * V;
#pragma unroll
        for (int k = 0; k < 6; ++k) c[i][k] = ld32(p, idx[k]);
    }
#pragma unroll
    for (int i = 0; i < SIL_NI; ++i) {
        if (b0 + i >= B) break;
        const unsigned int bits = ok ? sil_bits_of(c[i][0], c[i][1], c[i][2], c[i][3], c[i][4], c[i][5], ad, V, hw, hh) : 0u;
        sil[(size_t)(b0 + i) * T + t] = (uint8_t)bits;
    }
}
