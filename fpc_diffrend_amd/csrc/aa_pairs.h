// Pixel-pair analysis of the antialias op (DESIGN.md "Antialias rules"), shared by antialias.hip and fused.hip.
#pragma once
#include "common.h"

// ------------------------------------------------------------------------------------------------
// pair analysis shared by forward and backward
// ------------------------------------------------------------------------------------------------
struct PairSel {
    int tau;      // chosen triangle, -1 = nothing to do
    int use1;     // 1: the triangle belongs to the second pixel of the pair
};

// ids are 1-based (0 = empty); z = rast.z
__device__ __forceinline__ PairSel pair_select(int id0, float z0, int id1, float z1, int T) {
    PairSel s;
    s.use1 = (id0 == 0) ? 1 : ((id1 == 0) ? 0 : (z1 < z0 ? 1 : 0));
    s.tau = (s.use1 ? id1 : id0) - 1;
    if (s.tau < 0 || s.tau >= T) s.tau = -1;
    return s;
}

struct EdgeEval {
    bool active;
    float t, Lx, Ly, Lz, qax, qay, wa, qbx, qby, wb;
};

// d: 0 = x pair, 1 = y pair; s = +1 if the partner pixel lies in +d direction from P, else -1
__device__ __forceinline__ EdgeEval edge_eval(float4 ca, float4 cb, float hw, float hh, float fxp, float fyp, int d, float s) {
    EdgeEval r;
    r.qax = ca.x * hw - fxp * ca.w; r.qay = ca.y * hh - fyp * ca.w; r.wa = ca.w;
    r.qbx = cb.x * hw - fxp * cb.w; r.qby = cb.y * hh - fyp * cb.w; r.wb = cb.w;
    r.Lx = r.qay * r.wb - r.wa * r.qby;
    r.Ly = r.wa * r.qbx - r.qax * r.wb;
    r.Lz = r.qax * r.qby - r.qay * r.qbx;
    float Ld, Lo, ya, yb;
    bool orient;
    if (d == 0) { Ld = r.Lx; Lo = r.Ly; ya = r.qay; yb = r.qby; orient = fabsf(Ld) >= fabsf(Lo); }
    else        { Ld = r.Ly; Lo = r.Lx; ya = r.qax; yb = r.qbx; orient = fabsf(Ld) > fabsf(Lo); }
    const bool extent = (ya < 0.0f) != (yb < 0.0f);
    const bool nz = Ld != 0.0f;
    const float den = nz ? s * Ld : 1.0f;
    r.t = -r.Lz / den;
    r.active = orient && extent && nz && (r.t >= 0.0f) && (r.t <= 1.0f);
    return r;
}

struct AAGeom {
    const float4 *pos;   // image's vertex buffer
    const int32_t *tri;
    const uint8_t *sil;  // image's silhouette bits
    int T, W, H;
    float hw, hh;
};

// Evaluate pair (x0,y0)-(x0+dx,y0+dy) (d = 0/1).  For every active silhouette edge calls
// f(t, Px, Py, Qx, Qy, va, vb, ev, s).
template <typename F>
__device__ __forceinline__ bool for_active_edges(const AAGeom &g, int x0, int y0, int d, int id0, float z0, int id1, float z1, F &&f) {
    const PairSel ps = pair_select(id0, z0, id1, z1, g.T);
    if (ps.tau < 0) return false;
    const unsigned int bits = g.sil[ps.tau];
    if (bits == 0) return false;
    const int x1 = x0 + (d == 0), y1 = y0 + (d == 1);
    const int Px = ps.use1 ? x1 : x0, Py = ps.use1 ? y1 : y0;
    const int Qx = ps.use1 ? x0 : x1, Qy = ps.use1 ? y0 : y1;
    const float s = ps.use1 ? -1.0f : 1.0f;
    const float fxp = (float)Px + 0.5f - g.hw, fyp = (float)Py + 0.5f - g.hh;
    const int vi[3] = {g.tri[3 * ps.tau], g.tri[3 * ps.tau + 1], g.tri[3 * ps.tau + 2]};
    const float4 c[3] = {g.pos[vi[0]], g.pos[vi[1]], g.pos[vi[2]]};
    bool any = false;
#pragma unroll
    for (int e = 0; e < 3; ++e) {
        if (!((bits >> e) & 1u)) continue;
        const int a = (e + 1) % 3, b = (e + 2) % 3;
        const EdgeEval ev = edge_eval(c[a], c[b], g.hw, g.hh, fxp, fyp, d, s);
        if (ev.active) {
            any = true;
            f(ev.t, Px, Py, Qx, Qy, vi[a], vi[b], ev, s);
        }
    }
    return any;
}

__device__ __forceinline__ float2 load_zid(const float4 *rast, size_t off) {
    return reinterpret_cast<const float2 *>(rast)[2 * off + 1];     // (z/w, id): the upper 8 bytes of the pixel
}

