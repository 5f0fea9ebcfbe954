// Per-pixel shading of the rasteriser (perspective-correct barycentrics, their screen-space derivatives) and its
// backward, shared by rasterize.hip and fused.hip.
#pragma once
#include "common.h"

// ---------------------------------------------------------------------------------------------
// f32 shading of one pixel: perspective-correct barycentrics of vertices 0,1, z/w and (optionally)
// the screen-space derivatives.  fx,fy = pixel centre in NDC.
struct Shade { float u, v, zw, dudx, dudy, dvdx, dvdy; };

__device__ __forceinline__ Shade shade_pixel(float4 v0, float4 v1, float4 v2, float fx, float fy, float sx, float sy) {
    float p0x = v0.x - fx * v0.w, p0y = v0.y - fy * v0.w;
    float p1x = v1.x - fx * v1.w, p1y = v1.y - fy * v1.w;
    float p2x = v2.x - fx * v2.w, p2y = v2.y - fy * v2.w;
    float a0 = p1x * p2y - p1y * p2x;
    float a1 = p2x * p0y - p2y * p0x;
    float a2 = p0x * p1y - p0y * p1x;
    float at = a0 + a1 + a2;
    float iw = 1.0f / at;
    float b0 = a0 * iw, b1 = a1 * iw;
    Shade s;
    float zw = (a0 * v0.z + a1 * v1.z + a2 * v2.z) / (a0 * v0.w + a1 * v1.w + a2 * v2.w);
    s.zw = fminf(fmaxf(zw, -1.0f), 1.0f);
    float da0x = v2.w * p1y - v1.w * p2y, da0y = v1.w * p2x - v2.w * p1x;
    float da1x = v0.w * p2y - v2.w * p0y, da1y = v2.w * p0x - v0.w * p2x;
    float da2x = v1.w * p0y - v0.w * p1y, da2y = v0.w * p1x - v1.w * p0x;
    float datx = da0x + da1x + da2x, daty = da0y + da1y + da2y;
    s.dudx = (da0x - b0 * datx) * iw * sx;
    s.dudy = (da0y - b0 * daty) * iw * sy;
    s.dvdx = (da1x - b1 * datx) * iw * sx;
    s.dvdy = (da1y - b1 * daty) * iw * sy;
    float uc = fminf(fmaxf(b0, 0.0f), 1.0f), vc = fminf(fmaxf(b1, 0.0f), 1.0f);
    // u = uc / max(uc + vc, 1): the division is by exactly 1 unless rounding pushed the sum past it -- waves without such a
    // pixel skip it (x * (1 / 1) == x, so the values are the same)
    float sc = 1.0f;
    if (__builtin_amdgcn_ballot_w64(uc + vc > 1.0f) != 0) {      // (a wave vote and an asm the compiler cannot hoist: a plain
        float sum = uc + vc;                                      //  `if` is if-converted, and every pixel pays the division)
        asm volatile("; renormalise" : "+v"(sum));
        if (sum > 1.0f) sc = 1.0f / sum;
    }
    s.u = uc * sc;
    s.v = vc * sc;
    return s;
}

// ---------------------------------------------------------------------------------------------
// Backward of shade_pixel for one pixel: chains (dL/du, dL/dv) = (g.x, g.y) and, with HAS_DDB, dL/d(rast_db) = gd
// to the (x, y, w) of the three clip-space vertices.
template <bool HAS_DDB>
__device__ __forceinline__ void shade_pixel_bwd(float4 v0, float4 v1, float4 v2, float fx, float fy, float sx, float sy, float4 g,
                                                float4 gd, float (&g0)[3], float (&g1)[3], float (&g2)[3]) {
    // ---- forward recompute ----
    const float w0 = v0.w, w1 = v1.w, w2 = v2.w;
    const float p0x = v0.x - fx * w0, p0y = v0.y - fy * w0;
    const float p1x = v1.x - fx * w1, p1y = v1.y - fy * w1;
    const float p2x = v2.x - fx * w2, p2y = v2.y - fy * w2;
    const float a0 = p1x * p2y - p1y * p2x;
    const float a1 = p2x * p0y - p2y * p0x;
    const float a2 = p0x * p1y - p0y * p1x;
    const float at = a0 + a1 + a2;
    const float iw = 1.0f / at;
    const float b0 = a0 * iw, b1 = a1 * iw;
    // ---- undo clamp / renormalise:  u = uc * s, v = vc * s, s = 1 / max(uc + vc, 1) ----
    const float uc = fminf(fmaxf(b0, 0.0f), 1.0f), vc = fminf(fmaxf(b1, 0.0f), 1.0f);
    float guc = g.x, gvc = g.y;
    const float sum = uc + vc;
    if (__builtin_amdgcn_ballot_w64(sum > 1.0f) != 0 && sum > 1.0f) {      // (wave vote first: see shade_pixel)
        float sum_ = sum;
        asm volatile("; renormalise" : "+v"(sum_));
        const float s = 1.0f / sum_;
        const float dot = (uc * g.x + vc * g.y) * s * s;
        guc = g.x * s - dot;
        gvc = g.y * s - dot;
    }
    float gb0 = (b0 >= 0.0f && b0 <= 1.0f) ? guc : 0.0f;
    float gb1 = (b1 >= 0.0f && b1 <= 1.0f) ? gvc : 0.0f;
    // ---- reverse through the derivative outputs ----
    float ga0 = 0.f, ga1 = 0.f, ga2 = 0.f, giw = 0.f;
    float gp0x = 0.f, gp0y = 0.f, gp1x = 0.f, gp1y = 0.f, gp2x = 0.f, gp2y = 0.f;
    float gw0 = 0.f, gw1 = 0.f, gw2 = 0.f;
    if (HAS_DDB) {
        const float da0x = w2 * p1y - w1 * p2y, da0y = w1 * p2x - w2 * p1x;
        const float da1x = w0 * p2y - w2 * p0y, da1y = w2 * p0x - w0 * p2x;
        const float da2x = w1 * p0y - w0 * p1y, da2y = w0 * p1x - w1 * p0x;
        const float datx = da0x + da1x + da2x, daty = da0y + da1y + da2y;
        const float hx0 = gd.x * sx, hy0 = gd.y * sy, hx1 = gd.z * sx, hy1 = gd.w * sy;
        // dudx = n0x * iw * sx,  n0x = da0x - b0 * datx   (same for the other three)
        const float n0x = da0x - b0 * datx, n0y = da0y - b0 * daty;
        const float n1x = da1x - b1 * datx, n1y = da1y - b1 * daty;
        giw += hx0 * n0x + hy0 * n0y + hx1 * n1x + hy1 * n1y;
        const float gn0x = hx0 * iw, gn0y = hy0 * iw, gn1x = hx1 * iw, gn1y = hy1 * iw;
        gb0 -= gn0x * datx + gn0y * daty;
        gb1 -= gn1x * datx + gn1y * daty;
        const float gdatx = -(gn0x * b0 + gn1x * b1), gdaty = -(gn0y * b0 + gn1y * b1);
        const float gda0x = gn0x + gdatx, gda0y = gn0y + gdaty;
        const float gda1x = gn1x + gdatx, gda1y = gn1y + gdaty;
        const float gda2x = gdatx, gda2y = gdaty;
        // da0x = w2*p1y - w1*p2y ; da0y = w1*p2x - w2*p1x
        gw2 += gda0x * p1y; gp1y += gda0x * w2; gw1 -= gda0x * p2y; gp2y -= gda0x * w1;
        gw1 += gda0y * p2x; gp2x += gda0y * w1; gw2 -= gda0y * p1x; gp1x -= gda0y * w2;
        // da1x = w0*p2y - w2*p0y ; da1y = w2*p0x - w0*p2x
        gw0 += gda1x * p2y; gp2y += gda1x * w0; gw2 -= gda1x * p0y; gp0y -= gda1x * w2;
        gw2 += gda1y * p0x; gp0x += gda1y * w2; gw0 -= gda1y * p2x; gp2x -= gda1y * w0;
        // da2x = w1*p0y - w0*p1y ; da2y = w0*p1x - w1*p0x
        gw1 += gda2x * p0y; gp0y += gda2x * w1; gw0 -= gda2x * p1y; gp1y -= gda2x * w0;
        gw0 += gda2y * p1x; gp1x += gda2y * w0; gw1 -= gda2y * p0x; gp0x -= gda2y * w1;
    }
    // b0 = a0 * iw ; b1 = a1 * iw
    ga0 += gb0 * iw; ga1 += gb1 * iw;
    giw += gb0 * a0 + gb1 * a1;
    // iw = 1 / at ; at = a0 + a1 + a2
    const float gat = -giw * iw * iw;
    ga0 += gat; ga1 += gat; ga2 += gat;
    // a0 = p1x*p2y - p1y*p2x ; a1 = p2x*p0y - p2y*p0x ; a2 = p0x*p1y - p0y*p1x
    gp1x += ga0 * p2y; gp2y += ga0 * p1x; gp1y -= ga0 * p2x; gp2x -= ga0 * p1y;
    gp2x += ga1 * p0y; gp0y += ga1 * p2x; gp2y -= ga1 * p0x; gp0x -= ga1 * p2y;
    gp0x += ga2 * p1y; gp1y += ga2 * p0x; gp0y -= ga2 * p1x; gp1x -= ga2 * p0y;
    // p_kx = x_k - fx * w_k ; p_ky = y_k - fy * w_k
    g0[0] = gp0x; g0[1] = gp0y; g0[2] = gw0 - fx * gp0x - fy * gp0y;
    g1[0] = gp1x; g1[1] = gp1y; g1[2] = gw1 - fx * gp1x - fy * gp1y;
    g2[0] = gp2x; g2[1] = gp2y; g2[2] = gw2 - fx * gp2x - fy * gp2y;
}

// ---------------------------------------------------------------------------------------------
// One-pass objective (objective.hip): the pixel that is shaded also chains its gradient back, so the forward's intermediates are
// KEPT instead of recomputed.  shade_uvz is shade_pixel without the derivative outputs, shade_uv_bwd is shade_pixel_bwd<false> on the
// kept values: same operations in the same order, bit-identical results.
struct ShadeKeep { float p0x, p0y, p1x, p1y, p2x, p2y, a0, a1, iw, b0, b1, uc, vc, sc; };      // sc: 1 / (uc + vc) where that sum exceeds 1, else 1

// z/w of shade_pixel from the three unnormalised barycentric weights
__device__ __forceinline__ float shade_zw(float4 v0, float4 v1, float4 v2, float a0, float a1, float a2) {
    const float z = (a0 * v0.z + a1 * v1.z + a2 * v2.z) / (a0 * v0.w + a1 * v1.w + a2 * v2.w);
    return fminf(fmaxf(z, -1.0f), 1.0f);
}

// WANT_Z false: z/w is not formed (two IEEE divisions' worth of instructions; only deferred pixels store it)
template <bool WANT_Z = true>
__device__ __forceinline__ void shade_uvz(float4 v0, float4 v1, float4 v2, float fx, float fy, ShadeKeep &k, float &u, float &v, float &zw) {
    k.p0x = v0.x - fx * v0.w; k.p0y = v0.y - fy * v0.w;
    k.p1x = v1.x - fx * v1.w; k.p1y = v1.y - fy * v1.w;
    k.p2x = v2.x - fx * v2.w; k.p2y = v2.y - fy * v2.w;
    k.a0 = k.p1x * k.p2y - k.p1y * k.p2x;
    k.a1 = k.p2x * k.p0y - k.p2y * k.p0x;
    const float a2 = k.p0x * k.p1y - k.p0y * k.p1x;
    const float at = k.a0 + k.a1 + a2;
    k.iw = 1.0f / at;
    k.b0 = k.a0 * k.iw; k.b1 = k.a1 * k.iw;
    if (WANT_Z) zw = shade_zw(v0, v1, v2, k.a0, k.a1, a2);
    k.uc = fminf(fmaxf(k.b0, 0.0f), 1.0f); k.vc = fminf(fmaxf(k.b1, 0.0f), 1.0f);
    float sc = 1.0f;
    if (__builtin_amdgcn_ballot_w64(k.uc + k.vc > 1.0f) != 0) {      // (wave vote first: see shade_pixel)
        float sum = k.uc + k.vc;
        asm volatile("; renormalise" : "+v"(sum));
        if (sum > 1.0f) sc = 1.0f / sum;
    }
    k.sc = sc;
    u = k.uc * sc;
    v = k.vc * sc;
}

__device__ __forceinline__ void shade_uv_bwd(const ShadeKeep &k, float fx, float fy, float gu, float gv, float (&g0)[3], float (&g1)[3],
                                             float (&g2)[3]) {
    float guc = gu, gvc = gv;
    const float sum = k.uc + k.vc;
    if (__builtin_amdgcn_ballot_w64(sum > 1.0f) != 0 && sum > 1.0f) {
        const float s = k.sc;      // (= 1.0f / sum, the forward's IEEE division of the same sum: kept, not repeated -- 11 instructions a pixel)
        const float dot = (k.uc * gu + k.vc * gv) * s * s;
        guc = gu * s - dot;
        gvc = gv * s - dot;
    }
    const float gb0 = (k.b0 >= 0.0f && k.b0 <= 1.0f) ? guc : 0.0f;
    const float gb1 = (k.b1 >= 0.0f && k.b1 <= 1.0f) ? gvc : 0.0f;
    float ga0 = 0.f, ga1 = 0.f, ga2 = 0.f, giw = 0.f;
    float gp0x = 0.f, gp0y = 0.f, gp1x = 0.f, gp1y = 0.f, gp2x = 0.f, gp2y = 0.f;
    ga0 += gb0 * k.iw; ga1 += gb1 * k.iw;
    giw += gb0 * k.a0 + gb1 * k.a1;
    const float gat = -giw * k.iw * k.iw;
    ga0 += gat; ga1 += gat; ga2 += gat;
    gp1x += ga0 * k.p2y; gp2y += ga0 * k.p1x; gp1y -= ga0 * k.p2x; gp2x -= ga0 * k.p1y;
    gp2x += ga1 * k.p0y; gp0y += ga1 * k.p2x; gp2y -= ga1 * k.p0x; gp0x -= ga1 * k.p2y;
    gp0x += ga2 * k.p1y; gp1y += ga2 * k.p0x; gp0y -= ga2 * k.p1x; gp1x -= ga2 * k.p0y;
    g0[0] = gp0x; g0[1] = gp0y; g0[2] = 0.f - fx * gp0x - fy * gp0y;
    g1[0] = gp1x; g1[1] = gp1y; g1[2] = 0.f - fx * gp1x - fy * gp1y;
    g2[0] = gp2x; g2[1] = gp2y; g2[2] = 0.f - fx * gp2x - fy * gp2y;
}
