// transform_clip for a minibatch, forward and backward, for gfx950 (MI355X).
//
// Performs the work of reference src/torch/camera.py:11-23 (`posw @ mvp.T` on homogeneous vertices) for all
// B = F x Nc images of a step at once: image b = f * Nc + c uses vertex buffer f and matrix b.  The reference (and
// fpc_diffrend_amd.camera.transform_clip) leave this to torch.matmul; at [288,15002,4] x [4,4] the batched GEMM
// picked by the BLAS library costs 0.3 ms forward and 0.6 ms backward for 69 MB of traffic, so the fit loop uses
// these streaming kernels instead (the views of a frame read the same vertices through L2).
#include "common.h"

namespace {

__global__ void __launch_bounds__(256) k_clip_fwd(const float *__restrict__ mvp, const float *__restrict__ verts, int V, int Nc,
                                                  float4 *__restrict__ out) {
    const int b = blockIdx.y, f = b / Nc;
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    const float *m = mvp + (size_t)b * 16;   // wave-uniform: scalar loads
    const float *p = verts + ((size_t)f * V + v) * 3;
    const float x = p[0], y = p[1], z = p[2];
    float4 o;
    o.x = x * m[0] + y * m[1] + z * m[2] + m[3];
    o.y = x * m[4] + y * m[5] + z * m[6] + m[7];
    o.z = x * m[8] + y * m[9] + z * m[10] + m[11];
    o.w = x * m[12] + y * m[13] + z * m[14] + m[15];
    out[(size_t)b * V + v] = o;
}

// d/d verts: thread = (frame, vertex), sums over the frame's Nc views (no atomics)
__global__ void __launch_bounds__(256) k_clip_bwd_verts(const float *__restrict__ mvp, const float4 *__restrict__ g, int V, int Nc,
                                                        float *__restrict__ g_verts) {
    const int f = blockIdx.y;
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    float gx = 0.f, gy = 0.f, gz = 0.f;
    for (int c = 0; c < Nc; ++c) {
        const int b = f * Nc + c;
        const float *m = mvp + (size_t)b * 16;
        const float4 q = g[(size_t)b * V + v];
        gx += q.x * m[0] + q.y * m[4] + q.z * m[8] + q.w * m[12];
        gy += q.x * m[1] + q.y * m[5] + q.z * m[9] + q.w * m[13];
        gz += q.x * m[2] + q.y * m[6] + q.z * m[10] + q.w * m[14];
    }
    float *o = g_verts + ((size_t)f * V + v) * 3;
    o[0] = gx; o[1] = gy; o[2] = gz;
}

// d/d mvp[b][i][j] = sum_v g[b][v][i] * (x,y,z,1)[j]: wave DPP sums, then 16 atomics per wave
constexpr int MVP_VPB = 2048;   // vertices per workgroup of k_clip_bwd_mvp (8 per thread)
__global__ void __launch_bounds__(256) k_clip_bwd_mvp(const float *__restrict__ verts, const float4 *__restrict__ g, int V, int Nc,
                                                      float *__restrict__ g_mvp) {
    // grad_mvp[b] = sum_v grad_out[b][v] (x) (verts[v], 1): per-thread partial sums over 8 vertices, one DPP reduction per
    // wave, LDS across the four waves, then 16 atomics per workgroup (same-address atomics are what this kernel waits for)
    __shared__ float s_part[4][16];
    const int b = blockIdx.y, f = b / Nc;
    float acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    const int v_end = min((int)(blockIdx.x + 1) * MVP_VPB, V);
    for (int v = blockIdx.x * MVP_VPB + threadIdx.x; v < v_end; v += 256) {
        const float4 q = g[(size_t)b * V + v];
        const float *p = verts + ((size_t)f * V + v) * 3;
        const float pw[4] = {p[0], p[1], p[2], 1.0f};
        const float gi[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i * 4 + j] += gi[i] * pw[j];
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const float s = wave_sum_dpp(acc[i]);
        if (lane == 0) s_part[wave][i] = s;
    }
    __syncthreads();
    if (threadIdx.x < 16) {
        const float s = s_part[0][threadIdx.x] + s_part[1][threadIdx.x] + s_part[2][threadIdx.x] + s_part[3][threadIdx.x];
        if (s != 0.0f) atomicAdd(g_mvp + (size_t)b * 16 + threadIdx.x, s);
    }
}

// Both gradients from ONE pass over grad_out (69 MB at 288 x 15 k vertices, read twice by the two kernels above: 2 x 16 us): a thread
// takes CLIP_VPT vertices of one frame (a workgroup 256 * CLIP_VPT consecutive ones) through all views, as k_clip_bwd_verts does for
// one; the 16 products of a view are summed over the thread's vertices first, then over the wave (DPP), then over the four waves through
// an LDS row per view: one atomic per (view, entry) and workgroup.  (One vertex per thread spends the kernel on the 144 wave sums.)
constexpr int CLIP_VIEWS = 16;      // views per LDS round
constexpr int CLIP_VPT = 4;
__global__ void __launch_bounds__(256) k_clip_bwd_both(const float *__restrict__ mvp, const float *__restrict__ verts,
                                                       const float4 *__restrict__ g, int V, int Nc, float *__restrict__ g_verts,
                                                       float *__restrict__ g_mvp) {
    __shared__ float s_part[CLIP_VIEWS][4][16];
    const int f = blockIdx.y;
    const int v0 = blockIdx.x * (256 * CLIP_VPT) + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float pw[CLIP_VPT][4], gx[CLIP_VPT], gy[CLIP_VPT], gz[CLIP_VPT];
    bool ok[CLIP_VPT];
#pragma unroll
    for (int u = 0; u < CLIP_VPT; ++u) {
        const int v = v0 + u * 256;
        ok[u] = v < V;
        const float *pv = verts + ((size_t)f * V + (ok[u] ? v : 0)) * 3;
        pw[u][0] = ok[u] ? pv[0] : 0.0f; pw[u][1] = ok[u] ? pv[1] : 0.0f; pw[u][2] = ok[u] ? pv[2] : 0.0f; pw[u][3] = ok[u] ? 1.0f : 0.0f;
        gx[u] = 0.f; gy[u] = 0.f; gz[u] = 0.f;
    }
    for (int c0 = 0; c0 < Nc; c0 += CLIP_VIEWS) {
        const int nc = min(CLIP_VIEWS, Nc - c0);
        for (int c = 0; c < nc; ++c) {
            const int b = f * Nc + c0 + c;
            const float *m = mvp + (size_t)b * 16;
            float4 q[CLIP_VPT];
#pragma unroll
            for (int u = 0; u < CLIP_VPT; ++u) q[u] = ok[u] ? g[(size_t)b * V + v0 + u * 256] : make_float4(0.f, 0.f, 0.f, 0.f);
            float acc[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
#pragma unroll
            for (int u = 0; u < CLIP_VPT; ++u) {
                gx[u] += q[u].x * m[0] + q[u].y * m[4] + q[u].z * m[8] + q[u].w * m[12];
                gy[u] += q[u].x * m[1] + q[u].y * m[5] + q[u].z * m[9] + q[u].w * m[13];
                gz[u] += q[u].x * m[2] + q[u].y * m[6] + q[u].z * m[10] + q[u].w * m[14];
                const float gi[4] = {q[u].x, q[u].y, q[u].z, q[u].w};
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i * 4 + j] += gi[i] * pw[u][j];
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float sm = wave_sum_dpp(acc[i]);
                if (lane == 0) s_part[c][wave][i] = sm;
            }
        }
        __syncthreads();
        for (int e = threadIdx.x; e < nc * 16; e += 256) {
            const int c = e >> 4, i = e & 15;
            const float sm = s_part[c][0][i] + s_part[c][1][i] + s_part[c][2][i] + s_part[c][3][i];
            if (sm != 0.0f) atomicAdd(g_mvp + (size_t)(f * Nc + c0 + c) * 16 + i, sm);
        }
        __syncthreads();
    }
#pragma unroll
    for (int u = 0; u < CLIP_VPT; ++u)
        if (ok[u]) {
            float *o = g_verts + ((size_t)f * V + v0 + u * 256) * 3;
            o[0] = gx[u]; o[1] = gy[u]; o[2] = gz[u];
        }
}

// Uniform-Laplacian gather over a static, padded one-ring table (reference regulariser, fit.py:581 via pytorch3d):
//   mode 0 (forward):   out[f][v] = inv_deg[v] * sum_n x[f][n] - x[f][v]            (L x,   L = D^-1 A - I)
//   mode 1 (backward):  out[f][v] = sum_n inv_deg[n] * x[f][n] - x[f][v]            (L^T x, L^T = A D^-1 - I)
// nbr [V,D] holds neighbour indices, entries >= V are padding.  Both directions are gathers: no atomics.
// One vertex's ring sum: sum over the ring of w_n * fetch(n), w_n = inv_deg[n] (mode 1) or 1 (mode 0).
// slot-major table: consecutive threads read consecutive entries; a vertex's ring is stored front to back, so the
// first pad ends it (a UV sphere has two poles of degree ~100 among vertices of degree 6).  Eight slots are
// fetched together and their neighbours gathered together: the kernel is a chain of dependent loads otherwise
// (one slot per trip is ~100 dependent round trips for the two pole threads, 40 us of a launch whose other threads are done after 3).
// Rings longer than the first eight slots (a pole, the centre of a fan) are finished by the WHOLE WAVE, one such vertex at a time: lane l
// takes slots 8 + l, 72 + l, ..., three DPP sums hand the result to the owner.  (Left to its own thread a ring of 150 is 19 dependent
// rounds of index load + gather, ~40 us during which the launch's other 480 k threads have long finished.)  Every lane of the wave must
// call this, `ok` = the lane has a vertex.
template <typename Fetch>
__device__ __forceinline__ void ring_sum(const int32_t *__restrict__ nbr, const float *__restrict__ inv_deg, int V, int D, int mode, int v, bool ok,
                                         Fetch fetch, float &sx, float &sy, float &sz) {
    constexpr int U = 8;
    sx = 0.f; sy = 0.f; sz = 0.f;
    int nb[U];
#pragma unroll
    for (int d = 0; d < U; ++d) nb[d] = (ok && d < D) ? nbr[(size_t)d * V + v] : V;
    {
        float wgt[U], gx[U], gy[U], gz[U];
#pragma unroll
        for (int d = 0; d < U; ++d) {
            const bool in = nb[d] < V;
            const int n = in ? nb[d] : (ok ? v : 0);
            wgt[d] = in ? (mode ? inv_deg[n] : 1.0f) : 0.0f;
            fetch(n, gx[d], gy[d], gz[d]);
        }
#pragma unroll
        for (int d = 0; d < U; ++d) { sx += wgt[d] * gx[d]; sy += wgt[d] * gy[d]; sz += wgt[d] * gz[d]; }
    }
    const int lane = threadIdx.x & 63;
    unsigned long long heavy = __ballot(D > U && nb[U - 1] < V);
    while (heavy) {
        const int leader = __builtin_ctzll(heavy);
        heavy &= heavy - 1;
        const int hv = __shfl(v, leader, 64);
        float px = 0.f, py = 0.f, pz = 0.f;
        for (int d = U + lane; d < D; d += 64) {
            const int n = nbr[(size_t)d * V + hv];
            if (n < V) {
                const float w = mode ? inv_deg[n] : 1.0f;
                float gx, gy, gz;
                fetch(n, gx, gy, gz);
                px += w * gx; py += w * gy; pz += w * gz;
            }
        }
        px = wave_sum_dpp(px); py = wave_sum_dpp(py); pz = wave_sum_dpp(pz);
        if (lane == leader) { sx += px; sy += py; sz += pz; }
    }
}

__global__ void __launch_bounds__(256) k_lap_gather(const float *__restrict__ x, const int32_t *__restrict__ nbr,
                                                    const float *__restrict__ inv_deg, int V, int D, int mode,
                                                    float *__restrict__ out) {
    const int f = blockIdx.y;
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    const bool ok = v < V;
    const float *xf = x + (size_t)f * V * 3;
    float sx, sy, sz;
    ring_sum(nbr, inv_deg, V, D, mode, v, ok, [&](int n, float &a, float &b, float &c) { a = xf[3 * n]; b = xf[3 * n + 1]; c = xf[3 * n + 2]; },
             sx, sy, sz);
    if (!ok) return;
    const float s = mode ? 1.0f : inv_deg[v];
    float *o = out + ((size_t)f * V + v) * 3;
    o[0] = s * sx - xf[3 * v]; o[1] = s * sy - xf[3 * v + 1]; o[2] = s * sz - xf[3 * v + 2];
}

// The Laplacian term of the reference's objective (fit.py:581: weight * mesh_laplacian_smoothing(mesh)^2, one mesh per step; a batch
// takes the mean of the squares) as ONE launch each way instead of a gather and a dozen torch kernels:
//   per_f = mean_v || (L x_f)_v ||,   value = weight / F * sum_f per_f^2.
// Forward: every workgroup adds its vertices' norms to its mesh's double accumulator (one relaxed device-scope add per workgroup);
// a one-workgroup kernel behind it -- a kernel boundary orders the adds before its loads, no fence or ticket -- forms the value,
// stores per_f for the backward and zeroes the accumulators for the next call.  (r3 let the last workgroup to finish do that behind
// a ticket counter with relaxed atomics: correct on gfx950, where agent-scope read-modify-writes execute at the memory side, but
// not by the memory model; a release fence per workgroup writes the XCD's whole L2 back and slowed a concurrent store stream 4x.)
__global__ void __launch_bounds__(256) k_lap_penalty_fwd(const float *__restrict__ x, const int32_t *__restrict__ nbr,
                                                         const float *__restrict__ inv_deg, int F, int V, int D,
                                                         float *__restrict__ lap, double *__restrict__ acc) {
    __shared__ double s_part[4];
    const int f = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    float nrm = 0.0f;
    const float *xf = x + (size_t)f * V * 3;
    float sx, sy, sz;
    ring_sum(nbr, inv_deg, V, D, 0, v, v < V, [&](int n, float &a, float &b, float &c) { a = xf[3 * n]; b = xf[3 * n + 1]; c = xf[3 * n + 2]; },
             sx, sy, sz);
    if (v < V) {
        const float s = inv_deg[v];
        const float lx = s * sx - xf[3 * v], ly = s * sy - xf[3 * v + 1], lz = s * sz - xf[3 * v + 2];
        float *o = lap + ((size_t)f * V + v) * 3;
        o[0] = lx; o[1] = ly; o[2] = lz;
        nrm = sqrtf(lx * lx + ly * ly + lz * lz);
    }
    const float ws = wave_sum_dpp(nrm);
    if (lane == 0) s_part[wave] = (double)ws;
    __syncthreads();
    if (threadIdx.x == 0)
        __hip_atomic_fetch_add(&acc[f], s_part[0] + s_part[1] + s_part[2] + s_part[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void __launch_bounds__(256) k_lap_penalty_finish(double *__restrict__ acc, int F, int V, float weight, float *__restrict__ per,
                                                            float *__restrict__ out) {
    __shared__ double s_tot[256];
    double tot = 0.0;
    for (int i = threadIdx.x; i < F; i += blockDim.x) {
        const double p = acc[i] / (double)V;
        per[i] = (float)p;
        tot += p * p;
        acc[i] = 0.0;
    }
    s_tot[threadIdx.x] = tot;
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) {
        if ((int)threadIdx.x < o) s_tot[threadIdx.x] += s_tot[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)((double)weight * s_tot[0] / (double)F);
}

// Backward: d value / d x = L^T y,  y_v = c_f * lap_v / ||lap_v||,  c_f = upstream * weight * 2 per_f / (F V)  (0 where lap_v = 0,
// as torch's norm does); y is formed on the fly from the saved Laplacian inside the transposed gather.
__global__ void __launch_bounds__(256) k_lap_penalty_bwd(const float *__restrict__ lap, const int32_t *__restrict__ nbr,
                                                         const float *__restrict__ inv_deg, const float *__restrict__ per,
                                                         const float *__restrict__ upstream, int F, int V, int D, float weight,
                                                         float *__restrict__ grad_x) {
    const int f = blockIdx.y;
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    const bool ok = v < V;
    const float *lf = lap + (size_t)f * V * 3;
    const float cf = upstream[0] * weight * 2.0f * per[f] / ((float)F * (float)V);
    auto y = [&](int n, float &a, float &b, float &c) {
        const float lx = lf[3 * n], ly = lf[3 * n + 1], lz = lf[3 * n + 2];
        const float nr = sqrtf(lx * lx + ly * ly + lz * lz);
        const float s = nr > 0.0f ? cf / nr : 0.0f;
        a = s * lx; b = s * ly; c = s * lz;
    };
    float sx, sy, sz, yx, yy, yz;
    ring_sum(nbr, inv_deg, V, D, 1, v, ok, y, sx, sy, sz);
    if (!ok) return;
    y(v, yx, yy, yz);
    float *o = grad_x + ((size_t)f * V + v) * 3;
    o[0] = sx - yx; o[1] = sy - yy; o[2] = sz - yz;
}

// ---------------------------------------------------------------------------------------------
// mvp[f,c] = P_c . Rt(q_f, t_f) . (Rt(q_c, t_c) . MV_c)      reference fit.py:541-553, camera.py:117-132 and
// roma.unitquat_to_rotmat (XYZW, applied WITHOUT normalisation, as the reference does after its whole-tensor
// "renormalisation", fit.py:616-618).  One thread per (frame, camera); ~120 tiny torch launches otherwise.
struct M4 { float m[4][4]; };
__device__ __forceinline__ M4 m4_load(const float *p) {
    M4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) r.m[i][j] = p[4 * i + j];
    return r;
}
__device__ __forceinline__ M4 m4_mul(const M4 &a, const M4 &b) {
    M4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float s = 0.0f;
#pragma unroll
            for (int k = 0; k < 4; ++k) s += a.m[i][k] * b.m[k][j];
            r.m[i][j] = s;
        }
    return r;
}
__device__ __forceinline__ M4 m4_mul_tn(const M4 &a, const M4 &b) {   // a^T b
    M4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float s = 0.0f;
#pragma unroll
            for (int k = 0; k < 4; ++k) s += a.m[k][i] * b.m[k][j];
            r.m[i][j] = s;
        }
    return r;
}
__device__ __forceinline__ M4 m4_mul_nt(const M4 &a, const M4 &b) {   // a b^T
    M4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float s = 0.0f;
#pragma unroll
            for (int k = 0; k < 4; ++k) s += a.m[i][k] * b.m[j][k];
            r.m[i][j] = s;
        }
    return r;
}
__device__ __forceinline__ M4 rigid(const float *q, const float *t) {
    const float x = q[0], y = q[1], z = q[2], w = q[3];
    const float tx = 2.0f * x, ty = 2.0f * y, tz = 2.0f * z;
    const float twx = tx * w, twy = ty * w, twz = tz * w;
    const float txx = tx * x, txy = ty * x, txz = tz * x;
    const float tyy = ty * y, tyz = tz * y, tzz = tz * z;
    M4 r;
    r.m[0][0] = 1.0f - (tyy + tzz); r.m[0][1] = txy - twz;          r.m[0][2] = txz + twy;          r.m[0][3] = t[0];
    r.m[1][0] = txy + twz;          r.m[1][1] = 1.0f - (txx + tzz); r.m[1][2] = tyz - twx;          r.m[1][3] = t[1];
    r.m[2][0] = txz - twy;          r.m[2][1] = tyz + twx;          r.m[2][2] = 1.0f - (txx + tyy); r.m[2][3] = t[2];
    r.m[3][0] = 0.0f; r.m[3][1] = 0.0f; r.m[3][2] = 0.0f; r.m[3][3] = 1.0f;
    return r;
}
// dL/d(q, t) from dL/d rigid(q, t): adds 7 values
__device__ __forceinline__ void rigid_bwd(const float *q, const M4 &G, float *gq, float *gt) {
    const float x = q[0], y = q[1], z = q[2], w = q[3];
    const float (*g)[4] = G.m;
    const float gx = 2.0f * (y * (g[0][1] + g[1][0]) + z * (g[0][2] + g[2][0]) + w * (g[2][1] - g[1][2])) - 4.0f * x * (g[1][1] + g[2][2]);
    const float gy = 2.0f * (x * (g[0][1] + g[1][0]) + z * (g[1][2] + g[2][1]) + w * (g[0][2] - g[2][0])) - 4.0f * y * (g[0][0] + g[2][2]);
    const float gz = 2.0f * (x * (g[0][2] + g[2][0]) + y * (g[1][2] + g[2][1]) + w * (g[1][0] - g[0][1])) - 4.0f * z * (g[0][0] + g[1][1]);
    const float gw = 2.0f * (z * (g[1][0] - g[0][1]) + y * (g[0][2] - g[2][0]) + x * (g[2][1] - g[1][2]));
    atomicAdd(gq + 0, gx); atomicAdd(gq + 1, gy); atomicAdd(gq + 2, gz); atomicAdd(gq + 3, gw);
    atomicAdd(gt + 0, g[0][3]); atomicAdd(gt + 1, g[1][3]); atomicAdd(gt + 2, g[2][3]);
}

__global__ void k_mvp_fwd(const float *__restrict__ proj, const float *__restrict__ t_mv, const float *__restrict__ q_cam,
                          const float *__restrict__ t_cam, const float *__restrict__ q_frame, const float *__restrict__ t_frame,
                          int Fb, int Nc, float *__restrict__ mvp) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Fb * Nc) return;
    const int f = i / Nc, c = i - f * Nc;
    const M4 B = m4_mul(rigid(q_cam + 4 * c, t_cam + 3 * c), m4_load(t_mv + 16 * c));
    const M4 X = m4_mul(rigid(q_frame + 4 * f, t_frame + 3 * f), B);
    const M4 M = m4_mul(m4_load(proj + 16 * c), X);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int k = 0; k < 4; ++k) mvp[(size_t)i * 16 + 4 * r + k] = M.m[r][k];
}

__global__ void k_mvp_bwd(const float *__restrict__ proj, const float *__restrict__ t_mv, const float *__restrict__ q_cam,
                          const float *__restrict__ t_cam, const float *__restrict__ q_frame, const float *__restrict__ t_frame,
                          const float *__restrict__ g_mvp, int Fb, int Nc, float *__restrict__ gq_cam, float *__restrict__ gt_cam,
                          float *__restrict__ gq_frame, float *__restrict__ gt_frame) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Fb * Nc) return;
    const int f = i / Nc, c = i - f * Nc;
    const M4 MV = m4_load(t_mv + 16 * c);
    const M4 B = m4_mul(rigid(q_cam + 4 * c, t_cam + 3 * c), MV);
    const M4 A = rigid(q_frame + 4 * f, t_frame + 3 * f);
    const M4 gX = m4_mul_tn(m4_load(proj + 16 * c), m4_load(g_mvp + (size_t)i * 16));   // P^T dL/dM
    const M4 gA = m4_mul_nt(gX, B);                                                       // dL/dX B^T
    const M4 gC = m4_mul_nt(m4_mul_tn(A, gX), MV);                                        // (A^T dL/dX) MV^T
    rigid_bwd(q_frame + 4 * f, gA, gq_frame + 4 * f, gt_frame + 3 * f);
    rigid_bwd(q_cam + 4 * c, gC, gq_cam + 4 * c, gt_cam + 3 * c);
}

// The same with the frames / views of the step given as INDEX arrays into the full parameter tables (the reference's run shape draws one
// random (camera, frame) per step: as torch ops that was seven index_select launches forward and four zero-fill + index_add pairs backward
// around kernels that take 3 us).  frame_idx [Fb] or null (frames 0 .. Fb - 1); view_idx [Nc] or null: rows of proj / t_mv; cam_of_view
// [views] or null: row of q_cam / t_cam for a view (the cameras a Fitter was given).  The backward adds into the FULL tables (zeroed by
// the caller); two step entries that name the same row add into it, as index_select's backward does.
__device__ __forceinline__ void mvp_rows(const long long *frame_idx, const long long *view_idx, const long long *cam_of_view, int f, int c,
                                         int &fr, int &cv, int &cp) {
    fr = frame_idx ? (int)frame_idx[f] : f;
    cv = view_idx ? (int)view_idx[c] : c;
    cp = cam_of_view ? (int)cam_of_view[cv] : cv;
}
__global__ void k_mvp_fwd_idx(const float *__restrict__ proj, const float *__restrict__ t_mv, const float *__restrict__ q_cam,
                              const float *__restrict__ t_cam, const float *__restrict__ q_frame, const float *__restrict__ t_frame,
                              const long long *__restrict__ frame_idx, const long long *__restrict__ view_idx,
                              const long long *__restrict__ cam_of_view, int Fb, int Nc, float *__restrict__ mvp) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Fb * Nc) return;
    const int f = i / Nc, c = i - f * Nc;
    int fr, cv, cp;
    mvp_rows(frame_idx, view_idx, cam_of_view, f, c, fr, cv, cp);
    const M4 B = m4_mul(rigid(q_cam + 4 * cp, t_cam + 3 * cp), m4_load(t_mv + 16 * cv));
    const M4 X = m4_mul(rigid(q_frame + 4 * fr, t_frame + 3 * fr), B);
    const M4 M = m4_mul(m4_load(proj + 16 * cv), X);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int k = 0; k < 4; ++k) mvp[(size_t)i * 16 + 4 * r + k] = M.m[r][k];
}
__global__ void k_mvp_bwd_idx(const float *__restrict__ proj, const float *__restrict__ t_mv, const float *__restrict__ q_cam,
                              const float *__restrict__ t_cam, const float *__restrict__ q_frame, const float *__restrict__ t_frame,
                              const long long *__restrict__ frame_idx, const long long *__restrict__ view_idx,
                              const long long *__restrict__ cam_of_view, const float *__restrict__ g_mvp, int Fb, int Nc,
                              float *__restrict__ gq_cam, float *__restrict__ gt_cam, float *__restrict__ gq_frame, float *__restrict__ gt_frame) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Fb * Nc) return;
    const int f = i / Nc, c = i - f * Nc;
    int fr, cv, cp;
    mvp_rows(frame_idx, view_idx, cam_of_view, f, c, fr, cv, cp);
    const M4 MV = m4_load(t_mv + 16 * cv);
    const M4 B = m4_mul(rigid(q_cam + 4 * cp, t_cam + 3 * cp), MV);
    const M4 A = rigid(q_frame + 4 * fr, t_frame + 3 * fr);
    const M4 gX = m4_mul_tn(m4_load(proj + 16 * cv), m4_load(g_mvp + (size_t)i * 16));
    const M4 gA = m4_mul_nt(gX, B);
    const M4 gC = m4_mul_nt(m4_mul_tn(A, gX), MV);
    rigid_bwd(q_frame + 4 * fr, gA, gq_frame + 4 * fr, gt_frame + 3 * fr);
    rigid_bwd(q_cam + 4 * cp, gC, gq_cam + 4 * cp, gt_cam + 3 * cp);
}

}  // namespace

extern "C" int fpcdr_mvp_fwd_indexed(const float *proj, const float *t_mv, const float *q_cam, const float *t_cam, const float *q_frame,
                                     const float *t_frame, const int64_t *frame_idx, const int64_t *view_idx, const int64_t *cam_of_view,
                                     float *mvp, int32_t Fb, int32_t Nc, void *stream) {
    FPCDR_REQUIRE(proj && t_mv && q_cam && t_cam && q_frame && t_frame && mvp, "null pointer");
    FPCDR_REQUIRE(Fb > 0 && Nc > 0, "bad sizes");
    hipLaunchKernelGGL(k_mvp_fwd_idx, dim3(fpcdr_cdiv((long long)Fb * Nc, 64)), dim3(64), 0, (hipStream_t)stream, proj, t_mv, q_cam, t_cam,
                       q_frame, t_frame, (const long long *)frame_idx, (const long long *)view_idx, (const long long *)cam_of_view, Fb, Nc, mvp);
    FPCDR_CHECK_LAUNCH();
    return FPCDR_OK;
}

extern "C" int fpcdr_mvp_bwd_indexed(const float *proj, const float *t_mv, const float *q_cam, const float *t_cam, const float *q_frame,
                                     const float *t_frame, const int64_t *frame_idx, const int64_t *view_idx, const int64_t *cam_of_view,
                                     const float *grad_mvp, float *gq_cam, float *gt_cam, float *gq_frame, float *gt_frame, int32_t Fb,
                                     int32_t Nc, void *stream) {
    FPCDR_REQUIRE(proj && t_mv && q_cam && t_cam && q_frame && t_frame && grad_mvp && gq_cam && gt_cam && gq_frame && gt_frame,
                  "null pointer");
    FPCDR_REQUIRE(Fb > 0 && Nc > 0, "bad sizes");
    hipLaunchKernelGGL(k_mvp_bwd_idx, dim3(fpcdr_cdiv((long long)Fb * Nc, 64)), dim3(64), 0, (hipStream_t)stream, proj, t_mv, q_cam, t_cam,
                       q_frame, t_frame, (const long long *)frame_idx, (const long long *)view_idx, (const long long *)cam_of_view, grad_mvp,
                       Fb, Nc, gq_cam, gt_cam, gq_frame, gt_frame);
    FPCDR_CHECK_LAUNCH();
    return FPCDR_OK;
}

extern "C" int fpcdr_mvp_fwd(const float *proj, const float *t_mv, const float *q_cam, const float *t_cam, const float *q_frame,
                             const float *t_frame, float *mvp, int32_t Fb, int32_t Nc, void *stream) {
    FPCDR_REQUIRE(proj && t_mv && q_cam && t_cam && q_frame && t_frame && mvp, "null pointer");
    FPCDR_REQUIRE(Fb > 0 && Nc > 0, "bad sizes");
    hipLaunchKernelGGL(k_mvp_fwd, dim3(fpcdr_cdiv((long long)Fb * Nc, 64)), dim3(64), 0, (hipStream_t)stream, proj, t_mv, q_cam,
                       t_cam, q_frame, t_frame, Fb, Nc, mvp);
    FPCDR_CHECK_LAUNCH();
    return FPCDR_OK;
}

extern "C" int fpcdr_mvp_bwd(const float *proj, const float *t_mv, const float *q_cam, const float *t_cam, const float *q_frame,
                             const float *t_frame, const float *grad_mvp, float *gq_cam, float *gt_cam, float *gq_frame,
                             float *gt_frame, int32_t Fb, int32_t Nc, void *stream) {
    FPCDR_REQUIRE(proj && t_mv && q_cam && t_cam && q_frame && t_frame && grad_mvp && gq_cam && gt_cam && gq_frame && gt_frame,
                  "null pointer");
    FPCDR_REQUIRE(Fb > 0 && Nc > 0, "bad sizes");
    hipLaunchKernelGGL(k_mvp_bwd, dim3(fpcdr_cdiv((long long)Fb * Nc, 64)), dim3(64), 0, (hipStream_t)stream, proj, t_mv, q_cam,
                       t_cam, q_frame, t_frame, grad_mvp, Fb, Nc, gq_cam, gt_cam, gq_frame, gt_frame);
    FPCDR_CHECK_LAUNCH();
    return FPCDR_OK;
}

extern "C" int fpcdr_laplacian_gather(const float *x, const int32_t *nbr, const float *inv_deg, float *out, int32_t F, int32_t V,
                                      int32_t D, int32_t transpose, void *stream) {
    FPCDR_REQUIRE(x && nbr && inv_deg && out, "null pointer");
    FPCDR_REQUIRE(F > 0 && V > 0 && D > 0 && F <= 65535, "bad sizes");
    hipLaunchKernelGGL(k_lap_gather, dim3(fpcdr_cdiv(V, 256), F), dim3(256), 0, (hipStream_t)stream, x, nbr, inv_deg, V, D,
                       transpose ? 1 : 0, out);
    FPCDR_CHECK_LAUNCH();
    return FPCDR_OK;
}

extern "C" int fpcdr_laplacian_penalty_fwd(const float *x, const int32_t *nbr, const float *inv_deg, float *lap, void *acc, float *per,
                                           float *out, float weight, int32_t F, int32_t V, int32_t D, void *stream) {
    FPCDR_REQUIRE(x && nbr && inv_deg && lap && acc && per && out, "null pointer");
    FPCDR_REQUIRE(F > 0 && V > 0 && D > 0 && F <= 65535, "bad sizes");
    FPCDR_REQUIRE(((size_t)acc & 7) == 0, "acc must be 8-byte aligned");
    // acc: F doubles (+ one spare 8-byte slot: the size of earlier ABI versions), zero on entry and zero again when the call has run
    hipLaunchKernelGGL(k_lap_penalty_fwd, dim3(fpcdr_cdiv(V, 256), F), dim3(256), 0, (hipStream_t)stream, x, nbr, inv_deg, F, V, D,
                       lap, (double *)acc);
    hipLaunchKernelGGL(k_lap_penalty_finish, dim3(1), dim3(256), 0, (hipStream_t)stream, (double *)acc, F, V, weight, per, out);
    FPCDR_CHECK_LAUNCH();
    return FPCDR_OK;
}

extern "C" int fpcdr_laplacian_penalty_bwd(const float *lap, const int32_t *nbr, const float *inv_deg, const float *per,
                                           const float *upstream, float *grad_x, float weight, int32_t F, int32_t V, int32_t D,
                                           void *stream) {
    FPCDR_REQUIRE(lap && nbr && inv_deg && per && upstream && grad_x, "null pointer");
    FPCDR_REQUIRE(F > 0 && V > 0 && D > 0 && F <= 65535, "bad sizes");
    hipLaunchKernelGGL(k_lap_penalty_bwd, dim3(fpcdr_cdiv(V, 256), F), dim3(256), 0, (hipStream_t)stream, lap, nbr, inv_deg, per, upstream,
                       F, V, D, weight, grad_x);
    FPCDR_CHECK_LAUNCH();
    return FPCDR_OK;
}

extern "C" int fpcdr_transform_clip_fwd(const float *mvp, const float *verts, float *out, int32_t F, int32_t Nc, int32_t V,
                                        void *stream) {
    FPCDR_REQUIRE(mvp && verts && out, "null pointer");
    FPCDR_REQUIRE(F > 0 && Nc > 0 && V > 0 && (long long)F * Nc <= 65535, "bad sizes");
    hipLaunchKernelGGL(k_clip_fwd, dim3(fpcdr_cdiv(V, 256), F * Nc), dim3(256), 0, (hipStream_t)stream, mvp, verts, V, Nc,
                       (float4 *)out);
    FPCDR_CHECK_LAUNCH();
    return FPCDR_OK;
}

extern "C" int fpcdr_transform_clip_bwd(const float *mvp, const float *verts, const float *grad_out, float *grad_verts,
                                        float *grad_mvp, int32_t F, int32_t Nc, int32_t V, void *stream) {
    FPCDR_REQUIRE(mvp && verts && grad_out, "null pointer");
    FPCDR_REQUIRE(F > 0 && Nc > 0 && V > 0 && (long long)F * Nc <= 65535, "bad sizes");
    hipStream_t st = (hipStream_t)stream;
    if (grad_verts && grad_mvp) {
        hipLaunchKernelGGL(k_clip_bwd_both, dim3(fpcdr_cdiv(V, 256 * CLIP_VPT), F), dim3(256), 0, st, mvp, verts, (const float4 *)grad_out, V, Nc,
                           grad_verts, grad_mvp);
        FPCDR_CHECK_LAUNCH();
        return FPCDR_OK;
    }
    if (grad_verts)
        hipLaunchKernelGGL(k_clip_bwd_verts, dim3(fpcdr_cdiv(V, 256), F), dim3(256), 0, st, mvp, (const float4 *)grad_out, V, Nc,
                           grad_verts);
    if (grad_mvp)
        hipLaunchKernelGGL(k_clip_bwd_mvp, dim3(fpcdr_cdiv(V, MVP_VPB), F * Nc), dim3(256), 0, st, verts, (const float4 *)grad_out, V,
                           Nc, grad_mvp);
    FPCDR_CHECK_LAUNCH();
    return FPCDR_OK;
}
