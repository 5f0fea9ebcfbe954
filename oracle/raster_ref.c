/*
 * oracle/raster_ref.c -- TEST INFRASTRUCTURE ONLY (never shipped, never on the product path).
 *
 * CPU restatement of the visibility stage of `dr.rasterize` as called at
 * reference src/torch/fit.py:151 (context created at fit.py:484).  The reference
 * gets this stage from nvdiffrast + the OpenGL hardware rasteriser, neither of
 * which exists in /root/reference -- PARITY UNPINNED: the
 * sampling / fill / depth rules below are this build's own specification
 * (DESIGN.md "Raster rules"), stated here once in plain scalar C and once more,
 * independently, in the HIP kernel.  Integer outputs (triangle ids, coverage)
 * must agree bit for bit between the two.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * the library built from this file.
 *
 * Rules (all arithmetic IEEE-754, no fused multiply-add: build with
 * -ffp-contract=off):
 *   R1  a triangle whose three w are > 0 is rasterised as it is.  Any other triangle is CLIPPED against the near plane
 *       z + w >= 0 (Sutherland-Hodgman in double; a crossing is always computed from the vertex inside to the one outside:
 *       t = d_in / (d_in - d_out), p = in + t (out - in) for x, y, z, w, so the two triangles of a shared edge get the same
 *       point).  The polygon left (3 or 4 vertices, fanned from its first vertex into 1 or 2 pieces) is rasterised piece by
 *       piece under the triangle's OWN index if every polygon vertex has w > 0; otherwise (and for NaNs) the triangle is
 *       dropped.  Float outputs (barycentrics, z/w) always come from the three original vertices;
 *   R2  vertex -> fixed point, 8 sub-pixel bits, in double, with r = 1.0 / w (one division per vertex):
 *         X = floor(((x * r) * 0.5 + 0.5) * (W*256) + 0.5), same for Y with H;
 *       dropped if any |X|,|Y| > 2^24 (guard band);
 *   R3  D = (X1-X0)(Y2-Y0) - (Y1-Y0)(X2-X0); D == 0 dropped; back faces are kept
 *       (edge functions are multiplied by sign(D));
 *   R4  pixel (px,py) is sampled at P = (256 px + 128, 256 py + 128); row 0 is the
 *       bottom row (OpenGL convention, reference fit.py:532 flips images on load);
 *   R5  E_ab(P) = s[(Xb-Xa)(Py-Ya) - (Yb-Ya)(Px-Xa)] for (a,b) = (1,2),(2,0),(0,1);
 *       covered iff every E >= 0, where an edge with E == 0 counts only if its
 *       normalised direction (dx,dy) has dy > 0, or dy == 0 and dx < 0;
 *   R6  depth is a float32 plane through the snapped vertices, anchored at vertex 0 (like a
 *       24-bit hardware depth buffer): with zw_i = z_i * r_i in double,
 *         zA = ((zw1-zw0)(Y2-Y0) - (zw2-zw0)(Y1-Y0)) / D,  zB = ((zw2-zw0)(X1-X0) - (zw1-zw0)(X2-X0)) / D
 *       rounded to float, depth(P) = fmaf(zA, (float)(Px-X0), fmaf(zB, (float)(Py-Y0), (float)zw0));
 *       fragments with depth outside [-1,1] are discarded; the smaller depth wins, ties
 *       go to the smaller triangle index.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define SUBPIX 256
#define HALFPIX 128
#define GUARD 16777216.0 /* 2^24 */

typedef struct {
    int64_t A[3], B[3], C[3];
    int own[3];
    float zA, zB, z0;
    int64_t X0, Y0;
    int px0, px1, py0, py1;
} tri_setup_t;

static int64_t floordiv(int64_t a, int64_t b) { /* b > 0 */
    int64_t q = a / b;
    if ((a % b != 0) && (a < 0)) q -= 1;
    return q;
}

/* R1: the pieces of a triangle (vertices in double).  Returns their number (0 = dropped, 1, or 2). */
static int clip_pieces(const float *v0, const float *v1, const float *v2, double pc[2][3][4]) {
    const float *v[3] = {v0, v1, v2};
    if (v0[3] > 0.0f && v1[3] > 0.0f && v2[3] > 0.0f) {
        for (int i = 0; i < 3; ++i)
            for (int c = 0; c < 4; ++c) pc[0][i][c] = (double)v[i][c];
        return 1;
    }
    double d[3], poly[4][4];
    int n = 0;
    for (int i = 0; i < 3; ++i) {
        d[i] = (double)v[i][2] + (double)v[i][3];
        if (!(d[i] == d[i])) return 0; /* NaN */
    }
    for (int i = 0; i < 3; ++i) {
        int j = (i + 1) % 3;
        int in_i = d[i] >= 0.0, in_j = d[j] >= 0.0;
        if (in_i) {
            for (int c = 0; c < 4; ++c) poly[n][c] = (double)v[i][c];
            ++n;
        }
        if (in_i != in_j) { /* the edge crosses the plane: from the inside vertex to the outside one */
            int a = in_i ? i : j, b = in_i ? j : i;
            double t = d[a] / (d[a] - d[b]);
            for (int c = 0; c < 4; ++c) poly[n][c] = (double)v[a][c] + t * ((double)v[b][c] - (double)v[a][c]);
            ++n;
        }
    }
    if (n < 3) return 0;
    for (int i = 0; i < n; ++i)
        if (!(poly[i][3] > 0.0)) return 0;
    for (int k = 0; k + 2 < n; ++k) { /* fan */
        memcpy(pc[k][0], poly[0], sizeof(double) * 4);
        memcpy(pc[k][1], poly[k + 1], sizeof(double) * 4);
        memcpy(pc[k][2], poly[k + 2], sizeof(double) * 4);
    }
    return n - 2;
}

/* returns 0 if the piece is dropped */
static int setup_triangle(double v[3][4], int H, int W, tri_setup_t *ts) {
    int64_t X[3], Y[3];
    double zw[3];
    for (int i = 0; i < 3; ++i) {
        double w = v[i][3];
        if (!(w > 0.0)) return 0; /* (also rejects NaN) */
        double rw = 1.0 / w; /* R2: ONE division per vertex; x, y and z are multiplied by the reciprocal */
        double xs = v[i][0] * rw;
        double ys = v[i][1] * rw;
        double fx = floor((xs * 0.5 + 0.5) * (double)(W * SUBPIX) + 0.5);
        double fy = floor((ys * 0.5 + 0.5) * (double)(H * SUBPIX) + 0.5);
        if (!(fabs(fx) <= GUARD) || !(fabs(fy) <= GUARD)) return 0; /* R2 */
        X[i] = (int64_t)fx;
        Y[i] = (int64_t)fy;
        zw[i] = v[i][2] * rw;
    }
    int64_t D = (X[1] - X[0]) * (Y[2] - Y[0]) - (Y[1] - Y[0]) * (X[2] - X[0]);
    if (D == 0) return 0; /* R3 */
    int64_t s = D > 0 ? 1 : -1;
    double Dd = (double)D;
    double dz1 = zw[1] - zw[0], dz2 = zw[2] - zw[0];
    ts->zA = (float)((dz1 * (double)(Y[2] - Y[0]) - dz2 * (double)(Y[1] - Y[0])) / Dd);
    ts->zB = (float)((dz2 * (double)(X[1] - X[0]) - dz1 * (double)(X[2] - X[0])) / Dd);
    ts->z0 = (float)zw[0];
    ts->X0 = X[0];
    ts->Y0 = Y[0];
    int64_t xmin = X[0], xmax = X[0], ymin = Y[0], ymax = Y[0];
    for (int i = 1; i < 3; ++i) {
        if (X[i] < xmin) xmin = X[i];
        if (X[i] > xmax) xmax = X[i];
        if (Y[i] < ymin) ymin = Y[i];
        if (Y[i] > ymax) ymax = Y[i];
    }
    /* pixel centres with xmin <= 256 px + 128 <= xmax */
    int64_t px0 = floordiv(xmin - HALFPIX + SUBPIX - 1, SUBPIX);
    int64_t px1 = floordiv(xmax - HALFPIX, SUBPIX);
    int64_t py0 = floordiv(ymin - HALFPIX + SUBPIX - 1, SUBPIX);
    int64_t py1 = floordiv(ymax - HALFPIX, SUBPIX);
    if (px0 < 0) px0 = 0;
    if (py0 < 0) py0 = 0;
    if (px1 > W - 1) px1 = W - 1;
    if (py1 > H - 1) py1 = H - 1;
    if (px0 > px1 || py0 > py1) return 0;
    ts->px0 = (int)px0; ts->px1 = (int)px1; ts->py0 = (int)py0; ts->py1 = (int)py1;
    static const int ea[3] = {1, 2, 0}, eb[3] = {2, 0, 1};
    for (int e = 0; e < 3; ++e) {
        int a = ea[e], b = eb[e];
        int64_t A = -(Y[b] - Y[a]) * s;
        int64_t Bc = (X[b] - X[a]) * s;
        ts->A[e] = A;
        ts->B[e] = Bc;
        ts->C[e] = -(A * X[a] + Bc * Y[a]);
        int64_t dx = Bc, dy = -A;
        ts->own[e] = (dy > 0) || (dy == 0 && dx < 0); /* R5 tie rule */
    }
    return 1;
}

/*
 * pos  [B][V][4] float32 clip-space positions
 * tri  [T][3]    int32
 * out_id    [B][H][W] int32   triangle index + 1, 0 = empty
 * out_depth [B][H][W] float   winning depth (may be NULL)
 */
int fpcdr_oracle_rasterize_ids(const float *pos, const int32_t *tri, int B, int V, int T, int H,
                               int W, int32_t *out_id, float *out_depth) {
    size_t npix = (size_t)H * W;
    float *zbuf = (float *)malloc(npix * sizeof(float));
    if (!zbuf) return 1;
    for (int b = 0; b < B; ++b) {
        const float *p = pos + (size_t)b * V * 4;
        int32_t *ids = out_id + (size_t)b * npix;
        memset(ids, 0, npix * sizeof(int32_t));
        for (size_t i = 0; i < npix; ++i) zbuf[i] = INFINITY;
        for (int t = 0; t < T; ++t) {
            int i0 = tri[3 * t], i1 = tri[3 * t + 1], i2 = tri[3 * t + 2];
            if (i0 < 0 || i0 >= V || i1 < 0 || i1 >= V || i2 < 0 || i2 >= V) continue;
            double pc[2][3][4];
            const int npieces = clip_pieces(p + 4 * (size_t)i0, p + 4 * (size_t)i1, p + 4 * (size_t)i2, pc);
            for (int piece = 0; piece < npieces; ++piece) {
            tri_setup_t ts;
            if (!setup_triangle(pc[piece], H, W, &ts))
                continue;
            for (int py = ts.py0; py <= ts.py1; ++py) {
                int64_t Py = (int64_t)py * SUBPIX + HALFPIX;
                for (int px = ts.px0; px <= ts.px1; ++px) {
                    int64_t Px = (int64_t)px * SUBPIX + HALFPIX;
                    int64_t E[3];
                    int inside = 1;
                    for (int e = 0; e < 3; ++e) {
                        E[e] = ts.A[e] * Px + ts.B[e] * Py + ts.C[e];
                        if (E[e] < 0 || (E[e] == 0 && !ts.own[e])) inside = 0;
                    }
                    if (!inside) continue;
                    float depth = fmaf(ts.zA, (float)(Px - ts.X0), fmaf(ts.zB, (float)(Py - ts.Y0), ts.z0)); /* R6 */
                    if (!(depth >= -1.0f && depth <= 1.0f)) continue;
                    size_t o = (size_t)py * W + px;
                    if (depth < zbuf[o]) {
                        zbuf[o] = depth;
                        ids[o] = t + 1;
                    }
                }
            }
            }
        }
        if (out_depth) memcpy(out_depth + (size_t)b * npix, zbuf, npix * sizeof(float));
    }
    free(zbuf);
    return 0;
}

/*
 * Edge -> opposite-vertex table used by the antialias restatement (reference call
 * site fit.py:160).  For every undirected edge (a<b) of `tri`: number of incident
 * triangles and the sum of their opposite vertex indices (the "other" opposite
 * vertex of a 2-triangle edge is sum - own).  Output arrays are per triangle edge,
 * edge e of triangle t joins vertices (e+1)%3 and (e+2)%3 (so e is also the index
 * of the triangle's own opposite vertex).
 *   out_count [T][3] int32, out_other [T][3] int32 (-1 unless count == 2)
 * Simple O(T log T) sort-based build.
 */
typedef struct { int32_t a, b, opp, slot; } edge_rec_t;
static int edge_cmp(const void *x, const void *y) {
    const edge_rec_t *p = (const edge_rec_t *)x, *q = (const edge_rec_t *)y;
    if (p->a != q->a) return p->a < q->a ? -1 : 1;
    if (p->b != q->b) return p->b < q->b ? -1 : 1;
    return p->slot < q->slot ? -1 : (p->slot > q->slot);
}
int fpcdr_oracle_edge_table(const int32_t *tri, int T, int32_t *out_count, int32_t *out_other) {
    edge_rec_t *r = (edge_rec_t *)malloc((size_t)T * 3 * sizeof(edge_rec_t));
    if (!r) return 1;
    for (int t = 0; t < T; ++t)
        for (int e = 0; e < 3; ++e) {
            int32_t a = tri[3 * t + (e + 1) % 3], b = tri[3 * t + (e + 2) % 3];
            edge_rec_t *x = &r[3 * t + e];
            x->a = a < b ? a : b;
            x->b = a < b ? b : a;
            x->opp = tri[3 * t + e];
            x->slot = 3 * t + e;
        }
    qsort(r, (size_t)T * 3, sizeof(edge_rec_t), edge_cmp);
    size_t n = (size_t)T * 3, i = 0;
    while (i < n) {
        size_t j = i;
        int64_t sum = 0;
        while (j < n && r[j].a == r[i].a && r[j].b == r[i].b) { sum += r[j].opp; ++j; }
        int32_t cnt = (int32_t)(j - i);
        for (size_t k = i; k < j; ++k) {
            out_count[r[k].slot] = cnt;
            out_other[r[k].slot] = (cnt == 2) ? (int32_t)(sum - r[k].opp) : -1;
        }
        i = j;
    }
    free(r);
    return 0;
}
