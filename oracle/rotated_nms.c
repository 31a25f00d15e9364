/*
 * CPU oracle, C part: rotated NMS for the CoAlign detection hot path.
 *
 * TEST INFRASTRUCTURE ONLY -- compiled by oracle/coalign_oracle.py:build_c() (gcc -O2 -ffp-contract=off)
 * into oracle/_build/librotated_nms.so; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg load it.  The shipped library (coalign_amd/csrc) never links or calls it.
 *
 * (1) oracle_nms_rotated / oracle_quad_iou
 *     Restates opencood/utils/box_utils.py:693-738 (nms_rotated) with the polygon IoU of
 *     opencood/utils/common_utils.py:196-236 (compute_iou / convert_format):
 *       polygon  = corners 0..3, (x, y) columns, float32 promoted to float64   (common_utils.py:234-236)
 *       order    = argsort(scores)[::-1][:1000]                               (box_utils.py:719-721)
 *       iou      = float32( inter.area / union.area )                         (common_utils.py:216-218)
 *       suppress = iou > threshold, threshold compared in float32             (box_utils.py:733)
 *     The area arithmetic itself is Shapely 2.0.0 -> GEOS (requirements.txt:13), which is not vendored
 *     in the reference and not installed in the build container: PARITY UNPINNED for that arithmetic.
 *     It is restated as Sutherland-Hodgman clipping of quad A against the half-planes of quad B in
 *     float64, union = |A| + |B| - inter.
 *
 * (2) oracle_pcdet_overlap / oracle_pcdet_iou / oracle_pcdet_nms
 *     Restates the OpenPCDet fp32 BEV overlap used by opencood/pcdet_utils/iou3d_nms
 *     (src/iou3d_cpu.cpp:128-229 box_overlap, :231-238 iou_bev; bitmask NMS semantics of
 *     src/iou3d_nms_kernel.cu:267-311 + host scan src/iou3d_nms.cpp:121-133).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ (1) fp64 convex clipping ---- */

typedef struct { double x, y; } P2;

static double signed_area(const P2 *p, int n) {
    double s = 0.0;
    for (int i = 0; i < n; ++i) {
        const P2 a = p[i], b = p[(i + 1 == n) ? 0 : i + 1];
        s += a.x * b.y - b.x * a.y;
    }
    return 0.5 * s;
}

/* Clip convex polygon `subj` (n vertices) by the half-plane to the left of directed edge q0->q1
 * (orientation factor `sgn` = +1 for a CCW clip polygon, -1 for CW).  Returns new vertex count. */
static int clip_halfplane(const P2 *subj, int n, P2 q0, P2 q1, double sgn, P2 *out) {
    int m = 0;
    const double ex = q1.x - q0.x, ey = q1.y - q0.y;
    for (int i = 0; i < n; ++i) {
        const P2 s = subj[i], e = subj[(i + 1 == n) ? 0 : i + 1];
        const double ds = sgn * (ex * (s.y - q0.y) - ey * (s.x - q0.x));
        const double de = sgn * (ex * (e.y - q0.y) - ey * (e.x - q0.x));
        const int s_in = ds >= 0.0, e_in = de >= 0.0;
        if (s_in) out[m++] = s;
        if (s_in != e_in) {
            const double t = ds / (ds - de);
            P2 r;
            r.x = s.x + t * (e.x - s.x);
            r.y = s.y + t * (e.y - s.y);
            out[m++] = r;
        }
    }
    return m;
}

double oracle_quad_intersection_area(const double *a_xy, const double *b_xy) {
    P2 buf0[16], buf1[16], b[4];
    int n = 4;
    for (int i = 0; i < 4; ++i) {
        buf0[i].x = a_xy[2 * i]; buf0[i].y = a_xy[2 * i + 1];
        b[i].x = b_xy[2 * i];    b[i].y = b_xy[2 * i + 1];
    }
    const double sgn = signed_area(b, 4) >= 0.0 ? 1.0 : -1.0;
    P2 *src = buf0, *dst = buf1;
    for (int k = 0; k < 4 && n > 0; ++k) {
        n = clip_halfplane(src, n, b[k], b[(k + 1) & 3], sgn, dst);
        P2 *t = src; src = dst; dst = t;
    }
    if (n < 3) return 0.0;
    return fabs(signed_area(src, n));
}

double oracle_quad_iou(const double *a_xy, const double *b_xy) {
    P2 a[4], b[4];
    for (int i = 0; i < 4; ++i) {
        a[i].x = a_xy[2 * i]; a[i].y = a_xy[2 * i + 1];
        b[i].x = b_xy[2 * i]; b[i].y = b_xy[2 * i + 1];
    }
    const double inter = oracle_quad_intersection_area(a_xy, b_xy);
    const double uni = fabs(signed_area(a, 4)) + fabs(signed_area(b, 4)) - inter;
    return inter / uni; /* 0/0 -> NaN, like a zero-area union in the reference (compare is then false) */
}

typedef struct { float s; int i; } ScoreIdx;

/* ascending by score, ties ascending by index: reversed afterwards -> ties in descending index order */
static int cmp_score_idx(const void *pa, const void *pb) {
    const ScoreIdx *a = (const ScoreIdx *)pa, *b = (const ScoreIdx *)pb;
    if (a->s < b->s) return -1;
    if (a->s > b->s) return 1;
    return (a->i > b->i) - (a->i < b->i);
}

/* boxes: [K, rows, cols] float32 with rows >= 4 and cols in {2,3}: corners 0..3, columns x,y are used.
 * `row_floats` = cols.  keep: out, capacity min(K, top).  Returns number kept. */
int oracle_nms_rotated(const float *boxes, int cols, const float *scores, int K, float threshold, int top,
                       int *keep) {
    if (K <= 0) return 0;
    /* the per-box stride is rows*cols; rows is 8 for [K,8,3] and 4 for [K,4,2] */
    const int rows = (cols == 3) ? 8 : 4;
    const int stride = rows * cols;
    ScoreIdx *ord = (ScoreIdx *)malloc(sizeof(ScoreIdx) * (size_t)K);
    for (int i = 0; i < K; ++i) { ord[i].s = scores[i]; ord[i].i = i; }
    qsort(ord, (size_t)K, sizeof(ScoreIdx), cmp_score_idx);
    const int n = K < top ? K : top;
    int *ix = (int *)malloc(sizeof(int) * (size_t)n);
    for (int r = 0; r < n; ++r) ix[r] = ord[K - 1 - r].i;
    free(ord);
    double *poly = (double *)malloc(sizeof(double) * 8 * (size_t)n);
    for (int r = 0; r < n; ++r)
        for (int c = 0; c < 4; ++c) {
            poly[r * 8 + 2 * c] = (double)boxes[(size_t)ix[r] * stride + c * cols + 0];
            poly[r * 8 + 2 * c + 1] = (double)boxes[(size_t)ix[r] * stride + c * cols + 1];
        }
    unsigned char *dead = (unsigned char *)calloc((size_t)n, 1);
    int nk = 0;
    for (int r = 0; r < n; ++r) {
        if (dead[r]) continue;
        keep[nk++] = ix[r];
        for (int q = r + 1; q < n; ++q) {
            if (dead[q]) continue;
            const float iou = (float)oracle_quad_iou(poly + r * 8, poly + q * 8);
            if (iou > threshold) dead[q] = 1;
        }
    }
    free(dead); free(poly); free(ix);
    return nk;
}

/* ------------------------------------------------------------------ (2) OpenPCDet fp32 overlap ---- */

typedef struct { float x, y; } F2;

static float crs3(F2 p1, F2 p2, F2 p0) { return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y); }
static float crs2(F2 a, F2 b) { return a.x * b.y - a.y * b.x; }
static float fmin2(float a, float b) { return a > b ? b : a; }
static float fmax2(float a, float b) { return a > b ? a : b; }

static int seg_isect(F2 p1, F2 p0, F2 q1, F2 q0, F2 *ans) {
    if (!(fmin2(p0.x, p1.x) <= fmax2(q0.x, q1.x) && fmin2(q0.x, q1.x) <= fmax2(p0.x, p1.x) &&
          fmin2(p0.y, p1.y) <= fmax2(q0.y, q1.y) && fmin2(q0.y, q1.y) <= fmax2(p0.y, p1.y)))
        return 0;
    const float s1 = crs3(q0, p1, p0), s2 = crs3(p1, q1, p0), s3 = crs3(p0, q1, q0), s4 = crs3(q1, p1, q0);
    if (!(s1 * s2 > 0 && s3 * s4 > 0)) return 0;
    const float s5 = crs3(q1, p1, p0);
    if (fabsf(s5 - s1) > 1e-8f) {
        ans->x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
        ans->y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
    } else {
        const float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
        const float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
        const float D = a0 * b1 - a1 * b0;
        ans->x = (b0 * c1 - b1 * c0) / D;
        ans->y = (a1 * c0 - a0 * c1) / D;
    }
    return 1;
}

/* The float cos / sin / atan2 of row N, DEFINED (round 6): evaluated in float64 and rounded once to float32 -- an admissible cosf / sinf / atan2f (<= 0.5 ulp up to
 * ~2^-29 double-rounding cases) like the CUDA run time's behind iou3d_nms_kernel.cu or glibc's behind iou3d_cpu.cpp, and the same bits here and in the device
 * kernel (coalign_amd/csrc/nms.hip: trig_*), so the IoU matrices are compared bit for bit (tests/test_hip_parity.py::test_pcdet_iou_bev_vs_oracle). */
static float trig_cos(float a) { return (float)cos((double)a); }
static float trig_sin(float a) { return (float)sin((double)a); }
static float trig_atan2(float y, float x) { return (float)atan2((double)y, (double)x); }

static int in_box_margin(const float *box, F2 p) {
    /* iou3d_cpu.cpp:76-87; cos/sin on a float argument resolve to the float overloads in C++. */
    const float cx = box[0], cy = box[1];
    const float ca = trig_cos(-box[6]), sa = trig_sin(-box[6]);
    const float rx = (p.x - cx) * ca + (p.y - cy) * (-sa);
    const float ry = (p.x - cx) * sa + (p.y - cy) * ca;
    return fabsf(rx) < box[3] / 2 + 1e-2f && fabsf(ry) < box[4] / 2 + 1e-2f;
}

static void box_corners_f32(const float *box, F2 *c /*[5]*/) {
    const float hx = box[3] / 2, hy = box[4] / 2;
    const float x1 = box[0] - hx, y1 = box[1] - hy, x2 = box[0] + hx, y2 = box[1] + hy;
    const float ca = trig_cos(box[6]), sa = trig_sin(box[6]);
    const F2 raw[4] = {{x1, y1}, {x2, y1}, {x2, y2}, {x1, y2}};
    for (int k = 0; k < 4; ++k) {
        c[k].x = (raw[k].x - box[0]) * ca + (raw[k].y - box[1]) * (-sa) + box[0];
        c[k].y = (raw[k].x - box[0]) * sa + (raw[k].y - box[1]) * ca + box[1];
    }
    c[4] = c[0];
}

float oracle_pcdet_overlap(const float *box_a, const float *box_b) {
    F2 A[5], B[5], pts[16], ctr = {0.f, 0.f};
    int cnt = 0;
    box_corners_f32(box_a, A);
    box_corners_f32(box_b, B);
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            if (seg_isect(A[i + 1], A[i], B[j + 1], B[j], &pts[cnt])) {
                ctr.x += pts[cnt].x; ctr.y += pts[cnt].y; ++cnt;
            }
    for (int k = 0; k < 4; ++k) {
        if (in_box_margin(box_a, B[k])) { ctr.x += B[k].x; ctr.y += B[k].y; pts[cnt++] = B[k]; }
        if (in_box_margin(box_b, A[k])) { ctr.x += A[k].x; ctr.y += A[k].y; pts[cnt++] = A[k]; }
    }
    ctr.x /= cnt; ctr.y /= cnt;
    for (int j = 0; j < cnt - 1; ++j)           /* bubble sort by polar angle, as the reference does */
        for (int i = 0; i < cnt - j - 1; ++i)
            if (trig_atan2(pts[i].y - ctr.y, pts[i].x - ctr.x) > trig_atan2(pts[i + 1].y - ctr.y, pts[i + 1].x - ctr.x)) {
                F2 t = pts[i]; pts[i] = pts[i + 1]; pts[i + 1] = t;
            }
    float area = 0.f;
    for (int k = 0; k < cnt - 1; ++k) {
        F2 u = {pts[k].x - pts[0].x, pts[k].y - pts[0].y}, v = {pts[k + 1].x - pts[0].x, pts[k + 1].y - pts[0].y};
        area += crs2(u, v);
    }
    return (float)(fabs((double)area) / 2.0);
}

float oracle_pcdet_iou(const float *box_a, const float *box_b) {
    const float sa = box_a[3] * box_a[4], sb = box_b[3] * box_b[4];
    const float so = oracle_pcdet_overlap(box_a, box_b);
    return so / fmaxf(sa + sb - so, 1e-8f);
}

/* boxes already sorted by score descending; keep[] receives positions in that order. */
int oracle_pcdet_nms(const float *boxes7, int n, float thr, int *keep) {
    unsigned char *dead = (unsigned char *)calloc((size_t)(n > 0 ? n : 1), 1);
    int nk = 0;
    for (int i = 0; i < n; ++i) {
        if (dead[i]) continue;
        keep[nk++] = i;
        for (int j = i + 1; j < n; ++j)
            if (!dead[j] && oracle_pcdet_iou(boxes7 + 7 * i, boxes7 + 7 * j) > thr) dead[j] = 1;
    }
    free(dead);
    return nk;
}

/* iou3d_nms_kernel.cu:313-325 (iou_normal): heading ignored, axis-aligned (x, y, dx, dy) rectangles in fp32. */
float oracle_pcdet_iou_normal(const float *a, const float *b) {
    const float lo_x = fmaxf(a[0] - a[3] / 2, b[0] - b[3] / 2), hi_x = fminf(a[0] + a[3] / 2, b[0] + b[3] / 2);
    const float lo_y = fmaxf(a[1] - a[4] / 2, b[1] - b[4] / 2), hi_y = fminf(a[1] + a[4] / 2, b[1] + b[4] / 2);
    const float w = fmaxf(hi_x - lo_x, 0.f), h = fmaxf(hi_y - lo_y, 0.f);
    const float inter = w * h;
    return inter / fmaxf(a[3] * a[4] + b[3] * b[4] - inter, 1e-8f);
}

/* nms_normal_gpu (iou3d_nms_utils.py:274-289): the same greedy walk on the axis-aligned IoU. */
int oracle_pcdet_nms_normal(const float *boxes7, int n, float thr, int *keep) {
    unsigned char *dead = (unsigned char *)calloc((size_t)(n > 0 ? n : 1), 1);
    int nk = 0;
    for (int i = 0; i < n; ++i) {
        if (dead[i]) continue;
        keep[nk++] = i;
        for (int j = i + 1; j < n; ++j)
            if (!dead[j] && oracle_pcdet_iou_normal(boxes7 + 7 * i, boxes7 + 7 * j) > thr) dead[j] = 1;
    }
    free(dead);
    return nk;
}

/* Smallest |IoU - thr| over all ordered pairs i < j whose IoU is not exactly 0 (test helper: tells a parity test whether a
 * one-ulp difference between the device's and glibc's sinf / cosf / atan2f could flip a suppression decision). */
float oracle_pcdet_min_margin(const float *boxes7, int n, float thr, int normal) {
    float best = INFINITY;
    for (int i = 0; i < n; ++i)
        for (int j = i + 1; j < n; ++j) {
            const float dx = boxes7[7 * i] - boxes7[7 * j], dy = boxes7[7 * i + 1] - boxes7[7 * j + 1];
            if (dx * dx + dy * dy > 100.f) continue;               /* boxes <= 6 m long cannot touch beyond 10 m */
            const float v = normal ? oracle_pcdet_iou_normal(boxes7 + 7 * i, boxes7 + 7 * j) : oracle_pcdet_iou(boxes7 + 7 * i, boxes7 + 7 * j);
            if (v == 0.f) continue;
            const float m = fabsf(v - thr);
            if (m < best) best = m;
        }
    return best;
}
