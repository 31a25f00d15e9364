// Rotated NMS on device: score ranking, float64 convex-clipping IoU bitmask, device-side greedy reduction,
// in-range gather.  gfx950.
//
// Reference semantics (see include/coalign_amd.h): nms_rotated, opencood/utils/box_utils.py:693-738, which
// copies boxes to the host, builds one Shapely polygon per box and runs a sequential Python loop with two GEOS
// boolean operations per remaining pair.  Here nothing leaves the device:
//   rank_kernel    rank-by-counting on the key (score desc, index desc): O(K^2) compares, LDS tiled, no sort
//                  passes, deterministic; writes the top-`top` order
//   mask_kernel    one 16-wave workgroup per 64x64 tile of the (upper-triangular) pair matrix, row and column
//                  polygons staged in LDS; lanes are the 64 columns, each wave owns 4 rows (row polygon = LDS
//                  broadcast) and one __ballot IS the row's 64-bit suppression word; IoU in float64 by
//                  Sutherland-Hodgman clipping, rounded to float32 before the strict '>' like the reference;
//                  disjoint bounding boxes are rejected without clipping (their IoU is exactly 0)
//   reduce_kernel  single wavefront: lane w owns word w of the "removed" set; the inherently sequential part walks
//                  only the still-alive boxes of each 64-row block (scalar find-first-set on the diagonal word),
//                  the row ORs into the later words are parallel over lanes
//   gather_kernel  kept boxes with all 8 corners inside the range (float64 compare), order preserving
// Compiled with -ffp-contract=off so the float64 arithmetic is bit-identical to the gcc-built CPU oracle.
#include <stdlib.h>

#include "common.h"

namespace {

struct P2 { double x, y; };

__device__ __forceinline__ double signed_area(const P2 *p, int n) {
    double s = 0.0;
    for (int i = 0; i < n; ++i) {
        const P2 a = p[i], b = p[(i + 1 == n) ? 0 : i + 1];
        s += a.x * b.y - b.x * a.y;
    }
    return 0.5 * s;
}

__device__ __forceinline__ int clip_halfplane(const P2 *subj, int n, P2 q0, P2 q1, double sgn, P2 *out) {
    int m = 0;
    const double ex = q1.x - q0.x, ey = q1.y - q0.y;
    for (int i = 0; i < n; ++i) {
        const P2 s = subj[i], e = subj[(i + 1 == n) ? 0 : i + 1];
        const double ds = sgn * (ex * (s.y - q0.y) - ey * (s.x - q0.x));
        const double de = sgn * (ex * (e.y - q0.y) - ey * (e.x - q0.x));
        const bool s_in = ds >= 0.0, e_in = de >= 0.0;
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

// IoU of two convex quads; area_a / area_b are |signed areas|.
__device__ double quad_iou(const P2 *a, double area_a, const P2 *b, double area_b, double sgn_b) {
    P2 buf0[12], buf1[12];
    int n = 4;
    for (int i = 0; i < 4; ++i) buf0[i] = a[i];
    P2 *src = buf0, *dst = buf1;
    for (int k = 0; k < 4 && n > 0; ++k) {
        n = clip_halfplane(src, n, b[k], b[(k + 1) & 3], sgn_b, dst);
        P2 *t = src; src = dst; dst = t;
    }
    const double inter = n < 3 ? 0.0 : fabs(signed_area(src, n));
    const double uni = area_a + area_b - inter;
    return inter / uni;
}

__device__ __forceinline__ int live_k(const int *K_dev, int K) {
    if (!K_dev) return K;
    const int k = *K_dev;
    return k < 0 ? 0 : (k < K ? k : K);
}

// ------------------------------------------------------------------------------------------------ rank
__global__ __launch_bounds__(256) void rank_kernel(const float *__restrict__ scores, const uint8_t *__restrict__ valid,
                                                   int Kcap, const int *K_dev, int top, int *__restrict__ order,
                                                   int *__restrict__ n_sorted) {
    __shared__ float tile[256];
    const int K = live_k(K_dev, Kcap);
    if ((int)(blockIdx.x * 256) >= K && blockIdx.x != 0) return;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const float nanv = __builtin_nanf("");
    float si = nanv;
    if (i < K && (!valid || valid[i])) si = scores[i];
    const bool vi = si == si;
    int rank = 0, nvalid = 0;
    for (int j0 = 0; j0 < K; j0 += 256) {
        const int j = j0 + threadIdx.x;
        float sj = nanv;
        if (j < K && (!valid || valid[j])) sj = scores[j];
        __syncthreads();
        tile[threadIdx.x] = sj;
        __syncthreads();
        const int lim = min(256, K - j0);
        for (int q = 0; q < lim; ++q) {
            const float s = tile[q];
            nvalid += (s == s);
            rank += (s > si) || (s == si && (j0 + q) > i);
        }
    }
    if (vi && rank < top) order[rank] = i;
    if (blockIdx.x == 0 && threadIdx.x == 0) *n_sorted = nvalid < top ? nvalid : top;
}

// ------------------------------------------------------------------------------------------------ mask
struct Poly { P2 v[4]; double area, sgn, xmin, xmax, ymin, ymax; };

__device__ __forceinline__ void load_poly(const float *boxes, int rows, int cols, int idx, Poly &p) {
    const float *b = boxes + (size_t)idx * rows * cols;
    p.xmin = p.ymin = INFINITY; p.xmax = p.ymax = -INFINITY;
    for (int c = 0; c < 4; ++c) {
        p.v[c].x = (double)b[c * cols];
        p.v[c].y = (double)b[c * cols + 1];
        p.xmin = fmin(p.xmin, p.v[c].x); p.xmax = fmax(p.xmax, p.v[c].x);
        p.ymin = fmin(p.ymin, p.v[c].y); p.ymax = fmax(p.ymax, p.v[c].y);
    }
    const double sa = signed_area(p.v, 4);
    p.area = fabs(sa);
    p.sgn = sa >= 0.0 ? 1.0 : -1.0;
}

constexpr int kMaskWaves = 16;   // waves per 64x64 tile: each wave owns 4 rows, lanes are the 64 columns

__global__ __launch_bounds__(kMaskWaves * 64) void mask_kernel(const float *__restrict__ boxes, int rows, int cols,
                                                               const int *__restrict__ order, const int *__restrict__ n_sorted,
                                                               float thr, int nb, unsigned long long *__restrict__ mask) {
    const int cb = blockIdx.x, rb = blockIdx.y;
    if (cb < rb) return;
    const int n = *n_sorted;
    if (rb * 64 >= n || cb * 64 >= n) return;
    __shared__ Poly colp[64], rowp[64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (wv == 0) {
        const int j = cb * 64 + lane;
        if (j < n) load_poly(boxes, rows, cols, order[j], colp[lane]);
    } else if (wv == 1) {
        const int i = rb * 64 + lane;
        if (i < n) load_poly(boxes, rows, cols, order[i], rowp[lane]);
    }
    __syncthreads();
    const int j = cb * 64 + lane;
    Poly c;
    if (j < n) c = colp[lane];
    constexpr int kRowsPerWave = 64 / kMaskWaves;
#pragma unroll 1
    for (int r = 0; r < kRowsPerWave; ++r) {
        const int il = wv * kRowsPerWave + r;
        const int i = rb * 64 + il;
        if (i >= n) break;                                   // wave-uniform
        const Poly &row = rowp[il];                          // LDS broadcast
        bool bit = false;
        if (j < n && j > i) {
            // disjoint bounding boxes => intersection exactly 0 => IoU 0 (or NaN): never '>' a non-negative thr
            const bool apart = thr >= 0.f && (row.xmax < c.xmin || c.xmax < row.xmin || row.ymax < c.ymin || c.ymax < row.ymin);
            if (!apart) bit = (float)quad_iou(row.v, row.area, c.v, c.area, c.sgn) > thr;
        }
        const unsigned long long word = __ballot(bit);
        if (lane == 0) mask[(size_t)i * nb + cb] = word;
    }
}

// ------------------------------------------------------------------------------------------------ reduce
__device__ __forceinline__ unsigned long long readlane64(unsigned long long v, int l) {
    const unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)(v & 0xffffffffull), l);
    const unsigned hi = __builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l);
    return ((unsigned long long)hi << 32) | lo;
}

__global__ __launch_bounds__(64) void reduce_kernel(const unsigned long long *__restrict__ mask, const int *__restrict__ order,
                                                    const int *__restrict__ n_sorted, int nb, int *__restrict__ keep,
                                                    int *__restrict__ keep_count) {
    extern __shared__ unsigned long long rowsbuf[];   // the 64 mask rows of the current block: [64][nb]
    const int lane = threadIdx.x;
    const int n = *n_sorted;
    const int nblk = (n + 63) / 64;
    unsigned long long removed = 0;  // lane w: word w of the suppressed set
    int cnt = 0;
    for (int b = 0; b < nblk; ++b) {
        const int rows = min(64, n - b * 64);
        // one coalesced sweep brings the block's rows (a contiguous rows*nb chunk) into LDS: a single memory round
        // trip per block instead of one dependent global load per kept box
        const unsigned long long *chunk = mask + (size_t)b * 64 * nb;
        for (int e = lane; e < rows * nb; e += 64) {
            const int w = e % nb;
            rowsbuf[e] = (w >= b) ? chunk[e] : 0ull;     // words left of the diagonal are never produced nor needed
        }
        coalign::wave_lds_sync();
        const unsigned long long diag = (lane < rows) ? rowsbuf[lane * nb + b] : 0ull;
        const unsigned long long valid = rows == 64 ? ~0ull : ((1ull << rows) - 1ull);
        // the sequential part: visit only the boxes that are still alive (scalar find-first-set), each kept box
        // ORs its diagonal word into the removed set
        unsigned long long rem = readlane64(removed, b);
        unsigned long long keepbits = 0;
        unsigned long long alive = ~rem & valid;
        while (alive) {
            const int t = __ffsll((long long)alive) - 1;
            keepbits |= 1ull << t;
            rem |= readlane64(diag, t);
            const unsigned long long above = (t == 63) ? 0ull : (~0ull << (t + 1));
            alive = ~rem & valid & above;
        }
        if ((keepbits >> lane) & 1ull) keep[cnt + __popcll(keepbits & ((1ull << lane) - 1ull))] = order[b * 64 + lane];
        cnt += __popcll(keepbits);
        if (lane > b && lane < nb) {                     // lane w ORs word w of every kept row (LDS, pipelined)
            unsigned long long kb = keepbits, acc = 0;
            while (kb) {
                const int t = __ffsll((long long)kb) - 1;
                kb &= kb - 1;
                acc |= rowsbuf[t * nb + lane];
            }
            removed |= acc;
        }
        coalign::wave_lds_sync();
    }
    if (lane == 0) *keep_count = cnt;
}

// ------------------------------------------------------------------------------------------------ round 3: the fast path (top <= 1024)
// The three kernels above cost 24 + 50 + 105-120 us at K ~ 600 (profiles/round2): a handful of workgroups looping over K serially in
// rank_kernel, 512 B of scratch per lane in mask_kernel (the clipper's dynamically indexed polygon buffers) with 9 of 10 lanes idle behind
// the bounding-box test, and a single wavefront walking global memory in reduce_kernel.  Same arithmetic, same results, restructured:
//   rank16_kernel   16 lanes per candidate (DPP row sum of the partial counts): 16 x the workgroups, 1/16 of the serial loop
//   mask2_kernel    per 64 x 64 tile: bounding-box test for all pairs (one ballot per row), the surviving pairs COMPACTED into an LDS list,
//                   then every lane clips one listed pair at a time with its polygon buffers in LDS ([slot][lane]: conflict free,
//                   dynamic indexing costs nothing there); bits land in the tile's words by LDS atomics
//   reduce2_kernel  one 16-wave workgroup: the whole bitmask (<= 1024 x 16 words = 128 KB) and the order list are brought into LDS by all
//                   threads at once, wavefront 0 walks the blocks out of LDS -- the suppression word of a block is an OR over the earlier
//                   kept rows taken lane-parallel (row q * 64 + lane, one LDS read per earlier block) and reduced across the wave -- and
//                   all 1024 threads then run the in-range gather (optional), so decode -> detections is rank, mask, reduce: 3 launches.
__device__ __forceinline__ int row16_sum(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false);      // quad_perm [1, 0, 3, 2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false);      // quad_perm [2, 3, 0, 1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, false);     // row_half_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, false);     // row_mirror
    return v;
}

__global__ __launch_bounds__(256) void rank16_kernel(const float *__restrict__ scores, const uint8_t *__restrict__ valid, int Kcap, const int *K_dev,
                                                     int top, int *__restrict__ order, int *__restrict__ n_sorted) {
    __shared__ float tile[256];
    const int K = live_k(K_dev, Kcap);
    const int i0 = blockIdx.x * 16;
    if (i0 >= K && blockIdx.x != 0) return;
    const int sub = threadIdx.x & 15, i = i0 + (threadIdx.x >> 4);
    const float nanv = __builtin_nanf("");
    float si = nanv;
    if (i < K && (!valid || valid[i])) si = scores[i];
    const bool vi = si == si;
    int rank = 0, nvalid = 0;
    for (int j0 = 0; j0 < K; j0 += 256) {
        const int j = j0 + threadIdx.x;
        float sj = nanv;
        if (j < K && (!valid || valid[j])) sj = scores[j];
        __syncthreads();
        tile[threadIdx.x] = sj;
        __syncthreads();
        const int lim = min(256, K - j0);
        for (int q = sub; q < lim; q += 16) {
            const float s = tile[q];
            nvalid += (s == s);
            rank += (s > si) || (s == si && (j0 + q) > i);     // the key of rank_kernel: score descending, index descending
        }
    }
    rank = row16_sum(rank);
    nvalid = row16_sum(nvalid);
    if (sub == 0 && vi && rank < top) order[rank] = i;
    if (blockIdx.x == 0 && threadIdx.x == 0) *n_sorted = nvalid < top ? nvalid : top;
}

constexpr int kM2 = 256;          // threads of a mask2 workgroup
constexpr int kSlots = 12;        // polygon buffer entries per lane (quad_iou's buf0 / buf1)

struct LaneBuf {                  // one lane's polygon buffer in LDS: entry i at base[i * kM2]
    P2 *base;
    __device__ __forceinline__ P2 &operator[](int i) const { return base[i * kM2]; }
};

// clip_halfplane / signed_area / quad_iou on LaneBuf storage: the same operations in the same order (bit-identical results)
__device__ __forceinline__ int clip_halfplane_l(const LaneBuf subj, int n, P2 q0, P2 q1, double sgn, const LaneBuf out) {
    int m = 0;
    const double ex = q1.x - q0.x, ey = q1.y - q0.y;
    for (int i = 0; i < n; ++i) {
        const P2 s = subj[i], e = subj[(i + 1 == n) ? 0 : i + 1];
        const double ds = sgn * (ex * (s.y - q0.y) - ey * (s.x - q0.x));
        const double de = sgn * (ex * (e.y - q0.y) - ey * (e.x - q0.x));
        const bool s_in = ds >= 0.0, e_in = de >= 0.0;
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

__device__ __forceinline__ double quad_iou_l(const P2 *a, double area_a, const P2 *b, double area_b, double sgn_b, LaneBuf buf0, LaneBuf buf1) {
    int n = 4;
    for (int i = 0; i < 4; ++i) buf0[i] = a[i];
    LaneBuf src = buf0, dst = buf1;
    for (int k = 0; k < 4 && n > 0; ++k) {
        n = clip_halfplane_l(src, n, b[k], b[(k + 1) & 3], sgn_b, dst);
        const LaneBuf t = src; src = dst; dst = t;
    }
    double inter = 0.0;
    if (n >= 3) {
        double s = 0.0;
        for (int i = 0; i < n; ++i) {
            const P2 p = src[i], q = src[(i + 1 == n) ? 0 : i + 1];
            s += p.x * q.y - q.x * p.y;
        }
        inter = fabs(0.5 * s);
    }
    const double uni = area_a + area_b - inter;
    return inter / uni;
}

__global__ __launch_bounds__(kM2) void mask2_kernel(const float *__restrict__ boxes, int rows, int cols, const int *__restrict__ order,
                                                    const int *__restrict__ n_sorted, float thr, int nb, unsigned long long *__restrict__ mask) {
    const int cb = blockIdx.x, rb = blockIdx.y;
    if (cb < rb) return;
    const int n = *n_sorted;
    if (rb * 64 >= n || cb * 64 >= n) return;
    __shared__ Poly colp[64], rowp[64];
    __shared__ unsigned long long cand[64];
    __shared__ unsigned res32[128];
    __shared__ int rowbase[65];
    __shared__ unsigned short list[4096];
    __shared__ __attribute__((aligned(16))) P2 pbuf[2 * kSlots * kM2];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (wv == 0) {
        const int j = cb * 64 + lane;
        if (j < n) load_poly(boxes, rows, cols, order[j], colp[lane]);
    } else if (wv == 1) {
        const int i = rb * 64 + lane;
        if (i < n) load_poly(boxes, rows, cols, order[i], rowp[lane]);
    }
    if (tid < 128) res32[tid] = 0u;
    __syncthreads();
    // ---- every pair of the tile: only the bounding-box test (disjoint boxes => intersection exactly 0 => never '>' a non-negative thr)
    const int j = cb * 64 + lane;
    {
        double cx0 = 0, cx1 = 0, cy0 = 0, cy1 = 0;
        if (j < n) { cx0 = colp[lane].xmin; cx1 = colp[lane].xmax; cy0 = colp[lane].ymin; cy1 = colp[lane].ymax; }
        for (int r = 0; r < 16; ++r) {
            const int il = wv * 16 + r, i = rb * 64 + il;
            bool c = false;
            if (i < n && j < n && j > i) {
                const Poly &row = rowp[il];
                c = !(thr >= 0.f && (row.xmax < cx0 || cx1 < row.xmin || row.ymax < cy0 || cy1 < row.ymin));
            }
            const unsigned long long w = __ballot(c);
            if (lane == 0) cand[il] = w;
        }
    }
    __syncthreads();
    if (wv == 0) {                                           // exclusive scan of the rows' candidate counts
        const int cnt = __popcll(cand[lane]);
        int incl = cnt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int t = __shfl_up(incl, d);
            if (lane >= d) incl += t;
        }
        rowbase[lane] = incl - cnt;
        if (lane == 63) rowbase[64] = incl;
    }
    __syncthreads();
    for (int r = 0; r < 16; ++r) {
        const int il = wv * 16 + r;
        const unsigned long long w = cand[il];
        if ((w >> lane) & 1ull) list[rowbase[il] + __popcll(w & ((1ull << lane) - 1ull))] = (unsigned short)((il << 6) | lane);
    }
    __syncthreads();
    // ---- the listed pairs, one per lane and round: float64 convex clipping, polygon buffers in LDS
    const int total = rowbase[64];
    const LaneBuf b0{pbuf + tid}, b1{pbuf + kSlots * kM2 + tid};
    for (int c = tid; c < total; c += kM2) {
        const int e = list[c], il = e >> 6, jl = e & 63;
        const Poly &row = rowp[il];
        const Poly &col = colp[jl];
        if ((float)quad_iou_l(row.v, row.area, col.v, col.area, col.sgn, b0, b1) > thr) atomicOr(&res32[il * 2 + (jl >> 5)], 1u << (jl & 31));
    }
    __syncthreads();
    if (tid < 64) {
        const int i = rb * 64 + tid;
        if (i < n) mask[(size_t)i * nb + cb] = (unsigned long long)res32[2 * tid] | ((unsigned long long)res32[2 * tid + 1] << 32);
    }
}

__device__ __forceinline__ unsigned long long wave_or64(unsigned long long v) {
    unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        lo |= (unsigned)__shfl_xor((int)lo, d);
        hi |= (unsigned)__shfl_xor((int)hi, d);
    }
    return ((unsigned long long)hi << 32) | lo;
}

constexpr int kR2Rows = 1024, kR2Words = 16;      // reduce2: top <= 1024

__global__ __launch_bounds__(1024) void reduce2_kernel(const unsigned long long *__restrict__ mask, const int *__restrict__ order,
                                                       const int *__restrict__ n_sorted, int nb, int *__restrict__ keep, int *__restrict__ keep_count,
                                                       int do_gather, const float *__restrict__ corners, int box_floats, const float *__restrict__ scores,
                                                       double x0, double y0, double z0, double x1, double y1, double z1,
                                                       float *__restrict__ out_corners, float *__restrict__ out_scores, int *__restrict__ out_count) {
    __shared__ unsigned long long rows[kR2Rows * kR2Words];
    __shared__ unsigned long long kbs[kR2Words];
    __shared__ int ord[kR2Rows], keep_l[kR2Rows];
    __shared__ int wcnt[16];
    __shared__ int s_total;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n = *n_sorted;
    const int nblk = (n + 63) / 64;
    for (int e = tid; e < n * nb; e += 1024) {
        const int r = e / nb, w = e - r * nb;
        rows[e] = (w >= (r >> 6)) ? mask[e] : 0ull;           // words left of the diagonal are never produced nor needed
    }
    for (int e = tid; e < n; e += 1024) ord[e] = order[e];
    __syncthreads();
    if (wv == 0) {
        int cnt = 0;
        for (int b = 0; b < nblk; ++b) {
            const int nrow = min(64, n - b * 64);
            unsigned long long acc = 0ull;
            for (int q = 0; q < b; ++q)
                if ((kbs[q] >> lane) & 1ull) acc |= rows[(q * 64 + lane) * nb + b];
            unsigned long long rem = readlane64(wave_or64(acc), 0);       // scalar from here on: the walk below runs on the scalar unit
            const unsigned long long diag = (lane < nrow) ? rows[(b * 64 + lane) * nb + b] : 0ull;
            const unsigned long long valid = nrow == 64 ? ~0ull : ((1ull << nrow) - 1ull);
            unsigned long long keepbits = 0, alive = ~rem & valid;
            while (alive) {                                   // the inherently sequential part: only the boxes still alive
                const int t = __ffsll((long long)alive) - 1;
                keepbits |= 1ull << t;
                rem |= readlane64(diag, t);
                const unsigned long long above = (t == 63) ? 0ull : (~0ull << (t + 1));
                alive = ~rem & valid & above;
            }
            if ((keepbits >> lane) & 1ull) keep_l[cnt + __popcll(keepbits & ((1ull << lane) - 1ull))] = ord[b * 64 + lane];
            cnt += __popcll(keepbits);
            if (lane == 0) kbs[b] = keepbits;
            coalign::wave_lds_sync();
        }
        if (lane == 0) { s_total = cnt; *keep_count = cnt; }
    }
    __syncthreads();
    const int total = s_total;
    if (tid < total) keep[tid] = keep_l[tid];
    if (!do_gather) return;
    // ---- in-range gather (gather_kernel's rule: all 8 corners inside the range, float64 compare), order preserving
    bool ok = false;
    int src = 0;
    if (tid < total) {
        src = keep_l[tid];
        const float *c = corners + (size_t)src * 24;
        ok = true;
        for (int k = 0; k < 8; ++k) {
            const double X = c[3 * k], Y = c[3 * k + 1], Z = c[3 * k + 2];
            ok = ok && X >= x0 && X <= x1 && Y >= y0 && Y <= y1 && Z >= z0 && Z <= z1;
        }
    }
    const unsigned long long m = __ballot(ok);
    if (lane == 0) wcnt[wv] = __popcll(m);
    __syncthreads();
    int pos = __popcll(m & ((1ull << lane) - 1ull)), tot = 0;
    for (int w = 0; w < 16; ++w) {
        if (w < wv) pos += wcnt[w];
        tot += wcnt[w];
    }
    if (ok) {
        for (int k = 0; k < 24; ++k) out_corners[(size_t)pos * 24 + k] = corners[(size_t)src * 24 + k];
        out_scores[pos] = scores[src];
    }
    if (tid == 0) *out_count = tot;
    (void)box_floats;
}

// ------------------------------------------------------------------------------------------------ gather
__global__ __launch_bounds__(1024) void gather_kernel(const float *__restrict__ corners, const float *__restrict__ scores,
                                                      const int *__restrict__ keep, const int *__restrict__ keep_count,
                                                      int keep_cap, double x0, double y0, double z0, double x1, double y1,
                                                      double z1, float *__restrict__ out_corners,
                                                      float *__restrict__ out_scores, int *__restrict__ out_count) {
    __shared__ int wcnt[16];
    __shared__ int s_base;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int nk = *keep_count;
    nk = nk < keep_cap ? nk : keep_cap;
    if (threadIdx.x == 0) s_base = 0;
    __syncthreads();
    for (int c0 = 0; c0 < nk; c0 += 1024) {
        const int r = c0 + threadIdx.x;
        bool ok = false;
        int src = 0;
        if (r < nk) {
            src = keep[r];
            const float *c = corners + (size_t)src * 24;
            ok = true;
            for (int k = 0; k < 8; ++k) {
                const double X = c[3 * k], Y = c[3 * k + 1], Z = c[3 * k + 2];
                ok = ok && X >= x0 && X <= x1 && Y >= y0 && Y <= y1 && Z >= z0 && Z <= z1;
            }
        }
        const unsigned long long m = __ballot(ok);
        if (lane == 0) wcnt[wv] = __popcll(m);
        __syncthreads();
        int pos = s_base + __popcll(m & ((1ull << lane) - 1ull));
        int tot = 0;
        for (int w = 0; w < 16; ++w) {
            if (w < wv) pos += wcnt[w];
            tot += wcnt[w];
        }
        if (ok) {
            for (int k = 0; k < 24; ++k) out_corners[(size_t)pos * 24 + k] = corners[(size_t)src * 24 + k];
            out_scores[pos] = scores[src];
        }
        __syncthreads();
        if (threadIdx.x == 0) s_base += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) *out_count = s_base;
}

// ------------------------------------------------------------------------------------------------ pcdet fp32 BEV IoU
// RESTATEMENT of OpenPCDet's fp32 overlap arithmetic, not a design of ours: row N asks for the rounding sequence of
// opencood/pcdet_utils/iou3d_nms/src/iou3d_nms_kernel.cu -- cross (:39-42), check_rect_cross + intersection (:44-84: the rectangle pre-test,
// the four orientation products, the ratio form and its line-equation fallback below 1e-8), check_in_box2d (:86-93, margin 1e-2),
// rotate_around_center (:95-101), box_overlap (:104-225: crossings first, then the corners inside the other box, bubble sort by atan2f
// about the centroid, shoelace sum), iou_bev (:227-234), iou_normal (:313-325) -- so every operation below appears in that order; the
// keep lists are compared bit for bit with the oracle's C restatement of the same lines (oracle/rotated_nms.c, tests/test_round2_gpu.py).
// What is ours: the 64 x 64 ballot tiles, the centre-distance early-out and the device-side greedy walk around it (below).
struct F2 { float x, y; };
__device__ __forceinline__ float orient(F2 u, F2 v, F2 origin) { return (u.x - origin.x) * (v.y - origin.y) - (v.x - origin.x) * (u.y - origin.y); }

// edges (e0 -> e1) of one rectangle and (f0 -> f1) of the other: 1 and the crossing point if they properly cross
__device__ int edge_crossing(F2 e1, F2 e0, F2 f1, F2 f0, F2 &hit) {
    if (!(fminf(e0.x, e1.x) <= fmaxf(f0.x, f1.x) && fminf(f0.x, f1.x) <= fmaxf(e0.x, e1.x) &&
          fminf(e0.y, e1.y) <= fmaxf(f0.y, f1.y) && fminf(f0.y, f1.y) <= fmaxf(e0.y, e1.y)))
        return 0;
    const float side_f0 = orient(f0, e1, e0), side_f1 = orient(e1, f1, e0), side_e0 = orient(e0, f1, f0), side_e1 = orient(f1, e1, f0);
    if (!(side_f0 * side_f1 > 0 && side_e0 * side_e1 > 0)) return 0;
    const float span = orient(f1, e1, e0);
    if (fabsf(span - side_f0) > 1e-8f) {
        hit.x = (span * f0.x - side_f0 * f1.x) / (span - side_f0);
        hit.y = (span * f0.y - side_f0 * f1.y) / (span - side_f0);
    } else {                              // nearly parallel: intersect the two line equations  l . (x, y, 1) = 0
        const float la = e0.y - e1.y, lb = e1.x - e0.x, lc = e0.x * e1.y - e1.x * e0.y;
        const float ma = f0.y - f1.y, mb = f1.x - f0.x, mc = f0.x * f1.y - f1.x * f0.y;
        const float det = la * mb - ma * lb;
        hit.x = (lb * mc - mb * lc) / det;
        hit.y = (ma * lc - la * mc) / det;
    }
    return 1;
}

// Round 6: the trigonometry of row N is DEFINED instead of left to the math library: cos / sin / atan2 evaluated in float64 and rounded once to float32 (a
// correctly rounded float function up to ~2^-29 double-rounding cases) -- an admissible cosf / sinf / atan2f like the CUDA run time's or glibc's, and the same bits
// on the device and in the gcc-built oracle (oracle/rotated_nms.c: trig_*), so the IoU MATRIX is bit-comparable, not only the keep lists.
// cos(-a) = cos(a) and sin(-a) = -sin(a) hold exactly for these, so one evaluation per box serves iou3d_nms_kernel.cu's cos(-angle) / sin(-angle) too.
__device__ __forceinline__ float trig_cos(float a) { return (float)cos((double)a); }
__device__ __forceinline__ float trig_sin(float a) { return (float)sin((double)a); }
__device__ __forceinline__ float trig_atan2(float y, float x) { return (float)atan2((double)y, (double)x); }

__device__ int inside_with_margin(const float *box, F2 pt, float cos_a, float sin_a) {
    const float ca = cos_a, sa = -sin_a;                // cos(-angle), sin(-angle)
    const float lx = (pt.x - box[0]) * ca + (pt.y - box[1]) * (-sa);
    const float ly = (pt.x - box[0]) * sa + (pt.y - box[1]) * ca;
    return fabsf(lx) < box[3] / 2 + 1e-2f && fabsf(ly) < box[4] / 2 + 1e-2f;
}

__device__ void box_corners_f32(const float *box, F2 *c, float ca, float sa) {
    const float hx = box[3] / 2, hy = box[4] / 2;
    const float x1 = box[0] - hx, y1 = box[1] - hy, x2 = box[0] + hx, y2 = box[1] + hy;
    const F2 raw[4] = {{x1, y1}, {x2, y1}, {x2, y2}, {x1, y2}};
    for (int k = 0; k < 4; ++k) {
        c[k].x = (raw[k].x - box[0]) * ca + (raw[k].y - box[1]) * (-sa) + box[0];
        c[k].y = (raw[k].x - box[0]) * sa + (raw[k].y - box[1]) * ca + box[1];
    }
    c[4] = c[0];
}

// iou3d_cpu.cpp:128-229 / iou3d_nms_kernel.cu:104-225 (box_overlap), :227-234 (iou_bev)
template <bool AREA>
__device__ float pcdet_iou(const float *A7, const float *B7) {
    F2 A[5], B[5], pts[16], ctr = {0.f, 0.f};
    int cnt = 0;
    const float cos_a = trig_cos(A7[6]), sin_a = trig_sin(A7[6]), cos_b = trig_cos(B7[6]), sin_b = trig_sin(B7[6]);
    box_corners_f32(A7, A, cos_a, sin_a);
    box_corners_f32(B7, B, cos_b, sin_b);
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            if (edge_crossing(A[i + 1], A[i], B[j + 1], B[j], pts[cnt])) { ctr.x += pts[cnt].x; ctr.y += pts[cnt].y; ++cnt; }
    for (int k = 0; k < 4; ++k) {
        if (inside_with_margin(A7, B[k], cos_a, sin_a)) { ctr.x += B[k].x; ctr.y += B[k].y; pts[cnt++] = B[k]; }
        if (inside_with_margin(B7, A[k], cos_b, sin_b)) { ctr.x += A[k].x; ctr.y += A[k].y; pts[cnt++] = A[k]; }
    }
    ctr.x /= cnt; ctr.y /= cnt;
    // the reference's bubble sort by polar angle; the angle of a point is evaluated once (a pure function of the point: the same comparisons, the same order)
    float ang[16];
    for (int i = 0; i < cnt; ++i) ang[i] = trig_atan2(pts[i].y - ctr.y, pts[i].x - ctr.x);
    for (int j = 0; j < cnt - 1; ++j)
        for (int i = 0; i < cnt - j - 1; ++i)
            if (ang[i] > ang[i + 1]) {
                const F2 t = pts[i]; pts[i] = pts[i + 1]; pts[i + 1] = t;
                const float ta = ang[i]; ang[i] = ang[i + 1]; ang[i + 1] = ta;
            }
    float area = 0.f;
    for (int k = 0; k < cnt - 1; ++k) {
        const float ux = pts[k].x - pts[0].x, uy = pts[k].y - pts[0].y;
        const float vx = pts[k + 1].x - pts[0].x, vy = pts[k + 1].y - pts[0].y;
        area += ux * vy - uy * vx;
    }
    const float so = fabsf(area) / 2.0f;
    if (AREA) return so;                   // boxes_overlap_bev_gpu: the overlap area itself
    const float sa = A7[3] * A7[4], sb = B7[3] * B7[4];
    return so / fmaxf(sa + sb - so, 1e-8f);
}

__device__ __forceinline__ float pcdet_iou_normal(const float *a, const float *b) {      // iou3d_nms_kernel.cu:313-325: heading ignored
    const float lo_x = fmaxf(a[0] - a[3] / 2, b[0] - b[3] / 2), hi_x = fminf(a[0] + a[3] / 2, b[0] + b[3] / 2);
    const float lo_y = fmaxf(a[1] - a[4] / 2, b[1] - b[4] / 2), hi_y = fminf(a[1] + a[4] / 2, b[1] + b[4] / 2);
    const float w = fmaxf(hi_x - lo_x, 0.f), h = fmaxf(hi_y - lo_y, 0.f);
    const float inter = w * h;
    return inter / fmaxf(a[3] * a[4] + b[3] * b[4] - inter, 1e-8f);
}

// OpenPCDet-semantics suppression bitmask (iou3d_nms_kernel.cu:267-311 nms_kernel, :328-372 nms_normal_kernel): boxes are already
// sorted by score; word (i, cb) bit l = "box cb*64+l is suppressed by box i", only for l > i.  Same tiling as mask_kernel above:
// one 16-wave workgroup per 64x64 tile of the upper triangle, lanes = columns, 4 rows per wave, one __ballot = the row's word.
// Row box = first argument of the IoU, column box = second (the reference's order; fp32 box_overlap is not symmetric in rounding).
template <bool NORMAL>
__global__ __launch_bounds__(kMaskWaves * 64) void pcdet_mask_kernel(const float *__restrict__ boxes, int n, float thr, int nb,
                                                                     unsigned long long *__restrict__ mask) {
    const int cb = blockIdx.x, rb = blockIdx.y;
    if (cb < rb) return;
    __shared__ float colb[64 * 7], rowb[64 * 7];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int e = threadIdx.x; e < 64 * 7; e += kMaskWaves * 64) {
        const int jc = cb * 64 * 7 + e, jr = rb * 64 * 7 + e;
        colb[e] = jc < n * 7 ? boxes[jc] : 0.f;
        rowb[e] = jr < n * 7 ? boxes[jr] : 0.f;
    }
    __syncthreads();
    const int j = cb * 64 + lane;
    float c[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) c[k] = colb[lane * 7 + k];
    constexpr int kRowsPerWave = 64 / kMaskWaves;
#pragma unroll 1
    for (int r = 0; r < kRowsPerWave; ++r) {
        const int il = wv * kRowsPerWave + r;
        const int i = rb * 64 + il;
        if (i >= n) break;                                   // wave-uniform
        float a[7];
#pragma unroll
        for (int k = 0; k < 7; ++k) a[k] = rowb[il * 7 + k];   // LDS broadcast
        bool bit = false;
        if (j < n && j > i) {
            if (NORMAL) bit = pcdet_iou_normal(a, c) > thr;
            else {
                // centres further apart than the two half diagonals: no edge crossing, no corner inside (margin 1e-2 included)
                // => overlap exactly 0 => IoU 0, never '>' a non-negative threshold; skips the trigonometry for most pairs
                const float dx = a[0] - c[0], dy = a[1] - c[1];
                const float ra = 0.5f * sqrtf(a[3] * a[3] + a[4] * a[4]) + 0.02f, rc = 0.5f * sqrtf(c[3] * c[3] + c[4] * c[4]) + 0.02f;
                const bool apart = thr >= 0.f && dx * dx + dy * dy > (ra + rc) * (ra + rc) * 1.0001f;
                if (!apart) bit = pcdet_iou<false>(a, c) > thr;
            }
        }
        const unsigned long long word = __ballot(bit);
        if (lane == 0) mask[(size_t)i * nb + cb] = word;
    }
}

// Greedy walk over the bitmask for more than 4096 boxes (nb > 64 words per row): lane w owns words w, w + 64, ... of the removed
// set; rows are read straight from global memory (the LDS-staged single-word-per-lane reduce_kernel above serves nb <= 64).
constexpr int kBigWords = 4;                                  // <= 16384 boxes
__global__ __launch_bounds__(64) void reduce_big_kernel(const unsigned long long *__restrict__ mask, int n, int nb, int *__restrict__ keep,
                                                        int *__restrict__ keep_count) {
    const int lane = threadIdx.x;
    unsigned long long removed[kBigWords] = {0ull, 0ull, 0ull, 0ull};
    int cnt = 0;
    const int nblk = (n + 63) / 64;
    for (int b = 0; b < nblk; ++b) {
        const int rows = min(64, n - b * 64);
        const unsigned long long valid = rows == 64 ? ~0ull : ((1ull << rows) - 1ull);
        const unsigned long long diag = (lane < rows) ? mask[(size_t)(b * 64 + lane) * nb + b] : 0ull;
        unsigned long long own = 0ull;                        // word b of the removed set lives in lane b % 64, slot b / 64
#pragma unroll
        for (int q = 0; q < kBigWords; ++q)
            if (q == b / 64) own = removed[q];
        unsigned long long rem = readlane64(own, b % 64);
        unsigned long long keepbits = 0, alive = ~rem & valid;
        while (alive) {
            const int t = __ffsll((long long)alive) - 1;
            keepbits |= 1ull << t;
            rem |= readlane64(diag, t);
            const unsigned long long above = (t == 63) ? 0ull : (~0ull << (t + 1));
            alive = ~rem & valid & above;
        }
        if ((keepbits >> lane) & 1ull) keep[cnt + __popcll(keepbits & ((1ull << lane) - 1ull))] = b * 64 + lane;
        cnt += __popcll(keepbits);
#pragma unroll
        for (int q = 0; q < kBigWords; ++q) {
            const int w = lane + 64 * q;
            if (w > b && w < nb) {
                unsigned long long kb = keepbits, acc = 0;
                while (kb) {
                    const int t = __ffsll((long long)kb) - 1;
                    kb &= kb - 1;
                    acc |= mask[(size_t)(b * 64 + t) * nb + w];
                }
                removed[q] |= acc;
            }
        }
    }
    if (lane == 0) *keep_count = cnt;
}

__global__ void iota_kernel(int *__restrict__ order, int *__restrict__ n_sorted, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) order[i] = i;
    if (i == 0) *n_sorted = n;
}

template <bool AREA>
__global__ __launch_bounds__(256) void iou_bev_kernel(const float *__restrict__ a, int Na, const float *__restrict__ b, int Nb,
                                                      float *__restrict__ iou) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)Na * Nb) return;
    const int i = (int)(idx / Nb), j = (int)(idx % Nb);
    iou[idx] = pcdet_iou<AREA>(a + (size_t)i * 7, b + (size_t)j * 7);
}

// ------------------------------------------------------------------------------------------------ float64 IoU matrix
// (evaluation: TP / FP matching, opencood/utils/eval_utils.py:45-96 builds the same numbers with one Shapely call per pair)
__global__ __launch_bounds__(256) void iou_matrix_kernel(const float *__restrict__ a, int rows_a, int cols_a, int Na,
                                                         const float *__restrict__ b, int rows_b, int cols_b, int Nb,
                                                         float *__restrict__ iou) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)Na * Nb) return;
    const int i = (int)(idx / Nb), j = (int)(idx % Nb);
    Poly pa, pb;
    load_poly(a, rows_a, cols_a, i, pa);
    load_poly(b, rows_b, cols_b, j, pb);
    float r = 0.f;
    const bool apart = pa.xmax < pb.xmin || pb.xmax < pa.xmin || pa.ymax < pb.ymin || pb.ymax < pa.ymin;
    if (!apart || pa.area + pb.area == 0.0) r = (float)quad_iou(pa.v, pa.area, pb.v, pb.area, pb.sgn);
    iou[idx] = r;
}

struct NmsWs {
    int *order, *n_sorted;
    unsigned long long *mask;
};

NmsWs carve(void *ws, int top) {
    NmsWs w;
    char *p = (char *)ws;
    const int nb = (top + 63) / 64;
    w.mask = (unsigned long long *)p; p += coalign::align_up((size_t)top * nb * 8, 256);
    w.order = (int *)p;               p += coalign::align_up((size_t)top * 4, 256);
    w.n_sorted = (int *)p;
    return w;
}

}  // namespace

extern "C" {

size_t coalign_nms_rotated_workspace_bytes(int K, int top) {
    (void)K;
    if (top <= 0) return 0;
    const int nb = (top + 63) / 64;
    return coalign::align_up((size_t)top * nb * 8, 256) + coalign::align_up((size_t)top * 4, 256) + 256;
}

static int nms_rotated_impl(const float *boxes, int rows, int cols, const float *scores, const uint8_t *valid, int K, const int32_t *K_dev,
                            float iou_thr, int top, int32_t *keep, int32_t *keep_count, void *workspace, size_t workspace_bytes,
                            const double *range6_host, float *out_corners, float *out_scores, int32_t *out_count, void *stream_) {
    using namespace coalign;
    hipStream_t stream = (hipStream_t)stream_;
    const bool gather = range6_host != nullptr;
    if (K < 0 || rows < 4 || cols < 2 || top <= 0) return COALIGN_ERR_BAD_SHAPE;
    if (top > 4096) return COALIGN_ERR_UNSUPPORTED;
    if (!keep_count || !workspace || !keep) return COALIGN_ERR_NULL_POINTER;
    if (gather && (rows != 8 || cols != 3)) return COALIGN_ERR_BAD_SHAPE;                 // the gather copies [8, 3] corner blocks
    if (gather && top > kR2Rows) return COALIGN_ERR_UNSUPPORTED;
    if (gather && (!out_corners || !out_scores || !out_count)) return COALIGN_ERR_NULL_POINTER;
    if (workspace_bytes < coalign_nms_rotated_workspace_bytes(K, top)) return COALIGN_ERR_WORKSPACE;
    if (K == 0) {
        int rc = fill_words(keep_count, 1, 0u, stream);
        if (!rc && gather) rc = fill_words(out_count, 1, 0u, stream);
        return rc;
    }
    if (!boxes || !scores) return COALIGN_ERR_NULL_POINTER;
    NmsWs w = carve(workspace, top);
    const int nb = (top + 63) / 64;
    static const bool legacy = [] { const char *e = getenv("COALIGN_NMS_LEGACY"); return e && e[0] == '1'; }();      // measurement switch: round-2 kernels
    const bool fast = top <= kR2Rows && !legacy;
    int rc;
    if (fast) hipLaunchKernelGGL(rank16_kernel, dim3((K + 15) / 16), dim3(256), 0, stream, scores, valid, K, K_dev, top, w.order, w.n_sorted);
    else hipLaunchKernelGGL(rank_kernel, dim3((K + 255) / 256), dim3(256), 0, stream, scores, valid, K, K_dev, top, w.order, w.n_sorted);
    if ((rc = check_launch())) return rc;
    if (fast) hipLaunchKernelGGL(mask2_kernel, dim3(nb, nb), dim3(kM2), 0, stream, boxes, rows, cols, w.order, w.n_sorted, iou_thr, nb, w.mask);
    else hipLaunchKernelGGL(mask_kernel, dim3(nb, nb), dim3(kMaskWaves * 64), 0, stream, boxes, rows, cols, w.order, w.n_sorted, iou_thr, nb, w.mask);
    if ((rc = check_launch())) return rc;
    const double zero6[6] = {0, 0, 0, 0, 0, 0};
    const double *r = gather ? range6_host : zero6;
    if (fast) {
        hipLaunchKernelGGL(reduce2_kernel, dim3(1), dim3(1024), 0, stream, w.mask, w.order, w.n_sorted, nb, keep, keep_count, gather ? 1 : 0, boxes,
                           rows * cols, scores, r[0], r[1], r[2], r[3], r[4], r[5], out_corners, out_scores, out_count);
        return check_launch();
    }
    hipLaunchKernelGGL(reduce_kernel, dim3(1), dim3(64), (size_t)64 * nb * sizeof(unsigned long long), stream, w.mask, w.order,
                       w.n_sorted, nb, keep, keep_count);
    if ((rc = check_launch())) return rc;
    if (gather) {
        hipLaunchKernelGGL(gather_kernel, dim3(1), dim3(1024), 0, stream, boxes, scores, keep, keep_count, top, r[0], r[1], r[2], r[3], r[4], r[5],
                           out_corners, out_scores, out_count);
        return check_launch();
    }
    return COALIGN_OK;
}

int coalign_nms_rotated(const float *boxes, int rows, int cols, const float *scores, const uint8_t *valid, int K,
                        const int32_t *K_dev, float iou_thr, int top, int32_t *keep, int32_t *keep_count, void *workspace,
                        size_t workspace_bytes, void *stream) {
    return nms_rotated_impl(boxes, rows, cols, scores, valid, K, K_dev, iou_thr, top, keep, keep_count, workspace, workspace_bytes, nullptr, nullptr,
                            nullptr, nullptr, stream);
}

int coalign_nms_rotated_gather(const float *corners, const float *scores, const uint8_t *valid, int K, const int32_t *K_dev, float iou_thr, int top,
                               int32_t *keep, int32_t *keep_count, const double *range6_host, float *out_corners, float *out_scores,
                               int32_t *out_count, void *workspace, size_t workspace_bytes, void *stream) {
    if (!range6_host) return COALIGN_ERR_NULL_POINTER;
    return nms_rotated_impl(corners, 8, 3, scores, valid, K, K_dev, iou_thr, top, keep, keep_count, workspace, workspace_bytes, range6_host, out_corners,
                            out_scores, out_count, stream);
}

int coalign_gather_in_range(const float *corners, const float *scores, const int32_t *keep, const int32_t *keep_count,
                            int keep_cap, const double *range6_host, float *out_corners, float *out_scores,
                            int32_t *out_count, void *stream_) {
    using namespace coalign;
    hipStream_t stream = (hipStream_t)stream_;
    if (keep_cap < 0) return COALIGN_ERR_BAD_SHAPE;
    if (keep_cap > 4096) return COALIGN_ERR_UNSUPPORTED;
    if (!keep_count || !range6_host || !out_count) return COALIGN_ERR_NULL_POINTER;
    if (keep_cap > 0 && (!corners || !scores || !keep || !out_corners || !out_scores)) return COALIGN_ERR_NULL_POINTER;
    const double *r = range6_host;
    hipLaunchKernelGGL(gather_kernel, dim3(1), dim3(1024), 0, stream, corners, scores, keep, keep_count, keep_cap, r[0], r[1],
                       r[2], r[3], r[4], r[5], out_corners, out_scores, out_count);
    return check_launch();
}

int coalign_iou_rotated_matrix(const float *boxes_a, int rows_a, int cols_a, int Na, const float *boxes_b, int rows_b, int cols_b,
                               int Nb, float *iou, void *stream_) {
    using namespace coalign;
    hipStream_t stream = (hipStream_t)stream_;
    if (Na < 0 || Nb < 0 || rows_a < 4 || cols_a < 2 || rows_b < 4 || cols_b < 2) return COALIGN_ERR_BAD_SHAPE;
    if (Na == 0 || Nb == 0) return COALIGN_OK;
    if (!boxes_a || !boxes_b || !iou) return COALIGN_ERR_NULL_POINTER;
    const long total = (long)Na * Nb;
    hipLaunchKernelGGL(iou_matrix_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, boxes_a, rows_a, cols_a, Na,
                       boxes_b, rows_b, cols_b, Nb, iou);
    return check_launch();
}

static int launch_iou_bev(bool area, const float *boxes_a, int Na, const float *boxes_b, int Nb, float *out, void *stream_) {
    using namespace coalign;
    hipStream_t stream = (hipStream_t)stream_;
    if (Na < 0 || Nb < 0) return COALIGN_ERR_BAD_SHAPE;
    if (Na == 0 || Nb == 0) return COALIGN_OK;
    if (!boxes_a || !boxes_b || !out) return COALIGN_ERR_NULL_POINTER;
    const long total = (long)Na * Nb;
    const dim3 grid((unsigned)((total + 255) / 256));
    if (area) hipLaunchKernelGGL(iou_bev_kernel<true>, grid, dim3(256), 0, stream, boxes_a, Na, boxes_b, Nb, out);
    else hipLaunchKernelGGL(iou_bev_kernel<false>, grid, dim3(256), 0, stream, boxes_a, Na, boxes_b, Nb, out);
    return check_launch();
}

int coalign_boxes_iou_bev(const float *boxes_a, int Na, const float *boxes_b, int Nb, float *iou, void *stream) {
    return launch_iou_bev(false, boxes_a, Na, boxes_b, Nb, iou, stream);
}

int coalign_boxes_overlap_bev(const float *boxes_a, int Na, const float *boxes_b, int Nb, float *overlap, void *stream) {
    return launch_iou_bev(true, boxes_a, Na, boxes_b, Nb, overlap, stream);
}

size_t coalign_pcdet_nms_workspace_bytes(int n) {
    if (n <= 0) return 0;
    const size_t nb = ((size_t)n + 63) / 64;
    return coalign::align_up((size_t)n * nb * 8, 256) + coalign::align_up((size_t)n * 4, 256) + 256;
}

int coalign_pcdet_nms(const float *boxes_sorted, int n, float thresh, int normal, int32_t *keep, int32_t *keep_count, void *workspace,
                      size_t workspace_bytes, void *stream_) {
    using namespace coalign;
    hipStream_t stream = (hipStream_t)stream_;
    if (n < 0) return COALIGN_ERR_BAD_SHAPE;
    if (n > 64 * 64 * kBigWords) return COALIGN_ERR_UNSUPPORTED;
    if (!keep_count) return COALIGN_ERR_NULL_POINTER;
    if (n == 0) return fill_words(keep_count, 1, 0u, stream);
    if (!boxes_sorted || !keep || !workspace) return COALIGN_ERR_NULL_POINTER;
    if (workspace_bytes < coalign_pcdet_nms_workspace_bytes(n)) return COALIGN_ERR_WORKSPACE;
    const int nb = (n + 63) / 64;
    char *p = (char *)workspace;
    unsigned long long *mask = (unsigned long long *)p; p += align_up((size_t)n * nb * 8, 256);
    int *order = (int *)p;                              p += align_up((size_t)n * 4, 256);
    int *n_sorted = (int *)p;
    if (normal) hipLaunchKernelGGL(pcdet_mask_kernel<true>, dim3(nb, nb), dim3(kMaskWaves * 64), 0, stream, boxes_sorted, n, thresh, nb, mask);
    else hipLaunchKernelGGL(pcdet_mask_kernel<false>, dim3(nb, nb), dim3(kMaskWaves * 64), 0, stream, boxes_sorted, n, thresh, nb, mask);
    int rc = check_launch();
    if (rc) return rc;
    if (nb <= 64) {                                        // the shared LDS-staged greedy reduction; order = identity (already sorted)
        hipLaunchKernelGGL(iota_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, order, n_sorted, n);
        if ((rc = check_launch())) return rc;
        hipLaunchKernelGGL(reduce_kernel, dim3(1), dim3(64), (size_t)64 * nb * sizeof(unsigned long long), stream, mask, order, n_sorted, nb,
                           keep, keep_count);
    } else {
        hipLaunchKernelGGL(reduce_big_kernel, dim3(1), dim3(64), 0, stream, mask, n, nb, keep, keep_count);
    }
    return check_launch();
}

}  // extern "C"
