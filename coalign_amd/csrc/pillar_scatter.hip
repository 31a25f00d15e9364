// Pillar feature encoder (PFN) + scatter to the dense BEV canvas, gfx950.
//
// Reference semantics: opencood/models/sub_modules/pillar_vfe.py:31-53,105-155 and
// opencood/models/sub_modules/point_pillar_scatter.py:15-72 (see include/coalign_amd.h).
//
// Three launches on the caller's stream:
//   memset(cell -> pillar map, -1)
//   pfn_kernel     one wavefront per pillar.  Phase A: lane = point  (coalesced 16 B/lane read of the pillar,
//                  xor-butterfly wave reduction for the per-pillar mean, 10-d augmentation computed once per
//                  point, staged in a wave-private LDS slab).  Phase B: lane = output channel (weights, folded
//                  BN scale/shift in registers; LDS broadcast reads; running max in a register).  Padded rows
//                  are never multiplied out: they all contribute relu(BN(0)), folded in as the max's seed.
//                  Writes pillar_features [M, C] (256 B coalesced per pillar) and atomicMax(cell map, row).
//   canvas_kernel  streaming writer of the NCHW canvas: every thread owns 4 consecutive cells and a block of
//                  channels, reads the cell map once (16 B), gathers the (rare, 5-6 % occupancy) pillar rows
//                  and issues 16 B non-temporal stores, so the dominant traffic -- the dense canvas the
//                  convolution backbone consumes -- is written exactly once, fully coalesced.
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int kWavesPerBlock = 4;
constexpr int kFeatStride = 12;  // floats per staged point (<= 11 features), 48 B keeps 16 B alignment

struct PfnArgs {
    const float4 *pts;
    const int *npts;
    const int4 *coords;
    int M, P;
    const float *weight, *bias, *bn_w, *bn_b, *bn_m, *bn_v;
    float eps;
    int C, Cin, use_abs, with_dist;
    float vx, vy, vz, xo, yo, zo;
    int n_agents, ny, nx;
    float *feats;
    int *cellmap;
    const int *M_dev;     // optional: the pillar count lives on the device (the voxeliser's voxel_counts word); M is then the capacity
    int unique;           // the caller guarantees one pillar per cell (voxeliser output): no cell-map lookup
    int debug;            // measurement switches (COALIGN_PILLAR_DEBUG): 1 no feature-row stores, 2 no canvas stores, 4 no matrix / epilogue work
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// Per-lane channel parameters: one row of the linear layer and the folded BatchNorm affine.
struct ChanParams {
    float w[kFeatStride];
    float alpha, shift;
};

__device__ __forceinline__ ChanParams load_chan(const PfnArgs &a, int c) {
    ChanParams cp;
    const bool c_ok = c < a.C;
#pragma unroll
    for (int k = 0; k < kFeatStride; ++k) cp.w[k] = (c_ok && k < a.Cin) ? a.weight[(size_t)c * a.Cin + k] : 0.f;
    cp.alpha = 1.f; cp.shift = 0.f;
    if (c_ok) {
        if (a.bn_w) {
            const float inv_std = 1.0f / sqrtf(a.bn_v[c] + a.eps);
            cp.alpha = a.bn_w[c] * inv_std;
            cp.shift = a.bn_b[c] - a.bn_m[c] * cp.alpha;
        } else if (a.bias) {
            cp.shift = a.bias[c];
        }
    }
    return cp;
}

// 10/11-d augmentation of one point (pillar_vfe.py:118-141), written to the wave's LDS slab row `row`.
// Compile-time feature layout (all indices static, so the staging array lives in registers).
template <bool ABS, bool DIST>
__device__ __forceinline__ void stage_point_t(float *slab, int row, float4 q, float mx, float my, float mz, float ctr_x,
                                              float ctr_y, float ctr_z) {
    float f[kFeatStride];
#pragma unroll
    for (int k = 0; k < kFeatStride; ++k) f[k] = 0.f;
    constexpr int B = ABS ? 4 : 1;
    if constexpr (ABS) { f[0] = q.x; f[1] = q.y; f[2] = q.z; f[3] = q.w; }
    else { f[0] = q.w; }
    f[B] = q.x - mx; f[B + 1] = q.y - my; f[B + 2] = q.z - mz;
    f[B + 3] = q.x - ctr_x; f[B + 4] = q.y - ctr_y; f[B + 5] = q.z - ctr_z;
    if constexpr (DIST) f[B + 6] = sqrtf(q.x * q.x + q.y * q.y + q.z * q.z);
    float4 *dst = reinterpret_cast<float4 *>(slab + row * kFeatStride);
    dst[0] = make_float4(f[0], f[1], f[2], f[3]);
    dst[1] = make_float4(f[4], f[5], f[6], f[7]);
    dst[2] = make_float4(f[8], f[9], f[10], f[11]);
}

__device__ __forceinline__ void stage_point(const PfnArgs &a, float *slab, int row, float4 q, float mx, float my, float mz,
                                            float ctr_x, float ctr_y, float ctr_z) {
    if (a.use_abs) {
        if (a.with_dist) stage_point_t<true, true>(slab, row, q, mx, my, mz, ctr_x, ctr_y, ctr_z);
        else stage_point_t<true, false>(slab, row, q, mx, my, mz, ctr_x, ctr_y, ctr_z);
    } else {
        if (a.with_dist) stage_point_t<false, true>(slab, row, q, mx, my, mz, ctr_x, ctr_y, ctr_z);
        else stage_point_t<false, false>(slab, row, q, mx, my, mz, ctr_x, ctr_y, ctr_z);
    }
}

__device__ __forceinline__ float point_response(const ChanParams &cp, const float *slab, int j) {
    const float4 *src = reinterpret_cast<const float4 *>(slab + j * kFeatStride);
    const float4 u = src[0], v = src[1];
    const float2 t = *reinterpret_cast<const float2 *>(slab + j * kFeatStride + 8);
    const float2 t2 = *reinterpret_cast<const float2 *>(slab + j * kFeatStride + 10);   // zeros (times zero weights) unless Cin > 10
    float x = cp.w[0] * u.x;
    x = fmaf(cp.w[1], u.y, x); x = fmaf(cp.w[2], u.z, x); x = fmaf(cp.w[3], u.w, x);
    x = fmaf(cp.w[4], v.x, x); x = fmaf(cp.w[5], v.y, x); x = fmaf(cp.w[6], v.z, x); x = fmaf(cp.w[7], v.w, x);
    x = fmaf(cp.w[8], t.x, x); x = fmaf(cp.w[9], t.y, x); x = fmaf(cp.w[10], t2.x, x); x = fmaf(cp.w[11], t2.y, x);
    return fmaf(x, cp.alpha, cp.shift);
}

// ---- packed variant used by the two-pillars-per-wavefront encoders: the staged points of a wavefront are laid out in PAIRS,
// slab[pair][k][2] (pair = row / 2), so that one ds_read_b128 yields features k, k + 1 of two neighbouring points as two register
// pairs (the linear layer ran on v_pk_fma_f32 in round 2; scalar FMA chains since round 3).  Same operations per point in the same order as
// point_response(): bit-identical results.

template <bool ABS, bool DIST>
__device__ __forceinline__ void stage_point_pk_t(float *slab, int row, float4 q, float mx, float my, float mz, float ctr_x, float ctr_y,
                                                 float ctr_z) {
    float f[kFeatStride];
#pragma unroll
    for (int k = 0; k < kFeatStride; ++k) f[k] = 0.f;
    constexpr int B = ABS ? 4 : 1;
    if constexpr (ABS) { f[0] = q.x; f[1] = q.y; f[2] = q.z; f[3] = q.w; }
    else { f[0] = q.w; }
    f[B] = q.x - mx; f[B + 1] = q.y - my; f[B + 2] = q.z - mz;
    f[B + 3] = q.x - ctr_x; f[B + 4] = q.y - ctr_y; f[B + 5] = q.z - ctr_z;
    if constexpr (DIST) f[B + 6] = sqrtf(q.x * q.x + q.y * q.y + q.z * q.z);
    float *dst = slab + (row >> 1) * (2 * kFeatStride) + (row & 1);
#pragma unroll
    for (int k = 0; k < kFeatStride; ++k) dst[2 * k] = f[k];
}

__device__ __forceinline__ void stage_point_pk(const PfnArgs &a, float *slab, int row, float4 q, float mx, float my, float mz, float ctr_x,
                                               float ctr_y, float ctr_z) {
    if (a.use_abs) {
        if (a.with_dist) stage_point_pk_t<true, true>(slab, row, q, mx, my, mz, ctr_x, ctr_y, ctr_z);
        else stage_point_pk_t<true, false>(slab, row, q, mx, my, mz, ctr_x, ctr_y, ctr_z);
    } else {
        if (a.with_dist) stage_point_pk_t<false, true>(slab, row, q, mx, my, mz, ctr_x, ctr_y, ctr_z);
        else stage_point_pk_t<false, false>(slab, row, q, mx, my, mz, ctr_x, ctr_y, ctr_z);
    }
}

// responses of the two points of pair `pr` (rows 2 pr, 2 pr + 1): .x / .y.  Two scalar FMA chains: until round 3 this was v_pk_fma_f32 on
// float2 operands (measured no faster: the kernel is not issue bound) -- packed fp32 instructions are gone from the library
// (profiles/round3/README.md); same operations per point in the same order, bit-identical results.
struct f2 { float x, y; };
__device__ __forceinline__ f2 pair_response(const ChanParams &cp, const float *slab, int pr) {
    const float4 *src = reinterpret_cast<const float4 *>(slab + pr * (2 * kFeatStride));
    const float4 v[6] = {src[0], src[1], src[2], src[3], src[4], src[5]};
    float x = cp.w[0] * v[0].x, y = cp.w[0] * v[0].y;
    x = fmaf(cp.w[1], v[0].z, x); y = fmaf(cp.w[1], v[0].w, y);
#pragma unroll
    for (int k = 1; k < 6; ++k) {
        x = fmaf(cp.w[2 * k], v[k].x, x); y = fmaf(cp.w[2 * k], v[k].y, y);
        x = fmaf(cp.w[2 * k + 1], v[k].z, x); y = fmaf(cp.w[2 * k + 1], v[k].w, y);
    }
    return f2{fmaf(x, cp.alpha, cp.shift), fmaf(y, cp.alpha, cp.shift)};
}

// relu(max over the np_eff staged points starting at (even) row0) -- rows_max() on the paired layout
__device__ __forceinline__ float rows_max_pk(const ChanParams &cp, const float *slab, int row0, int np_eff, int P) {
    float best = (np_eff < P) ? cp.shift : -INFINITY;
    const int pr0 = row0 >> 1;
    int j = 0;
    for (; j + 4 <= np_eff; j += 4) {
        const f2 y01 = pair_response(cp, slab, pr0 + (j >> 1)), y23 = pair_response(cp, slab, pr0 + (j >> 1) + 1);
        best = fmaxf(fmaxf(best, fmaxf(y01.x, y01.y)), fmaxf(y23.x, y23.y));
    }
    if (j + 2 <= np_eff) {
        const f2 y = pair_response(cp, slab, pr0 + (j >> 1));
        best = fmaxf(best, fmaxf(y.x, y.y));
        j += 2;
    }
    if (j < np_eff) best = fmaxf(best, pair_response(cp, slab, pr0 + (j >> 1)).x);     // odd tail: the pair's second slot is stale
    return fmaxf(best, 0.f);
}

// Fast path, P <= 64 and C <= 64: the pillar is read exactly once (lane = point), the NEXT pillar's loads are issued
// before the current one is processed (one HBM round trip per pillar, hidden behind compute), phase B walks the
// staged points four at a time with independent FMA chains.
__global__ __launch_bounds__(kWavesPerBlock * 64) void pfn_kernel_p64(PfnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    const int wib = threadIdx.x >> 6;
    float *slab = smem + (size_t)wib * 64 * kFeatStride;
    const int gwave = blockIdx.x * kWavesPerBlock + wib;
    const int nwave = gridDim.x * kWavesPerBlock;
    const int ncell = a.ny * a.nx;
    const ChanParams cp = load_chan(a, lane);
    const bool c_ok = lane < a.C;

    int m = gwave;
    float4 q_nx = make_float4(0.f, 0.f, 0.f, 0.f); int np_nx = 0; int4 cd_nx = make_int4(0, 0, 0, 0);
    if (m < a.M) {
        if (lane < a.P) q_nx = a.pts[(size_t)m * a.P + lane];
        np_nx = a.npts[m];
        cd_nx = a.coords[m];
    }
    for (; m < a.M; m += nwave) {
        const float4 q = q_nx;
        const int np_raw = np_nx;
        const int4 cd = cd_nx;
        const int mn = m + nwave;
        if (mn < a.M) {  // prefetch
            q_nx = make_float4(0.f, 0.f, 0.f, 0.f);
            if (lane < a.P) q_nx = a.pts[(size_t)mn * a.P + lane];
            np_nx = a.npts[mn];
            cd_nx = a.coords[mn];
        }
        const int np_eff = min(max(np_raw, 0), a.P);
        const float npf = (float)np_raw;
        const float mx = wave_sum(q.x) / npf, my = wave_sum(q.y) / npf, mz = wave_sum(q.z) / npf;
        const float ctr_x = (float)cd.w * a.vx + a.xo;
        const float ctr_y = (float)cd.z * a.vy + a.yo;
        const float ctr_z = (float)cd.y * a.vz + a.zo;
        if (lane < np_eff) stage_point(a, slab, lane, q, mx, my, mz, ctr_x, ctr_y, ctr_z);
        coalign::wave_lds_sync();
        float best = (np_eff < a.P) ? cp.shift : -INFINITY;
        int j = 0;
        for (; j + 4 <= np_eff; j += 4) {
            const float y0 = point_response(cp, slab, j), y1 = point_response(cp, slab, j + 1);
            const float y2 = point_response(cp, slab, j + 2), y3 = point_response(cp, slab, j + 3);
            best = fmaxf(fmaxf(best, fmaxf(y0, y1)), fmaxf(y2, y3));
        }
        for (; j < np_eff; ++j) best = fmaxf(best, point_response(cp, slab, j));
        coalign::wave_lds_sync();
        if (c_ok) a.feats[(size_t)m * a.C + lane] = fmaxf(best, 0.f);
        if (lane == 0) {
            const int cell = cd.y + cd.z * a.nx + cd.w;  // z + y*nx + x (point_pillar_scatter.py:54)
            if (cd.x >= 0 && cd.x < a.n_agents && cell >= 0 && cell < ncell)
                atomicMax(a.cellmap + (size_t)cd.x * ncell + cell, m);
        }
    }
}

// Generic path (any P, any C): channel blocks of 64, point chunks of 64.
__global__ __launch_bounds__(kWavesPerBlock * 64) void pfn_kernel(PfnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    const int wib = threadIdx.x >> 6;
    float *slab = smem + (size_t)wib * 64 * kFeatStride;
    const int gwave = blockIdx.x * kWavesPerBlock + wib;
    const int nwave = gridDim.x * kWavesPerBlock;
    const int ncell = a.ny * a.nx;

    for (int cb = 0; cb < a.C; cb += 64) {
        const int c = cb + lane;
        const bool c_ok = c < a.C;
        const ChanParams cp = load_chan(a, c);
        for (int m = gwave; m < a.M; m += nwave) {
            const int np_raw = a.npts[m];
            const int4 cd = a.coords[m];  // (agent, z, y, x)
            const int np_eff = min(max(np_raw, 0), a.P);
            const float4 *prow = a.pts + (size_t)m * a.P;
            // mean over ALL P slots divided by num_points (pillar_vfe.py:118-120)
            float sx = 0.f, sy = 0.f, sz = 0.f;
            for (int p0 = 0; p0 < a.P; p0 += 64) {
                const int p = p0 + lane;
                if (p < a.P) {
                    const float4 q = prow[p];
                    sx += q.x; sy += q.y; sz += q.z;
                }
            }
            const float npf = (float)np_raw;
            const float mx = wave_sum(sx) / npf, my = wave_sum(sy) / npf, mz = wave_sum(sz) / npf;
            const float ctr_x = (float)cd.w * a.vx + a.xo;
            const float ctr_y = (float)cd.z * a.vy + a.yo;
            const float ctr_z = (float)cd.y * a.vz + a.zo;
            // rows >= num_points are zeroed before the linear layer: Linear(0) = 0 -> BN -> `shift`
            float best = (np_eff < a.P) ? cp.shift : -INFINITY;
            for (int p0 = 0; p0 < np_eff; p0 += 64) {
                const int p = p0 + lane;
                if (p < np_eff) stage_point(a, slab, lane, prow[p], mx, my, mz, ctr_x, ctr_y, ctr_z);
                coalign::wave_lds_sync();
                const int cnt = min(64, np_eff - p0);
                for (int j = 0; j < cnt; ++j) best = fmaxf(best, point_response(cp, slab, j));
                coalign::wave_lds_sync();
            }
            if (c_ok) a.feats[(size_t)m * a.C + c] = fmaxf(best, 0.f);
            if (cb == 0 && lane == 0) {
                const int cell = cd.y + cd.z * a.nx + cd.w;  // z + y*nx + x (point_pillar_scatter.py:54)
                if (cd.x >= 0 && cd.x < a.n_agents && cell >= 0 && cell < ncell)
                    atomicMax(a.cellmap + (size_t)cd.x * ncell + cell, m);
            }
        }
    }
}

__global__ __launch_bounds__(256) void cellmap_kernel(const int4 *__restrict__ coords, int M, int n_agents, int ny, int nx,
                                                      int *__restrict__ cellmap) {
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    const int4 cd = coords[m];
    const int ncell = ny * nx;
    const int cell = cd.y + cd.z * nx + cd.w;
    if (cd.x >= 0 && cd.x < n_agents && cell >= 0 && cell < ncell) atomicMax(cellmap + (size_t)cd.x * ncell + cell, m);
}

template <int VEC>
__global__ __launch_bounds__(256) void canvas_kernel(const int *__restrict__ cellmap, const float *__restrict__ feats, int C,
                                                     int ncell, int ch_per_block, float *__restrict__ canvas) {
    const long cell0 = ((long)blockIdx.x * 256 + threadIdx.x) * VEC;
    if (cell0 >= ncell) return;
    const int agent = blockIdx.z;
    const int c0 = blockIdx.y * ch_per_block;
    const int c1 = min(C, c0 + ch_per_block);
    int id[VEC];
    if constexpr (VEC == 4) {
        const int4 t = *reinterpret_cast<const int4 *>(cellmap + (size_t)agent * ncell + cell0);
        id[0] = t.x; id[1] = t.y; id[2] = t.z; id[3] = t.w;
    } else {
        id[0] = cellmap[(size_t)agent * ncell + cell0];
    }
    bool any = false;
#pragma unroll
    for (int j = 0; j < VEC; ++j) any |= id[j] >= 0;
    float *dst = canvas + ((size_t)agent * C + c0) * ncell + cell0;
    for (int c = c0; c < c1; ++c, dst += ncell) {
        float v[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) v[j] = 0.f;
        if (any) {
#pragma unroll
            for (int j = 0; j < VEC; ++j)
                if (id[j] >= 0) v[j] = feats[(size_t)id[j] * C + c];
        }
        if constexpr (VEC == 4) {
            typedef float v4f __attribute__((ext_vector_type(4)));
            v4f o = {v[0], v[1], v[2], v[3]};
            __builtin_nontemporal_store(o, reinterpret_cast<v4f *>(dst));
        } else {
            __builtin_nontemporal_store(v[0], dst);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Fused encoder + canvas writer (P <= 64, C <= 64): one workgroup owns a strip of 256 consecutive cells of one agent.
// It looks the strip up in the cell map, runs the PFN for the (few: ~5-6 % occupancy) pillars that live there -- phase A
// lane = point, phase B lane = channel, exactly as pfn_kernel_p64 -- keeps their 64-float feature rows in LDS and streams
// the strip's 64 channel rows to the NCHW canvas with 16 B non-temporal stores.  The feature rows never make an HBM
// round trip, the PFN arithmetic of one workgroup hides behind the stores of the others, and the encoder + scatter is
// two launches (cell map, this) instead of four.  Pillars that lost their cell (duplicates) or lie outside the canvas
// still get their pillar_features row: every workgroup also checks a slice of the pillar list for such orphans.
constexpr int kStrip = 256;      // cells per workgroup
constexpr int kMaxLds = 64;      // feature rows kept in LDS (a 256-cell strip holds ~15 at LiDAR occupancy); denser strips read the rest back from L2

struct FusedArgs {
    PfnArgs p;
    float *canvas;
    int strips_per_agent, pillars_per_wg;
};

struct PillarIn { float4 q; int np; int4 cd; };

__device__ __forceinline__ PillarIn pillar_load(const PfnArgs &a, int lane, int m) {
    PillarIn in;
    in.q = make_float4(0.f, 0.f, 0.f, 0.f);
    if (lane < a.P) in.q = a.pts[(size_t)m * a.P + lane];
    in.np = a.npts[m];
    in.cd = a.coords[m];
    return in;
}

__device__ __forceinline__ float pfn_compute(const PfnArgs &a, const ChanParams &cp, float *slab, int lane, const PillarIn &in) {
    const float4 q = in.q;
    const int np_raw = in.np;
    const int4 cd = in.cd;
    const int np_eff = min(max(np_raw, 0), a.P);
    const float npf = (float)np_raw;
    const float mx = wave_sum(q.x) / npf, my = wave_sum(q.y) / npf, mz = wave_sum(q.z) / npf;
    const float ctr_x = (float)cd.w * a.vx + a.xo;
    const float ctr_y = (float)cd.z * a.vy + a.yo;
    const float ctr_z = (float)cd.y * a.vz + a.zo;
    if (lane < np_eff) stage_point(a, slab, lane, q, mx, my, mz, ctr_x, ctr_y, ctr_z);
    coalign::wave_lds_sync();
    float best = (np_eff < a.P) ? cp.shift : -INFINITY;
    int j = 0;
    for (; j + 4 <= np_eff; j += 4) {
        const float y0 = point_response(cp, slab, j), y1 = point_response(cp, slab, j + 1);
        const float y2 = point_response(cp, slab, j + 2), y3 = point_response(cp, slab, j + 3);
        best = fmaxf(fmaxf(best, fmaxf(y0, y1)), fmaxf(y2, y3));
    }
    for (; j < np_eff; ++j) best = fmaxf(best, point_response(cp, slab, j));
    coalign::wave_lds_sync();
    return fmaxf(best, 0.f);
}

__device__ __forceinline__ float pfn_one_pillar(const PfnArgs &a, const ChanParams &cp, float *slab, int lane, int m) {
    return pfn_compute(a, cp, slab, lane, pillar_load(a, lane, m));
}

// Two pillars per wavefront in phase A (P <= 32): lanes 0-31 hold pillar A's points, lanes 32-63 pillar B's.  The mean
// reduction (xor butterfly inside each 32-lane half via ds_swizzle: LDS crossbar, no memory access), the three IEEE
// divisions and the 10-d augmentation are issued once for both pillars; phase B (lane = channel) then walks A's rows
// 0.. and B's rows 32.. of the slab.  Returns relu(max) for A in `va`, for B in `vb` (mB < 0: no second pillar).
__device__ __forceinline__ float half_sum(float v) {
    // BitMode swizzle: lane' = ((lane & and_mask) | or_mask) ^ xor_mask inside each group of 32; pattern = xor<<10 | or<<5 | and
    v += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), (16 << 10) | 0x1f));
    v += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), (8 << 10) | 0x1f));
    v += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), (4 << 10) | 0x1f));
    v += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), (2 << 10) | 0x1f));
    v += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), (1 << 10) | 0x1f));
    return v;
}

__device__ __forceinline__ float rows_max(const ChanParams &cp, const float *slab, int row0, int np_eff, int P) {
    float best = (np_eff < P) ? cp.shift : -INFINITY;
    int j = 0;
    for (; j + 4 <= np_eff; j += 4) {
        const float y0 = point_response(cp, slab, row0 + j), y1 = point_response(cp, slab, row0 + j + 1);
        const float y2 = point_response(cp, slab, row0 + j + 2), y3 = point_response(cp, slab, row0 + j + 3);
        best = fmaxf(fmaxf(best, fmaxf(y0, y1)), fmaxf(y2, y3));
    }
    for (; j < np_eff; ++j) best = fmaxf(best, point_response(cp, slab, row0 + j));
    return fmaxf(best, 0.f);
}

__device__ __forceinline__ void pfn_pair(const PfnArgs &a, const ChanParams &cp, float *slab, int lane, int mA, int mB,
                                         float &va, float &vb) {
    const int half = lane >> 5, pl = lane & 31;
    const int m = half ? mB : mA;
    float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
    int np_raw = 0;
    int4 cd = make_int4(0, 0, 0, 0);
    if (m >= 0) {
        if (pl < a.P) q = a.pts[(size_t)m * a.P + pl];
        np_raw = a.npts[m];
        cd = a.coords[m];
    }
    const int np_eff = min(max(np_raw, 0), a.P);
    const float npf = (float)np_raw;
    const float mx = half_sum(q.x) / npf, my = half_sum(q.y) / npf, mz = half_sum(q.z) / npf;
    const float ctr_x = (float)cd.w * a.vx + a.xo;
    const float ctr_y = (float)cd.z * a.vy + a.yo;
    const float ctr_z = (float)cd.y * a.vz + a.zo;
    if (pl < np_eff) stage_point_pk(a, slab, lane, q, mx, my, mz, ctr_x, ctr_y, ctr_z);   // row = lane: A -> 0.., B -> 32..
    coalign::wave_lds_sync();
    const int npA = __builtin_amdgcn_readlane(np_eff, 0), npB = __builtin_amdgcn_readlane(np_eff, 32);
    va = rows_max_pk(cp, slab, 0, npA, a.P);
    vb = (mB >= 0) ? rows_max_pk(cp, slab, 32, npB, a.P) : 0.f;
    coalign::wave_lds_sync();
}

// ---------------------------------------------------------------------------------------------------------------
// Channels-last canvas [n_agents, ny, nx, C]: a pillar's feature row IS its canvas cell (C contiguous floats), so the scatter is one
// 256-byte store per pillar and the dense canvas is a plain memset (6.8 TB/s on this chip) -- no strips, no LDS feature buffer, no
// read-back.  One wavefront encodes two pillars (pfn_pair), writes their pillar_features rows and, for the pillar that owns its
// cell in the cell map ("larger row wins"), the canvas row.  Launch order on the stream: memset(canvas), memset(cell map),
// cellmap_kernel, this.
struct PairIn {     // lanes 0-31: pillar A's point / count / coords, lanes 32-63: pillar B's
    float4 q;
    int np;
    int4 cd;
};

__device__ __forceinline__ PairIn pair_load(const PfnArgs &a, int lane, int mA) {
    const int half = lane >> 5, pl = lane & 31;
    const int m = mA + half;
    PairIn in;
    in.q = make_float4(0.f, 0.f, 0.f, 0.f);
    in.np = 0;
    in.cd = make_int4(-1, 0, 0, 0);
    if (m < a.M) {
        if (pl < a.P) in.q = a.pts[(size_t)m * a.P + pl];
        in.np = a.npts[m];
        in.cd = a.coords[m];
    }
    return in;
}

// the same for two arbitrary rows (mB < 0: no second pillar)
__device__ __forceinline__ PairIn pair_load_idx(const PfnArgs &a, int lane, int mA, int mB) {
    const int half = lane >> 5, pl = lane & 31;
    const int m = half ? mB : mA;
    PairIn in;
    in.q = make_float4(0.f, 0.f, 0.f, 0.f);
    in.np = 0;
    in.cd = make_int4(-1, 0, 0, 0);
    if (m >= 0) {
        if (pl < a.P) in.q = a.pts[(size_t)m * a.P + pl];
        in.np = a.npts[m];
        in.cd = a.coords[m];
    }
    return in;
}

// pfn_pair's arithmetic on already loaded operands (same instruction sequence, so the rows are bit-identical to the NCHW route)
__device__ __forceinline__ void pair_compute(const PfnArgs &a, const ChanParams &cp, float *slab, int lane, const PairIn &in, bool hasB,
                                             float &va, float &vb) {
    const int pl = lane & 31;
    const float4 q = in.q;
    const int np_eff = min(max(in.np, 0), a.P);
    const float npf = (float)in.np;
    const float mx = half_sum(q.x) / npf, my = half_sum(q.y) / npf, mz = half_sum(q.z) / npf;
    const float ctr_x = (float)in.cd.w * a.vx + a.xo;
    const float ctr_y = (float)in.cd.z * a.vy + a.yo;
    const float ctr_z = (float)in.cd.y * a.vz + a.zo;
    if (pl < np_eff) stage_point_pk(a, slab, lane, q, mx, my, mz, ctr_x, ctr_y, ctr_z);   // row = lane: A -> 0.., B -> 32..
    coalign::wave_lds_sync();
    const int npA = __builtin_amdgcn_readlane(np_eff, 0), npB = __builtin_amdgcn_readlane(np_eff, 32);
    va = rows_max_pk(cp, slab, 0, npA, a.P);
    vb = hasB ? rows_max_pk(cp, slab, 32, npB, a.P) : 0.f;
    coalign::wave_lds_sync();
}

// ---------------------------------------------------------------------------------------------------------------
// Matrix-core encoder (P <= 32, C <= 64, no distance feature) -- the default of the channels-last routes since round 3.
// The VALU encoder above evaluates the 10-term dot product per point and channel (9.6 M vector instructions per 40 k-pillar frame,
// 26 us: VALU bound).  Two observations remove almost all of it:
//   (1) the augmented point is an affine function of the raw point (pillar_vfe.py:118-141).  With c = the pillar centre, d = p - c
//       (the reference's own f_center, one fp32 subtraction) and e = mean - c:
//           x = c + d,  f_cluster = d - e,  f_center = d
//           W f = (w_abs + w_cluster + w_center) d + w_i intensity  +  [ w_abs c - w_cluster e ]
//       i.e. a 4-term product per point plus ONE constant per (pillar, channel).  No cancellation is introduced: the large coordinates
//       only enter through w_abs c, evaluated once per pillar in fp32 exactly as large as the reference's own w_abs x term.
//   (2) that 4-term product is a [points x 4] x [4 x 64] GEMM.  Like the 3x3 convolutions (conv3x3_emu.hip) it runs on the bf16 matrix
//       cores by error-free 3-way operand splitting (x = x_h + x_m + x_l, each bf16; the six products w_h x_h, w_h x_m, w_m x_h, w_h x_l,
//       w_l x_h, w_m x_m are exact in the fp32 accumulator; dropped terms <= 2^-24 |w x|: fp32-width arithmetic).  The six products of the
//       four inputs are 24 K-slots = two v_mfma_f32_32x32x16_bf16 per 32 rows x 32 channels.
// A wavefront still takes two pillars per pass (lanes 0-31 / 32-63 = their point slots).  Matrix rows 0-15 = pillar A's slots 0-15, rows
// 16-31 = pillar B's slots 0-15 (a second tile with slots 16-31 only when either pillar has more than 16 points: 24 % of the passes at
// LiDAR statistics); a row past num_points re-reads slot 0 of its pillar, so the row maximum needs no masking.  BatchNorm's scale can be
// negative: the weights of such channels are negated (exact) so that the row reduction is always a max; max commutes with the monotone
// fp32 operations that follow (add constant, fma with alpha).  In the accumulator layout a lane holds 8 rows of pillar A and 8 of pillar B
// for one channel: 2 x 4 v_max3/v_max + one v_permlane32_swap + one v_max per 32 channels give both pillars' maxima, one more swap puts
// them in lane = channel order.  Per pass: ~160 vector instructions + 4-8 matrix instructions instead of ~480.
// Numerics: |result - fp64 evaluation| is within the fp32 rounding noise of the reference's own evaluation (tests: oracle + reference
// goldens to 1e-4 / 1e-5 of scale as before); the canvas stays an exact copy of pillar_features.
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int kRowBytes = 80;        // staged point row: [l h | m m | h h | 0 0] (4 x 16 B) + 16 B pad -> ds_read_b128 of 16 rows hits 16 distinct bank groups

struct MxChan {
    bf16x8 b1[2], b2[2];             // B operands (column = lane & 31 of channel group g): step 1 / step 2
    // epilogue parameters of channel 32 g + (lane & 31) ("half layout": lanes 0-31 work for pillar A, lanes 32-63 for pillar B)
    float wc[2][3], wen[2][3];       // weights of the centre term, negated weights of the mean-offset term
    float alpha[2], shift[2], sgn[2];
};

__device__ __forceinline__ void split3(float v, __bf16 &h, __bf16 &m, __bf16 &l) {
    h = (__bf16)v;
    const float r = v - (float)h;    // exact
    m = (__bf16)r;
    l = (__bf16)(r - (float)m);      // exact subtraction, exact conversion
}

__device__ __forceinline__ void bn_affine(const PfnArgs &a, int c, float &alpha, float &shift) {
    alpha = 1.f; shift = 0.f;
    if (c < a.C) {
        if (a.bn_w) {
            const float inv_std = 1.0f / sqrtf(a.bn_v[c] + a.eps);
            alpha = a.bn_w[c] * inv_std;
            shift = a.bn_b[c] - a.bn_m[c] * alpha;
        } else if (a.bias) {
            shift = a.bias[c];
        }
    }
}

template <bool ABS>
__device__ __forceinline__ MxChan load_mx(const PfnArgs &a, int lane) {
    MxChan mc;
    constexpr int B = ABS ? 4 : 1;
    const int half = lane >> 5, col = lane & 31;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int c = g * 32 + col;
        float w4[4] = {0.f, 0.f, 0.f, 0.f};
        float alpha, shift;
        bn_affine(a, c, alpha, shift);
        if (c < a.C) {
            const float *w = a.weight + (size_t)c * a.Cin;
#pragma unroll
            for (int k = 0; k < 3; ++k) w4[k] = ((ABS ? w[k] : 0.f) + w[B + k]) + w[B + 3 + k];
            w4[3] = w[ABS ? 3 : 0];
        }
        const float sg = alpha < 0.f ? -1.f : 1.f;
        mc.alpha[g] = alpha; mc.shift[g] = shift; mc.sgn[g] = sg;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            mc.wc[g][k] = (ABS && c < a.C) ? a.weight[(size_t)c * a.Cin + k] : 0.f;
            mc.wen[g][k] = (c < a.C) ? -a.weight[(size_t)c * a.Cin + B + k] : 0.f;
        }
        bf16x4 h, m, l;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            __bf16 th, tm, tl;
            split3(sg * w4[k], th, tm, tl);
            h[k] = th; m[k] = tm; l[k] = tl;
        }
        const bf16x4 z = {(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
        // K order of a step: half 0 = [first 4 | next 4], half 1 likewise.  step 1: A = [p_l p_h | p_m p_m], step 2: A = [p_h p_h | 0 0]
        const bf16x4 s1a = half ? m : h, s1b = half ? h : l;      // x p_l, p_h  |  x p_m, p_m
        const bf16x4 s2a = half ? z : m, s2b = half ? z : h;      // x p_h, p_h  |  0
        mc.b1[g] = __builtin_shufflevector(s1a, s1b, 0, 1, 2, 3, 4, 5, 6, 7);
        mc.b2[g] = __builtin_shufflevector(s2a, s2b, 0, 1, 2, 3, 4, 5, 6, 7);
    }
    return mc;
}

__device__ __forceinline__ void swap32(float &a, float &b) {       // lanes 32-63 of a <-> lanes 0-31 of b
    const auto q = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
    const unsigned x = q[0], y = q[1];                              // (bit-casting q[1] directly is miscompiled by hipcc 7.2: go through locals)
    a = __builtin_bit_cast(float, x);
    b = __builtin_bit_cast(float, y);
}

template <int CTRL>
__device__ __forceinline__ float dppf(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}

// sum over each 32-lane half, result in every lane of the half: four DPP steps inside the 16-lane rows, one v_permlane16_swap across rows
__device__ __forceinline__ float half_sum_dpp(float v) {
    v += dppf<0xB1>(v);              // quad_perm [1, 0, 3, 2]
    v += dppf<0x4E>(v);              // quad_perm [2, 3, 0, 1]
    v += dppf<0x141>(v);             // row_half_mirror
    v += dppf<0x140>(v);             // row_mirror
    const auto q = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
    const unsigned x = q[0], y = q[1];          // x = [r0 r0 r2 r2], y = [r1 r1 r3 r3]
    return __builtin_bit_cast(float, x) + __builtin_bit_cast(float, y);
}

// max of accumulator elements o .. o + 7: v_max3 on matrix-instruction results only (a two-operand fmaxf of raw results costs two extra
// canonicalising v_max each)
__device__ __forceinline__ float max8(const floatx16 &v, int o) {
    const float t1 = fmaxf(fmaxf(v[o], v[o + 1]), v[o + 2]), t2 = fmaxf(fmaxf(v[o + 3], v[o + 4]), v[o + 5]);
    const float t3 = fmaxf(fmaxf(v[o + 6], v[o + 7]), t1);
    return fmaxf(t2, t3);
}

// plain v_max_f32 for values the compiler cannot prove canonical (results of a lane swap): no canonicalising pre-pass
__device__ __forceinline__ float vmax_raw(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// rows: this wavefront's 64 x kRowBytes LDS rows (part 3 of every row zeroed once by the caller).
// -> y[g], g = 0, 1: relu(max) of channel 32 g + (lane & 31) for pillar A (lanes 0-31) / pillar B (lanes 32-63) -- the "half layout".
template <bool ABS>
__device__ __forceinline__ void mx_pair_half(const PfnArgs &a, const MxChan &mc, char *rows, int lane, const PairIn &in, bool hasB, float (&y)[2]) {
    const float4 q = in.q;
    const int np_eff = min(max(in.np, 0), a.P);
    // mean over ALL P slots divided by num_points (pillar_vfe.py:118-120); the division as v_rcp_f32 + multiply (<= 1.5 ulp of the mean)
    const float rn = __builtin_amdgcn_rcpf((float)in.np);
    const float ctr_x = (float)in.cd.w * a.vx + a.xo;
    const float ctr_y = (float)in.cd.z * a.vy + a.yo;
    const float ctr_z = (float)in.cd.y * a.vz + a.zo;
    const float ex = half_sum_dpp(q.x) * rn - ctr_x, ey = half_sum_dpp(q.y) * rn - ctr_y, ez = half_sum_dpp(q.z) * rn - ctr_z;
    const float d[4] = {q.x - ctr_x, q.y - ctr_y, q.z - ctr_z, q.w};
    bf16x4 h, m, l;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        __bf16 th, tm, tl;
        split3(d[k], th, tm, tl);
        h[k] = th; m[k] = tm; l[k] = tl;
    }
    const uint2 hu = __builtin_bit_cast(uint2, h), mu = __builtin_bit_cast(uint2, m), lu = __builtin_bit_cast(uint2, l);
    uint4 *dst = reinterpret_cast<uint4 *>(rows + lane * kRowBytes);         // row = lane: A -> 0.., B -> 32..
    dst[0] = make_uint4(lu.x, lu.y, hu.x, hu.y);
    dst[1] = make_uint4(mu.x, mu.y, mu.x, mu.y);
    dst[2] = make_uint4(hu.x, hu.y, hu.x, hu.y);
    coalign::wave_lds_sync();
    const int npA = __builtin_amdgcn_readlane(np_eff, 0), npB = hasB ? __builtin_amdgcn_readlane(np_eff, 32) : 0;
    const int i = lane & 31, half = lane >> 5, pil = i >> 4;
    const int npp = pil ? npB : npA;
    float r[2];
    auto tile = [&](int t, float (&o)[2]) {
        const int slot = (i & 15) + 16 * t;
        const int row = pil * 32 + (slot < npp ? slot : 0);
        const char *src = rows + row * kRowBytes + half * 16;
        const bf16x8 a1 = *reinterpret_cast<const bf16x8 *>(src), a2 = *reinterpret_cast<const bf16x8 *>(src + 32);
        floatx16 acc[2];
        const floatx16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < 2; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, mc.b1[g], zero, 0, 0, 0);       // the small products first
#pragma unroll
        for (int g = 0; g < 2; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, mc.b2[g], acc[g], 0, 0, 0);
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            float mA = max8(acc[g], 0), mB = max8(acc[g], 8);  // rows 0-15 = pillar A, 16-31 = pillar B; this lane holds 8 of each
            swap32(mA, mB);                                     // mA = [A.lo B.lo], mB = [A.hi B.hi]
            o[g] = vmax_raw(mA, mB);                            // lanes 0-31: A's channel 32 g + lane, lanes 32-63: B's channel 32 g + lane - 32
        }
    };
    tile(0, r);
    if (max(npA, npB) > 16) {                                   // wave-uniform
        float s2[2];
        tile(1, s2);
        r[0] = vmax_raw(r[0], s2[0]); r[1] = vmax_raw(r[1], s2[1]);
    }
    coalign::wave_lds_sync();
    // the per-pillar constant  w_abs c - w_cluster e: every lane already holds its own pillar's centre / mean offset / point count
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        float b = mc.wen[g][0] * ex;
        b = fmaf(mc.wen[g][1], ey, b); b = fmaf(mc.wen[g][2], ez, b);
        if constexpr (ABS) { b = fmaf(mc.wc[g][0], ctr_x, b); b = fmaf(mc.wc[g][1], ctr_y, b); b = fmaf(mc.wc[g][2], ctr_z, b); }
        float v = fmaf(mc.sgn[g] * r[g] + b, mc.alpha[g], mc.shift[g]);
        if (np_eff < a.P) v = fmaxf(v, mc.shift[g]);           // padded rows: Linear(0) = 0 -> BN -> shift
        if (np_eff == 0) v = mc.shift[g];                       // (documented deviation: the reference divides by zero here)
        y[g] = fmaxf(v, 0.f);
    }
}

// the same with the result in lane = channel order: va / vb = pillar A's / B's 64 channels
template <bool ABS>
__device__ __forceinline__ void pair_compute_mx(const PfnArgs &a, const MxChan &mc, char *rows, int lane, const PairIn &in, bool hasB,
                                                float &va, float &vb) {
    float y[2];
    mx_pair_half<ABS>(a, mc, rows, lane, in, hasB, y);
    swap32(y[0], y[1]);                                         // [A ch 0-31 | B ch 0-31], [A ch 32-63 | B ch 32-63] -> [A 0-63], [B 0-63]
    va = y[0];
    vb = hasB ? y[1] : 0.f;
}

// ENC as in pillar_rows_nhwc_kernel: the strips' pillar pairs go through the same pair_compute / pair_compute_mx, so the NCHW and the
// channels-last route produce bit-identical rows.
template <int ENC>
__global__ __launch_bounds__(256) void pillar_canvas_kernel(FusedArgs f) {
    const PfnArgs &a = f.p;
    constexpr int kWaveLds = ENC ? 64 * kRowBytes : 64 * kFeatStride * (int)sizeof(float);
    __shared__ __attribute__((aligned(16))) char slab_bytes[4 * kWaveLds];
    __shared__ __attribute__((aligned(16))) float featbuf[kMaxLds * 64];
    __shared__ int work_m[kStrip];           // pillar rows living in this strip
    __shared__ int orphan_m[256];            // orphans found in this workgroup's slice of the pillar list (normally none)
    __shared__ int slot_of_cell[kStrip];     // cell -> index into work_m / featbuf, -1 = empty
    __shared__ int n_occ, n_orph;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int ncell = a.ny * a.nx;
    const int agent = blockIdx.y;
    const int cell0 = blockIdx.x * kStrip;
    const int wg = blockIdx.y * gridDim.x + blockIdx.x;
    if (tid == 0) { n_occ = 0; n_orph = 0; }
    // ---- issue both independent loads first: this thread's cell of the strip, and one pillar of this workgroup's slice of
    //      the pillar list (orphan check: duplicate-cell losers and out-of-canvas pillars still need their feature row)
    const int cell = cell0 + tid;
    const int id = (cell < ncell) ? a.cellmap[(size_t)agent * ncell + cell] : -1;
    const int m_o = (tid < f.pillars_per_wg) ? wg * f.pillars_per_wg + tid : a.M;
    int4 cd_o = make_int4(-1, 0, 0, 0);
    if (m_o < a.M) cd_o = a.coords[m_o];
    __syncthreads();
    int slot = -1;
    if (id >= 0) { slot = atomicAdd(&n_occ, 1); work_m[slot] = id; }   // order inside the list is irrelevant
    slot_of_cell[tid] = slot;
    int winner = -2;                                                   // second hop of the orphan check, overlaps the PFN below
    if (m_o < a.M) {
        const int c = cd_o.y + cd_o.z * a.nx + cd_o.w;
        winner = (cd_o.x >= 0 && cd_o.x < a.n_agents && c >= 0 && c < ncell) ? a.cellmap[(size_t)cd_o.x * ncell + c] : -1;
    }
    __syncthreads();
    const int nocc = n_occ;
    // ---- PFN for the strip's pillars
    const ChanParams cp = load_chan(a, lane);
    char *wave_lds = slab_bytes + wv * kWaveLds;
    float *slab = reinterpret_cast<float *>(wave_lds);
    MxChan mc;
    if constexpr (ENC != 0) {
        mc = load_mx<ENC == 1>(a, lane);
        *reinterpret_cast<uint4 *>(wave_lds + lane * kRowBytes + 48) = make_uint4(0u, 0u, 0u, 0u);
    }
    auto one_pillar = [&](int m) -> float {
        if constexpr (ENC == 0) return pfn_one_pillar(a, cp, slab, lane, m);
        else {
            float va, vb;
            pair_compute_mx<ENC == 1>(a, mc, wave_lds, lane, pair_load_idx(a, lane, m, -1), false, va, vb);
            return va;
        }
    };
    if (a.P <= 32) {
        for (int w = 2 * wv; w < nocc; w += 8) {          // two pillars per wave and pass
            const int mA = work_m[w], mB = (w + 1 < nocc) ? work_m[w + 1] : -1;
            float va, vb;
            if constexpr (ENC == 0) pfn_pair(a, cp, slab, lane, mA, mB, va, vb);
            else pair_compute_mx<ENC == 1>(a, mc, wave_lds, lane, pair_load_idx(a, lane, mA, mB), mB >= 0, va, vb);
            if (lane < a.C) {
                a.feats[(size_t)mA * a.C + lane] = va;
                if (w < kMaxLds) featbuf[w * 64 + lane] = va;
                if (mB >= 0) {
                    a.feats[(size_t)mB * a.C + lane] = vb;
                    if (w + 1 < kMaxLds) featbuf[(w + 1) * 64 + lane] = vb;
                }
            }
        }
    } else {
        for (int w = wv; w < nocc; w += 4) {
            const int m = work_m[w];
            const float v = one_pillar(m);
            if (lane < a.C) {
                a.feats[(size_t)m * a.C + lane] = v;
                if (w < kMaxLds) featbuf[w * 64 + lane] = v;
            }
        }
    }
    if (nocc > kMaxLds) __threadfence();     // overflow rows are read back from L2 below
    // ---- one barrier publishes featbuf AND tells whether anybody found an orphan (well-formed input: nobody)
    const bool orphan = (m_o < a.M) && (winner != m_o);
    if (__syncthreads_or(orphan || f.pillars_per_wg > 256)) {
        if (orphan) orphan_m[atomicAdd(&n_orph, 1)] = m_o;
        __syncthreads();
        for (int w = wv; w < n_orph; w += 4) {
            const float v = one_pillar(orphan_m[w]);
            if (lane < a.C) a.feats[(size_t)orphan_m[w] * a.C + lane] = v;
        }
        for (int r0 = 256; r0 < f.pillars_per_wg; r0 += 256) {        // slices longer than the block (huge M, tiny canvas)
            __syncthreads();
            if (tid == 0) n_orph = 0;
            __syncthreads();
            const int m = wg * f.pillars_per_wg + r0 + tid;
            if (r0 + tid < f.pillars_per_wg && m < a.M) {
                const int4 cd = a.coords[m];
                const int c = cd.y + cd.z * a.nx + cd.w;
                const bool placed = cd.x >= 0 && cd.x < a.n_agents && c >= 0 && c < ncell && a.cellmap[(size_t)cd.x * ncell + c] == m;
                if (!placed) orphan_m[atomicAdd(&n_orph, 1)] = m;
            }
            __syncthreads();
            for (int w = wv; w < n_orph; w += 4) {
                const float v = one_pillar(orphan_m[w]);
                if (lane < a.C) a.feats[(size_t)orphan_m[w] * a.C + lane] = v;
            }
        }
    }
    // ---- the occupied groups: thread = 4 consecutive cells x (C/4) channels, 16 B stores of LDS-resident feature rows
    // (stores are issued only now: a wave that stores and then loads stalls -- vmcnt retires in order and counts stores --
    //  and an early zero-writer wave measured 2x slower, so the whole strip goes out after the encoder)
    const int q4 = lane * 4;                 // first of this thread's 4 cells inside the strip
    const int cpw = (a.C + 3) / 4;           // channels per wave
    const int c_lo = wv * cpw, c_hi = min(a.C, c_lo + cpw);
    const int gcell = cell0 + q4;
    const bool vec = (ncell % 4 == 0);
    int sl[4];
    bool any = false;
#pragma unroll
    for (int j = 0; j < 4; ++j) { sl[j] = slot_of_cell[q4 + j]; any |= sl[j] >= 0; }
    if (gcell >= ncell) return;
    float *dst = f.canvas + ((size_t)agent * a.C + c_lo) * ncell + gcell;
    for (int c = c_lo; c < c_hi; ++c, dst += ncell) {
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (any) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (sl[j] >= 0) {
                    if (sl[j] < kMaxLds) v[j] = featbuf[sl[j] * 64 + c];
                    else v[j] = __hip_atomic_load(a.feats + (size_t)work_m[sl[j]] * a.C + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        if (vec) {
            typedef float v4f __attribute__((ext_vector_type(4)));
            v4f o = {v[0], v[1], v[2], v[3]};
            __builtin_nontemporal_store(o, reinterpret_cast<v4f *>(dst));
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (gcell + j < ncell) dst[j] = v[j];
        }
    }
}


// Persistent wavefronts: the channel parameters are loaded once, the operands of the NEXT pair and the cell-map entry of the
// current one are in flight while the current pair is encoded -- one exposed memory round trip per wave instead of four per pair.
// `dest` (optional): dest[m] = the pillar's canvas slot agent * ncell + cell, or -1 when it lost its cell or lies outside the canvas --
// what the persistent-canvas form clears before the next frame.
// `reset_cellmap`: the winner puts its cell-map entry back to -1 once it has read it (a loser that looks later sees -1 instead of the
// winner's row -- not its own either way), so a persistent cell map is all -1 again when the kernel ends and needs no memset per frame.
// (This kernel runs the fp32 VALU encoder: feature sets with the distance term, tensors past 4 GB, COALIGN_PILLAR_MFMA=0.  The default is
// pillar_rows_mx_kernel below.)  `a.feats` may be NULL (callers that only consume the canvas skip 256 B of writes per pillar).
__global__ __launch_bounds__(kWavesPerBlock * 64) void pillar_rows_nhwc_kernel(PfnArgs a, float *__restrict__ canvas, int *__restrict__ dest,
                                                                               int reset_cellmap) {
    __shared__ __attribute__((aligned(16))) float slabs[kWavesPerBlock * 64 * kFeatStride];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float *slab = slabs + wv * 64 * kFeatStride;
    const int gwave = blockIdx.x * kWavesPerBlock + wv, nwave = gridDim.x * kWavesPerBlock;
    if (a.M_dev) a.M = min(max(*a.M_dev, 0), a.M);
    if (a.M_dev && dest && blockIdx.x == 0 && threadIdx.x == 0) dest[-1] = a.M;      // the device-count form keeps "rows written" in front of the list
    const int npairs = (a.M + 1) / 2;
    if (gwave >= npairs) return;
    const int ncell = a.ny * a.nx;
    // (Handing the pairs out dynamically -- one atomicAdd per pair on a global counter, fetched an iteration ahead -- was measured at
    //  281 us instead of 26: 20 000 same-address atomics serialise in one L2 channel.  Static round-robin it is.)
    // cell-map slot of a lane's pillar (lanes 0 / 32 speak for A / B), -1 when the pillar lies outside the canvas
    auto slot_of = [&](const PairIn &p) -> long {
        const int cell = p.cd.y + p.cd.z * a.nx + p.cd.w;                // z + y * nx + x (point_pillar_scatter.py:54)
        const bool ok = p.cd.x >= 0 && p.cd.x < a.n_agents && cell >= 0 && cell < ncell;
        return ok ? (long)p.cd.x * ncell + cell : -1;
    };
    PairIn nxt = pair_load(a, lane, 2 * gwave);
    const ChanParams cp = load_chan(a, lane);
    long slot_nxt = slot_of(nxt);
    int owner_nxt = a.unique ? 0 : a.cellmap[slot_nxt < 0 ? 0 : slot_nxt];
    for (int pair = gwave; pair < npairs; pair += nwave) {
        const PairIn in = nxt;
        const long slot = slot_nxt;
        const int owner = owner_nxt;
        const int mA = 2 * pair;
        const bool hasB = mA + 1 < a.M;
        const bool more = pair + nwave < npairs;
        if (more) nxt = pair_load(a, lane, 2 * (pair + nwave));
        float va, vb;
        pair_compute(a, cp, slab, lane, in, hasB, va, vb);
        // The next pair's owner lookup goes out BEFORE this pair's stores: vmcnt retires in order and counts stores, so a load
        // issued behind the stores would make the next iteration wait for them to reach memory (measured: 5 us per pair).
        if (more) {
            slot_nxt = slot_of(nxt);
            if (!a.unique) owner_nxt = a.cellmap[slot_nxt < 0 ? 0 : slot_nxt];
        }
        const int m_lane = mA + (lane >> 5);
        const bool win = slot >= 0 && (a.unique || owner == m_lane) && m_lane < a.M;
        const int winA = __builtin_amdgcn_readlane((int)win, 0), winB = __builtin_amdgcn_readlane((int)win, 32);
        const unsigned slo = (unsigned)(unsigned long)slot, shi = (unsigned)((unsigned long)slot >> 32);
        const size_t slotA = ((size_t)(unsigned)__builtin_amdgcn_readlane((int)shi, 0) << 32) | (unsigned)__builtin_amdgcn_readlane((int)slo, 0);
        const size_t slotB = ((size_t)(unsigned)__builtin_amdgcn_readlane((int)shi, 32) << 32) | (unsigned)__builtin_amdgcn_readlane((int)slo, 32);
        if (lane < a.C) {
            if (a.feats) a.feats[(size_t)mA * a.C + lane] = va;
            if (winA) canvas[slotA * a.C + lane] = va;
            if (hasB) {
                if (a.feats) a.feats[(size_t)(mA + 1) * a.C + lane] = vb;
                if (winB) canvas[slotB * a.C + lane] = vb;
            }
        }
        if (lane == 0 || (lane == 32 && hasB)) {
            if (dest) dest[m_lane] = win ? (int)slot : -1;
            if (reset_cellmap && win) a.cellmap[slot] = -1;
        }
    }
}

// Persistent canvas, one launch: zero the rows the PREVIOUS frame wrote (its dest list) and enter the NEW frame's pillars into the
// cell map (which the previous rows kernel left all -1).  Grid-stride, so the counts may live on the device (M_dev: the voxeliser's
// count word, M = capacity; dest[-1] = the previous call's count): the launch geometry never depends on them.
__global__ __launch_bounds__(256) void pillar_prep_kernel(const int *__restrict__ dest_prev, int M_prev, int C, float *__restrict__ canvas,
                                                          const int4 *__restrict__ coords, int M, const int *__restrict__ M_dev, int n_agents,
                                                          int ny, int nx, int *__restrict__ cellmap) {
    const int c4 = C / 4;
    if (M_dev) {
        M = min(max(*M_dev, 0), M);
        M_prev = min(max(dest_prev[-1], 0), M_prev);
    }
    const long stride = (long)gridDim.x * 256;
    const long n_clear = (long)M_prev * c4;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < n_clear; idx += stride) {
        const long m = idx / c4;
        const int d = dest_prev[m];
        if (d >= 0) reinterpret_cast<float4 *>(canvas + (size_t)d * C)[(int)(idx - m * c4)] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (!cellmap) return;                         // unique pillars: no cell map
    const int ncell = ny * nx;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < M; idx += stride) {
        const int4 cd = coords[idx];
        const int cell = cd.y + cd.z * nx + cd.w;
        if (cd.x >= 0 && cd.x < n_agents && cell >= 0 && cell < ncell) atomicMax(cellmap + (size_t)cd.x * ncell + cell, (int)idx);
    }
}

// The rows kernel of the matrix-core encoder.  Against pillar_rows_nhwc_kernel:
//   * everything stays in the half layout -- lanes 0-31 work for pillar A, lanes 32-63 for pillar B from the point loads to the stores (a
//     store instruction writes 128 B of A's row and 128 B of B's), nothing is broadcast through scalar registers on the way;
//   * all addresses are 32-bit byte offsets from scalar bases (the launcher checks the sizes);
//   * a wavefront owns a CONTIGUOUS run of up to 32 pairs and works through it in ROUNDS of five pairs.  Memory latency is paid once per
//     WAVEFRONT: in the prologue lane j reads pillar j's count and coordinates and looks its cell up in the cell map (two dependent round
//     trips, all 64 pillars of the run in parallel) and leaves a 32-byte record in LDS, while the first round's points (5 x 1 KB) travel
//     straight into LDS (global_load_lds: no staging registers) and the channel parameters are fetched and split.  From then on the points
//     of round r + 1 are in flight into the OTHER point buffer while the five passes of round r run out of LDS.
//     (History, profiles/round3/pillar_rows_ablation.txt: round-robin pairs with register prefetch 20 us, of which 12 us remained with the
//      arithmetic AND the stores switched off -- 4096 waves x 5 dependent passes x one exposed round trip each; one round of five pairs per
//      wavefront, 4096 wavefronts: 20 us / 12.5 us -- every wavefront still paid prologue + one exposed transfer for five pairs of work.)
typedef __attribute__((address_space(3))) void *lptr_ps_t;
#ifdef COALIGN_LAB
constexpr bool kPillarLab = true;       // the ablation bits of PfnArgs::debug (no arithmetic / no stores) exist in the laboratory build only
#else
constexpr bool kPillarLab = false;
#endif
constexpr int kRunPairs = 32;                               // pairs per wavefront at most: one lane per pillar in the prologue

// ROUND: pairs per round (ROUND KB of points per wavefront and buffer)
template <bool ABS, int ROUND>
__global__ __launch_bounds__(kWavesPerBlock * 64) void pillar_rows_mx_kernel(PfnArgs a, float *__restrict__ canvas, int *__restrict__ dest, int reset_cellmap) {
    constexpr int kRound = ROUND;
    constexpr int kMxWaveLds = 64 * kRowBytes + 2 * kRound * 1024 + 64 * 32;      // staged rows | two point buffers | the run's pillar records
    __shared__ __attribute__((aligned(16))) char lds[kWavesPerBlock * kMxWaveLds];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), half = lane >> 5, col = lane & 31;
    char *wave_lds = lds + wv * kMxWaveLds;
    char *pbuf = wave_lds + 64 * kRowBytes;
    char *meta = pbuf + 2 * kRound * 1024;
    const int gwave = blockIdx.x * kWavesPerBlock + wv, nwave = gridDim.x * kWavesPerBlock;
    if (a.M_dev) a.M = min(max(*a.M_dev, 0), a.M);
    if (a.M_dev && dest && blockIdx.x == 0 && threadIdx.x == 0) dest[-1] = a.M;      // the device-count form keeps "rows written" in front of the list
    const int npairs = (a.M + 1) / 2;
    const int per_wave = (npairs + nwave - 1) / nwave;                                  // <= kRunPairs: the launcher sizes the grid on the capacity
    const int p0 = gwave * per_wave, p1 = min(p0 + per_wave, npairs);
    if (p0 >= p1) return;
    const int ncell = a.ny * a.nx;
    const char *pts_b = reinterpret_cast<const char *>(a.pts);
    const char *np_b = reinterpret_cast<const char *>(a.npts), *cd_b = reinterpret_cast<const char *>(a.coords);
    char *cm_b = reinterpret_cast<char *>(a.cellmap), *feat_b = reinterpret_cast<char *>(a.feats), *cv_b = reinterpret_cast<char *>(canvas),
         *dest_b = reinterpret_cast<char *>(dest);
    // a round's points go straight into LDS (issued from inline assembly: after a BUILTIN LDS-DMA hipcc puts s_waitcnt vmcnt(0) in front of every
    // later LDS read it cannot prove unrelated -- here in front of every pass, i.e. each pass would wait for the previous pass's stores)
    auto issue_round = [&](int r0, int buf) {
        const int nr = min(kRound, p1 - r0);
#pragma unroll
        for (int k = 0; k < kRound; ++k) {
            if (k < nr) {
                const int m = min(2 * (r0 + k) + half, a.M - 1);                       // a pillar B past the end re-reads the last pillar; its lanes store nothing
                const char *src = pts_b + (unsigned)(m * a.P + min(col, a.P - 1)) * 16u;
                const unsigned dst = (unsigned)(size_t)(lptr_ps_t)(pbuf + (buf * kRound + k) * 1024);   // wave-uniform LDS byte address -> M0; lane i lands at dst + 16 i
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(__builtin_amdgcn_readfirstlane(dst)), "v"(src) : "memory", "m0");
            }
        }
    };
    issue_round(p0, 0);                                                                // in flight while the records are built and the channel parameters are split
    if (lane < 2 * (p1 - p0)) {                                                        // counts, coordinates, cell-map entries of the whole run: once per wavefront
        const int mj = min(2 * p0 + lane, a.M - 1);
        const int np_j = *reinterpret_cast<const int *>(np_b + (unsigned)mj * 4u);
        const int4 cd_j = *reinterpret_cast<const int4 *>(cd_b + (unsigned)mj * 16u);
        const int cell = cd_j.y + cd_j.z * a.nx + cd_j.w;                              // z + y * nx + x (point_pillar_scatter.py:54)
        const bool ok = cd_j.x >= 0 && cd_j.x < a.n_agents && cell >= 0 && cell < ncell;
        const int slot_j = ok ? cd_j.x * ncell + cell : -1;
        const int own_j = a.unique ? 0 : *reinterpret_cast<const int *>(cm_b + (unsigned)max(slot_j, 0) * 4u);
        int4 *rec = reinterpret_cast<int4 *>(meta + lane * 32);
        rec[0] = make_int4(np_j, cd_j.y, cd_j.z, cd_j.w);
        rec[1] = make_int4(slot_j, own_j, 0, 0);
    }
    const MxChan mc = load_mx<ABS>(a, lane);
    *reinterpret_cast<uint4 *>(wave_lds + lane * kRowBytes + 48) = make_uint4(0u, 0u, 0u, 0u);          // part 3 of every row: the zero half of step 2
    const bool ch0 = col < a.C, ch1 = 32 + col < a.C;
    int buf = 0;
    for (int r0 = p0; r0 < p1; r0 += kRound, buf ^= 1) {
        const int nr = min(kRound, p1 - r0);
        __builtin_amdgcn_s_waitcnt(0x0F70);                                            // vmcnt(0): this round's points are in LDS (and the last round's stores are out)
        coalign::wave_lds_sync();
        if (r0 + kRound < p1) issue_round(r0 + kRound, buf ^ 1);                       // the other buffer was consumed by the passes of the round before this one
        const char *pb = pbuf + buf * (kRound * 1024);
#pragma unroll
        for (int k = 0; k < kRound; ++k) {
            if (k < nr) {
                const int pair = r0 + k;
                const int m = 2 * pair + half;
                const bool live = m < a.M, hasB = 2 * pair + 1 < a.M;
                PairIn in;
                in.q = *reinterpret_cast<const float4 *>(pb + k * 1024 + lane * 16);
                if (a.P < 32 && col >= a.P) in.q = make_float4(0.f, 0.f, 0.f, 0.f);
                const int4 *rec = reinterpret_cast<const int4 *>(meta + (2 * (pair - p0) + half) * 32);
                const int4 r0v = rec[0], r1v = rec[1];
                in.np = r0v.x;
                in.cd = make_int4(0, r0v.y, r0v.z, r0v.w);
                const int slot = r1v.x, owner = r1v.y;
                float y[2];
                if (kPillarLab && (a.debug & 4)) { y[0] = in.q.x + (float)in.np; y[1] = in.q.y + (float)in.cd.w; }
                else mx_pair_half<ABS>(a, mc, wave_lds, lane, in, hasB, y);
                const bool win = live && slot >= 0 && (a.unique || owner == m);
                if (live) {
                    if (feat_b && !(kPillarLab && (a.debug & 1))) {
                        const unsigned fo = ((unsigned)m * (unsigned)a.C + (unsigned)col) * 4u;
                        if (ch0) *reinterpret_cast<float *>(feat_b + fo) = y[0];
                        if (ch1) *reinterpret_cast<float *>(feat_b + fo + 128u) = y[1];
                    }
                    if (win && !(kPillarLab && (a.debug & 2))) {
                        const unsigned co = ((unsigned)slot * (unsigned)a.C + (unsigned)col) * 4u;
                        if (ch0) *reinterpret_cast<float *>(cv_b + co) = y[0];
                        if (ch1) *reinterpret_cast<float *>(cv_b + co + 128u) = y[1];
                    }
                    if (col == 0) {
                        if (dest_b) *reinterpret_cast<int *>(dest_b + (unsigned)m * 4u) = win ? slot : -1;
                        if (reset_cellmap && win) *reinterpret_cast<int *>(cm_b + (unsigned)slot * 4u) = -1;
                    }
                }
            }
        }
    }
}

// The channels-last rows kernel with the encoder the feature set allows (the distance feature is not affine in the point: VALU encoder).
bool pillar_mfma_enabled() {
    static const bool on = [] { const char *e = getenv("COALIGN_PILLAR_MFMA"); return !(e && e[0] == '0'); }();
    return on;
}

void launch_rows(const PfnArgs &a_in, float *canvas, int *dest, int reset_cellmap, int blocks, hipStream_t stream) {
    static const int debug = coalign::lab_env("COALIGN_PILLAR_DEBUG", 0);                  // laboratory build only (the product kernel has no debug branches: kPillarLab)
    static const int blocks_override = coalign::lab_env("COALIGN_PILLAR_BLOCKS", 0);
    PfnArgs a = a_in;
    a.debug = debug;
    if (blocks_override > 0 && blocks > blocks_override) blocks = blocks_override;
    const dim3 grid(blocks), block(kWavesPerBlock * 64);
    // 32-bit byte offsets in the matrix-core kernel: pillars, feature rows and canvas each below 4 GB
    const bool small = (size_t)a.M * a.P * 16 < ((size_t)1 << 32) && (size_t)a.M * a.C * 4 < ((size_t)1 << 32) &&
                       (size_t)a.n_agents * a.ny * a.nx * a.C * 4 < ((size_t)1 << 32);
    if (a.with_dist || !small || !pillar_mfma_enabled()) {
        hipLaunchKernelGGL(pillar_rows_nhwc_kernel, grid, block, 0, stream, a, canvas, dest, reset_cellmap);
        return;
    }
    // matrix-core kernel: two workgroups (8 wavefronts, 2 x 68 KB of LDS) per CU; a wavefront takes two rounds' pairs unless the pillar count asks
    // for more than the 512 resident workgroups -- then up to kRunPairs each, then more workgroups.  (Measured at 40 000 pillars, rocprofv3:
    // 18.9 us; rounds of three pairs -- 129 registers, three workgroups per CU -- 19.1 us; 16 / 20 / 32 pairs per wavefront 23.8 / 23.6 / 32.2 us;
    // without arithmetic and stores 9-10 us, without arithmetic 12.2 us, without stores 16.6 us: the six-product matrix steps and their
    // reductions, ~750 issue cycles per pair and SIMD, are what is left.  profiles/round3/pillar_rows_ablation.txt)
    static const int pairs_target = [] { const int v = coalign::lab_env("COALIGN_PILLAR_PAIRS", 0); return v > 0 && v <= kRunPairs ? v : 0; }();
    constexpr int kRound = 5;
    const int target = pairs_target ? pairs_target : 2 * kRound, resident = 512;
    const int pairs = (a.M + 1) / 2;
    int mx_blocks = (pairs + kWavesPerBlock * target - 1) / (kWavesPerBlock * target);
    if (mx_blocks > resident) mx_blocks = resident;
    const int need = (pairs + kWavesPerBlock * kRunPairs - 1) / (kWavesPerBlock * kRunPairs);
    if (mx_blocks < need) mx_blocks = need;
    if (blocks_override > 0 && mx_blocks > blocks_override && blocks_override >= need) mx_blocks = blocks_override;
    const dim3 mx_grid(mx_blocks < 1 ? 1 : mx_blocks);
    if (a.use_abs) hipLaunchKernelGGL((pillar_rows_mx_kernel<true, kRound>), mx_grid, block, 0, stream, a, canvas, dest, reset_cellmap);
    else hipLaunchKernelGGL((pillar_rows_mx_kernel<false, kRound>), mx_grid, block, 0, stream, a, canvas, dest, reset_cellmap);
}

int launch_canvas(const int *cellmap, const float *feats, int C, int ncell, int n_agents, float *canvas, hipStream_t stream) {
    const int ch_per_block = 16;
    const int ych = (C + ch_per_block - 1) / ch_per_block;
    if (ncell % 4 == 0) {
        dim3 grid((ncell / 4 + 255) / 256, ych, n_agents);
        hipLaunchKernelGGL(canvas_kernel<4>, grid, dim3(256), 0, stream, cellmap, feats, C, ncell, ch_per_block, canvas);
    } else {
        dim3 grid((ncell + 255) / 256, ych, n_agents);
        hipLaunchKernelGGL(canvas_kernel<1>, grid, dim3(256), 0, stream, cellmap, feats, C, ncell, ch_per_block, canvas);
    }
    return coalign::check_launch();
}

}  // namespace

extern "C" {

size_t coalign_pillar_scatter_workspace_bytes(int n_agents, int ny, int nx) {
    if (n_agents <= 0 || ny <= 0 || nx <= 0) return 0;
    return coalign::align_up((size_t)n_agents * ny * nx * sizeof(int), 256);
}

static int pillar_vfe_scatter_impl(const float *voxel_features, const int32_t *voxel_num_points, const int32_t *voxel_coords,
                               int M, int P, const float *pfn_weight, const float *pfn_bias, const float *bn_weight,
                               const float *bn_bias, const float *bn_mean, const float *bn_var, float bn_eps, int C,
                               int use_absolute_xyz, int with_distance, const double *voxel_size, const double *range_min,
                               int n_agents, int ny, int nx, float *pillar_features, float *canvas, void *workspace,
                               size_t workspace_bytes, void *stream_, bool nhwc) {
    using namespace coalign;
    hipStream_t stream = (hipStream_t)stream_;
    if (M < 0 || P <= 0 || C <= 0 || n_agents <= 0 || ny <= 0 || nx <= 0) return COALIGN_ERR_BAD_SHAPE;
    if (!pfn_weight || !voxel_size || !range_min || !canvas || !workspace) return COALIGN_ERR_NULL_POINTER;
    if (M > 0 && (!voxel_features || !voxel_num_points || !voxel_coords || !pillar_features)) return COALIGN_ERR_NULL_POINTER;
    const bool has_bn = bn_weight || bn_bias || bn_mean || bn_var;
    if (has_bn && !(bn_weight && bn_bias && bn_mean && bn_var)) return COALIGN_ERR_NULL_POINTER;
    if ((size_t)ny * nx > (size_t)INT32_MAX) return COALIGN_ERR_BAD_SHAPE;
    if (workspace_bytes < coalign_pillar_scatter_workspace_bytes(n_agents, ny, nx)) return COALIGN_ERR_WORKSPACE;
    const int Cin = (use_absolute_xyz ? 4 : 1) + 6 + (with_distance ? 1 : 0);
    if (Cin > kFeatStride) return COALIGN_ERR_UNSUPPORTED;
    if (nhwc && (P > 32 || C > 64)) return COALIGN_ERR_UNSUPPORTED;       // two pillars per wavefront, lane = channel

    const int ncell = ny * nx;
    int *cellmap = (int *)workspace;
    int rc = fill_words(cellmap, (size_t)n_agents * ncell, 0xFFFFFFFFu, stream);       // kernels, not memset nodes: see common.h
    if (rc) return rc;
    if (nhwc && (rc = fill_words(canvas, (size_t)n_agents * ncell * C, 0u, stream))) return rc;

    if (M > 0) {
        PfnArgs a{};
        a.pts = (const float4 *)voxel_features; a.npts = voxel_num_points; a.coords = (const int4 *)voxel_coords;
        a.M = M; a.P = P;
        a.weight = pfn_weight; a.bias = pfn_bias; a.bn_w = bn_weight; a.bn_b = bn_bias; a.bn_m = bn_mean; a.bn_v = bn_var;
        a.eps = bn_eps; a.C = C; a.Cin = Cin; a.use_abs = use_absolute_xyz; a.with_dist = with_distance;
        a.vx = (float)voxel_size[0]; a.vy = (float)voxel_size[1]; a.vz = (float)voxel_size[2];
        a.xo = (float)(voxel_size[0] / 2 + range_min[0]);
        a.yo = (float)(voxel_size[1] / 2 + range_min[1]);
        a.zo = (float)(voxel_size[2] / 2 + range_min[2]);
        a.n_agents = n_agents; a.ny = ny; a.nx = nx; a.feats = pillar_features; a.cellmap = cellmap;
        const size_t lds = (size_t)kWavesPerBlock * 64 * kFeatStride * sizeof(float);
        if (nhwc) {
            hipLaunchKernelGGL(cellmap_kernel, dim3((M + 255) / 256), dim3(256), 0, stream, (const int4 *)voxel_coords, M, n_agents,
                               ny, nx, cellmap);
            if ((rc = check_launch())) return rc;
            const int pairs = (M + 1) / 2;
            const int want = (pairs + kWavesPerBlock - 1) / kWavesPerBlock;
            const int cap = 256 * 4;                       // 4 workgroups (16 wavefronts) per CU are resident at this kernel's 112 registers
            launch_rows(a, canvas, nullptr, 0, want < cap ? want : cap, stream);
            return check_launch();
        }
        if (P <= 64 && C <= 64 && !coalign::lab_env("COALIGN_UNFUSED_PILLARS", 0)) {
            // cell map first, then ONE fused encoder + canvas pass
            hipLaunchKernelGGL(cellmap_kernel, dim3((M + 255) / 256), dim3(256), 0, stream, (const int4 *)voxel_coords, M, n_agents,
                               ny, nx, cellmap);
            if ((rc = check_launch())) return rc;
            FusedArgs f;
            f.p = a; f.canvas = canvas;
            f.strips_per_agent = (ncell + kStrip - 1) / kStrip;
            const long nwg = (long)f.strips_per_agent * n_agents;
            f.pillars_per_wg = (int)((M + nwg - 1) / nwg);
            const dim3 grid(f.strips_per_agent, n_agents);
            if (a.with_dist || P > 32 || !pillar_mfma_enabled()) hipLaunchKernelGGL(pillar_canvas_kernel<0>, grid, dim3(256), 0, stream, f);
            else if (a.use_abs) hipLaunchKernelGGL(pillar_canvas_kernel<1>, grid, dim3(256), 0, stream, f);
            else hipLaunchKernelGGL(pillar_canvas_kernel<2>, grid, dim3(256), 0, stream, f);
            return check_launch();
        } else if (P <= 64 && C <= 64) {   // ~4 pillars per wave: enough to amortise the parameter load, prefetch hides the rest
            const int blocks = (int)min((long)(M + 4 * kWavesPerBlock - 1) / (4 * kWavesPerBlock), (long)256 * 16);
            hipLaunchKernelGGL(pfn_kernel_p64, dim3(max(blocks, 1)), dim3(kWavesPerBlock * 64), lds, stream, a);
        } else {
            const int blocks = (int)min((long)(M + kWavesPerBlock - 1) / kWavesPerBlock, (long)256 * 8);
            hipLaunchKernelGGL(pfn_kernel, dim3(blocks), dim3(kWavesPerBlock * 64), lds, stream, a);
        }
        if ((rc = check_launch())) return rc;
    }

    if (nhwc) return COALIGN_OK;          // M == 0: the memset was the whole canvas
    return launch_canvas(cellmap, pillar_features, C, ncell, n_agents, canvas, stream);
}

int coalign_pillar_vfe_scatter(const float *voxel_features, const int32_t *voxel_num_points, const int32_t *voxel_coords,
                               int M, int P, const float *pfn_weight, const float *pfn_bias, const float *bn_weight,
                               const float *bn_bias, const float *bn_mean, const float *bn_var, float bn_eps, int C,
                               int use_absolute_xyz, int with_distance, const double *voxel_size, const double *range_min,
                               int n_agents, int ny, int nx, float *pillar_features, float *canvas, void *workspace,
                               size_t workspace_bytes, void *stream) {
    return pillar_vfe_scatter_impl(voxel_features, voxel_num_points, voxel_coords, M, P, pfn_weight, pfn_bias, bn_weight, bn_bias, bn_mean, bn_var,
                                   bn_eps, C, use_absolute_xyz, with_distance, voxel_size, range_min, n_agents, ny, nx, pillar_features, canvas,
                                   workspace, workspace_bytes, stream, false);
}

int coalign_pillar_vfe_scatter_nhwc(const float *voxel_features, const int32_t *voxel_num_points, const int32_t *voxel_coords,
                                    int M, int P, const float *pfn_weight, const float *pfn_bias, const float *bn_weight,
                                    const float *bn_bias, const float *bn_mean, const float *bn_var, float bn_eps, int C,
                                    int use_absolute_xyz, int with_distance, const double *voxel_size, const double *range_min,
                                    int n_agents, int ny, int nx, float *pillar_features, float *canvas, void *workspace,
                                    size_t workspace_bytes, void *stream) {
    return pillar_vfe_scatter_impl(voxel_features, voxel_num_points, voxel_coords, M, P, pfn_weight, pfn_bias, bn_weight, bn_bias, bn_mean, bn_var,
                                   bn_eps, C, use_absolute_xyz, with_distance, voxel_size, range_min, n_agents, ny, nx, pillar_features, canvas,
                                   workspace, workspace_bytes, stream, true);
}

static int encode_persistent_impl(const float *voxel_features, const int32_t *voxel_num_points, const int32_t *voxel_coords, int M, const int32_t *M_dev,
                                  int P, const float *pfn_weight, const float *pfn_bias, const float *bn_weight, const float *bn_bias,
                                  const float *bn_mean, const float *bn_var, float bn_eps, int C, int use_absolute_xyz, int with_distance,
                                  const double *voxel_size, const double *range_min, int n_agents, int ny, int nx, float *pillar_features,
                                  int32_t *dest, int M_prev, float *canvas, int32_t *cellmap, int unique, void *stream_) {
    using namespace coalign;
    hipStream_t stream = (hipStream_t)stream_;
    if (M < 0 || M_prev < 0 || P <= 0 || P > 32 || C < 4 || C > 64 || C % 4 || n_agents <= 0 || ny <= 0 || nx <= 0) return COALIGN_ERR_BAD_SHAPE;
    if (!pfn_weight || !voxel_size || !range_min || !canvas || (!cellmap && !unique) || ((M > 0 || M_prev > 0) && !dest)) return COALIGN_ERR_NULL_POINTER;
    if (M > 0 && (!voxel_features || !voxel_num_points || !voxel_coords)) return COALIGN_ERR_NULL_POINTER;
    const bool has_bn = bn_weight || bn_bias || bn_mean || bn_var;
    if (has_bn && !(bn_weight && bn_bias && bn_mean && bn_var)) return COALIGN_ERR_NULL_POINTER;
    if ((size_t)n_agents * ny * nx > (size_t)INT32_MAX) return COALIGN_ERR_BAD_SHAPE;
    const int Cin = (use_absolute_xyz ? 4 : 1) + 6 + (with_distance ? 1 : 0);
    if (Cin > kFeatStride) return COALIGN_ERR_UNSUPPORTED;
    const long threads = (long)M_prev * (C / 4) > M ? (long)M_prev * (C / 4) : M;
    if (threads > 0) {
        const long blocks = (threads + 255) / 256;
        hipLaunchKernelGGL(pillar_prep_kernel, dim3((unsigned)(M_dev && blocks > 4096 ? 4096 : blocks)), dim3(256), 0, stream, dest, M_prev, C, canvas,
                           (const int4 *)voxel_coords, M, M_dev, n_agents, ny, nx, unique ? (int *)nullptr : cellmap);
        int rc = check_launch();
        if (rc) return rc;
    }
    if (M == 0) return COALIGN_OK;
    PfnArgs a{};
    a.pts = (const float4 *)voxel_features; a.npts = voxel_num_points; a.coords = (const int4 *)voxel_coords;
    a.M = M; a.P = P;
    a.weight = pfn_weight; a.bias = pfn_bias; a.bn_w = bn_weight; a.bn_b = bn_bias; a.bn_m = bn_mean; a.bn_v = bn_var;
    a.eps = bn_eps; a.C = C; a.Cin = Cin; a.use_abs = use_absolute_xyz; a.with_dist = with_distance;
    a.vx = (float)voxel_size[0]; a.vy = (float)voxel_size[1]; a.vz = (float)voxel_size[2];
    a.xo = (float)(voxel_size[0] / 2 + range_min[0]);
    a.yo = (float)(voxel_size[1] / 2 + range_min[1]);
    a.zo = (float)(voxel_size[2] / 2 + range_min[2]);
    a.n_agents = n_agents; a.ny = ny; a.nx = nx; a.feats = pillar_features; a.cellmap = cellmap;
    a.M_dev = M_dev; a.unique = unique ? 1 : 0;
    const int pairs = (M + 1) / 2;
    const int want = (pairs + kWavesPerBlock - 1) / kWavesPerBlock;
    const int cap = 256 * 4;
    launch_rows(a, canvas, dest, unique ? 0 : 1, want < cap ? want : cap, stream);
    return check_launch();
}

int coalign_pillar_encode_persistent(const float *voxel_features, const int32_t *voxel_num_points, const int32_t *voxel_coords, int M, int P,
                                     const float *pfn_weight, const float *pfn_bias, const float *bn_weight, const float *bn_bias,
                                     const float *bn_mean, const float *bn_var, float bn_eps, int C, int use_absolute_xyz, int with_distance,
                                     const double *voxel_size, const double *range_min, int n_agents, int ny, int nx, float *pillar_features,
                                     int32_t *dest, int M_prev, float *canvas, int32_t *cellmap, void *stream) {
    if (M > 0 && !pillar_features) return COALIGN_ERR_NULL_POINTER;
    return encode_persistent_impl(voxel_features, voxel_num_points, voxel_coords, M, nullptr, P, pfn_weight, pfn_bias, bn_weight, bn_bias, bn_mean, bn_var,
                                  bn_eps, C, use_absolute_xyz, with_distance, voxel_size, range_min, n_agents, ny, nx, pillar_features, dest, M_prev,
                                  canvas, cellmap, 0, stream);
}

int coalign_pillar_encode_stream(const float *voxel_features, const int32_t *voxel_num_points, const int32_t *voxel_coords, int M_capacity,
                                 const int32_t *M_dev, int P, const float *pfn_weight, const float *pfn_bias, const float *bn_weight,
                                 const float *bn_bias, const float *bn_mean, const float *bn_var, float bn_eps, int C, int use_absolute_xyz,
                                 int with_distance, const double *voxel_size, const double *range_min, int n_agents, int ny, int nx,
                                 float *pillar_features, int32_t *dest_state, float *canvas, int32_t *cellmap, int unique_cells, void *stream) {
    if (!M_dev || !dest_state) return COALIGN_ERR_NULL_POINTER;
    if (M_capacity <= 0) return COALIGN_ERR_BAD_SHAPE;
    // dest_state[0] = rows the previous call wrote (device side), dest_state[1 ..] = their canvas slots
    return encode_persistent_impl(voxel_features, voxel_num_points, voxel_coords, M_capacity, M_dev, P, pfn_weight, pfn_bias, bn_weight, bn_bias, bn_mean,
                                  bn_var, bn_eps, C, use_absolute_xyz, with_distance, voxel_size, range_min, n_agents, ny, nx, pillar_features,
                                  dest_state + 1, M_capacity, canvas, cellmap, unique_cells, stream);
}

int coalign_scatter_to_bev(const float *pillar_features, const int32_t *voxel_coords, int M, int C, int n_agents, int ny,
                           int nx, float *canvas, void *workspace, size_t workspace_bytes, void *stream_) {
    using namespace coalign;
    hipStream_t stream = (hipStream_t)stream_;
    if (M < 0 || C <= 0 || n_agents <= 0 || ny <= 0 || nx <= 0) return COALIGN_ERR_BAD_SHAPE;
    if (!canvas || !workspace || (M > 0 && (!pillar_features || !voxel_coords))) return COALIGN_ERR_NULL_POINTER;
    if ((size_t)ny * nx > (size_t)INT32_MAX) return COALIGN_ERR_BAD_SHAPE;
    if (workspace_bytes < coalign_pillar_scatter_workspace_bytes(n_agents, ny, nx)) return COALIGN_ERR_WORKSPACE;
    const int ncell = ny * nx;
    int *cellmap = (int *)workspace;
    int rc = fill_words(cellmap, (size_t)n_agents * ncell, 0xFFFFFFFFu, stream);
    if (rc) return rc;
    if (M > 0) {
        hipLaunchKernelGGL(cellmap_kernel, dim3((M + 255) / 256), dim3(256), 0, stream, (const int4 *)voxel_coords, M, n_agents,
                           ny, nx, cellmap);
        if ((rc = check_launch())) return rc;
    }
    return launch_canvas(cellmap, pillar_features, C, ncell, n_agents, canvas, stream);
}

}  // extern "C"
