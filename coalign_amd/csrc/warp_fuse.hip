// Pose-aware affine warp + multi-agent fusion (attention / max / none), gfx950.
//
// Reference semantics (see include/coalign_amd.h): warp_affine_simple
// (opencood/models/sub_modules/torch_transformation_utils.py:322-331) feeding AttFusion / MaxFusion
// (opencood/models/fuse_modules/fusion_in_one.py:51-136).  The reference materialises the warped copy of every
// agent, permutes it to (H*W, N, C), runs two bmm + softmax over all N rows and throws N-1 of them away.
//
// This kernel fuses everything and computes the ego row only.  One workgroup owns an 8x8 tile of output pixels and
// ALL channels of ALL agents of the frame; thread = (pixel = tid & 63, channel group g = tid >> 6), CPT channels each.
//
//  phase 0  the sampling geometry (float64 grid -> float32 coords -> top-left tap + 4 bilinear weights) is computed
//           ONCE per (pixel, agent) -- the waves split the agents -- and parked in LDS together with the bounding box
//           of the tile's footprint in the source plane (wave min/max reduction).
//  phase 1  LDS-staged gather.  For a rigid pose (rotation + translation at unit scale, what normalize_pairwise_tfm
//           produces) the footprint of an 8x8 tile fits a 16x16 source patch for ANY yaw.  Per (agent, channel) a wave
//           loads that patch with ONE coalesced 16 B/lane load (lane -> row = lane>>2, 4 floats at col 4*(lane&3);
//           out-of-image lanes contribute zeros = grid_sample's zero padding), writes it to a wave-private LDS slab
//           (row stride 24 floats: conflict-free for axis-aligned reads) and every pixel fetches its 4 taps with two
//           ds_read2_b32.  v1 issued four dword gathers per value through the texture-address path (4 lanes/clk);
//           this is one 16 B/lane load per 64 values.  The warped values stay in REGISTERS (X[NA][CPT]).
//  phase 2  <X0, Xn> partials cross the waves through LDS (G x N x 64 floats), softmax over the agents and the
//           weighted sum are per-thread; max / none modes skip the reduction.
//  fallback non-rigid theta (footprint larger than the patch), W % 4 != 0 or unaligned planes: the same workgroup
//           takes a generic two-pass direct-gather route (no register-resident X, clamped taps).
//
// The QK^T / AV contractions are per-pixel dot products of *different* vectors for every pixel (a batch of 1xC . CxN
// GEMVs, N <= 8, no operand shared between pixels): there is no tile to feed an MFMA with and the f32 MFMA rate equals
// the f32 VALU rate on gfx950, so they are lane-local FMAs on registers; the kernel is bound by the gather.
// Workgroup -> tile order is XCD-aware so neighbouring tiles (which share source lines) share an L2.
#include "common.h"

namespace {

constexpr int kPatchRows = 16;
constexpr int kPatchStride = 24;                        // floats per patch row (16 used), keeps 16 B alignment
constexpr int kPatchFloats = kPatchRows * kPatchStride;  // 384 floats = 1.5 KB per wave

struct WarpArgs {
    const float *x;       // [n, C, H, W] of this frame
    const double *theta;  // [n, 2, 3]
    float *out;           // ATT/MAX: [C, Ho, Wo]; NONE: [n, C, Ho, Wo]
    int n, C, H, W, Ho, Wo, tiles_x, ntiles, mode, vec_ok;
    float sqrt_dim;
    int rows[8];          // physical row of x holding logical agent i (identity unless the caller routes agents, coalign_warp_fuse_rows)
};

struct Tap {       // per (agent, pixel), parked in LDS
    int idx;       // staged: patch-relative index of the top-left tap
    float w00, w01, w10, w11;
};

struct Origin {    // per agent, wave-uniform: patch origin (x aligned to 4), used extent, fits-the-slab flag
    int ax0, py0, nc4, nrows, fit;
};

// grid_sample coordinates of output pixel (ox, oy) in agent n's plane, reference arithmetic:
// F.affine_grid on a float64 theta (x_n = (2j+1)/W - 1 in float64) -> .to(float32) -> (g + 1) * (size / 2) - 0.5
__device__ __forceinline__ void sample_coords(const WarpArgs &a, int n, int ox, int oy, float &ix, float &iy) {
    const double xn = (2.0 * ox + 1.0) / a.Wo - 1.0;
    const double yn = (2.0 * oy + 1.0) / a.Ho - 1.0;
    const double *th = a.theta + n * 6;
    const float gx = (float)(th[0] * xn + th[1] * yn + th[2]);
    const float gy = (float)(th[3] * xn + th[4] * yn + th[5]);
    ix = (gx + 1.f) * ((float)a.W / 2) - 0.5f;
    iy = (gy + 1.f) * ((float)a.H / 2) - 0.5f;
}

__device__ __forceinline__ int wave_min(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
    return v;
}

// direct (un-staged) bilinear sample with clamped addresses and masked weights -- the generic fallback
__device__ __forceinline__ float sample_direct(const WarpArgs &a, const float *__restrict__ plane, float ix, float iy) {
    if (!(ix > -1.f && ix < (float)a.W && iy > -1.f && iy < (float)a.H)) return 0.f;
    const float x0f = floorf(ix), y0f = floorf(iy);
    const float tx = ix - x0f, ty = iy - y0f, ex = 1.f - tx, ey = 1.f - ty;
    const int x0 = (int)x0f, y0 = (int)y0f;
    const bool vx0 = x0 >= 0, vx1 = x0 + 1 <= a.W - 1, vy0 = y0 >= 0, vy1 = y0 + 1 <= a.H - 1;
    const int xc0 = vx0 ? x0 : 0, yc0 = vy0 ? y0 : 0, xc1 = vx1 ? x0 + 1 : a.W - 1, yc1 = vy1 ? y0 + 1 : a.H - 1;
    const float v00 = plane[yc0 * a.W + xc0], v01 = plane[yc0 * a.W + xc1];
    const float v10 = plane[yc1 * a.W + xc0], v11 = plane[yc1 * a.W + xc1];
    const float w00 = (vx0 && vy0) ? ey * ex : 0.f, w01 = (vx1 && vy0) ? ey * tx : 0.f;
    const float w10 = (vx0 && vy1) ? ty * ex : 0.f, w11 = (vx1 && vy1) ? ty * tx : 0.f;
    return v00 * w00 + v01 * w01 + v10 * w10 + v11 * w11;
}

template <int NA>
__device__ __forceinline__ void softmax_weights(const WarpArgs &a, const float *red, int G, int px, float (&s)[NA]) {
    float smax = -INFINITY;
#pragma unroll
    for (int n = 0; n < NA; ++n) {
        float acc = 0.f;
        for (int gg = 0; gg < G; ++gg) acc += red[(gg * NA + n) * 64 + px];
        s[n] = acc / a.sqrt_dim;                      // score / np.sqrt(C)   (att_fuse.py:44)
        if (n < a.n) smax = fmaxf(smax, s[n]);
    }
    float den = 0.f;
#pragma unroll
    for (int n = 0; n < NA; ++n) {
        s[n] = (n < a.n) ? expf(s[n] - smax) : 0.f;
        den += s[n];
    }
#pragma unroll
    for (int n = 0; n < NA; ++n) s[n] = s[n] / den;
}

template <int NA, int CPT, bool KEEP, int MAXT, int PD>
__global__ __launch_bounds__(MAXT, (NA > 5 && MAXT <= 512) ? 2 : ((CPT == 8 && NA <= 5 && MAXT <= 512) ? 5 : 4)) void warp_fuse_kernel(WarpArgs a) {   // waves / SIMD the register allocation must allow
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int px = threadIdx.x & 63, G = blockDim.x >> 6;
    const int g = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave id: provably uniform -> scalar plane addresses
    // LDS carve: [G][kPatchFloats] patches | [G][NA][64] score partials | [NA][64] taps | [NA] origins
    float *patch = smem + (size_t)g * kPatchFloats;
    float *red = smem + (size_t)G * kPatchFloats;
    Tap *taps = reinterpret_cast<Tap *>(red + (size_t)G * NA * 64);
    Origin *origins = reinterpret_cast<Origin *>(taps + NA * 64);

    const int tile = coalign::xcd_remap(blockIdx.x, a.ntiles);
    const int oy = (tile / a.tiles_x) * 8 + (px >> 3), ox = (tile % a.tiles_x) * 8 + (px & 7);
    const bool pix_ok = oy < a.Ho && ox < a.Wo;
    const int HW = a.H * a.W;
    const size_t HWo = (size_t)a.Ho * a.Wo;
    const int c_base = g * CPT;
    const size_t opix = (size_t)oy * a.Wo + ox;

    // ---------------------------------------------------------------- phase 0: taps + footprint per agent
    for (int n = g; n < a.n; n += G) {
        Tap t;
        t.idx = 0; t.w00 = t.w01 = t.w10 = t.w11 = 0.f;
        int x0 = 0, y0 = 0;
        bool live = false;
        if (pix_ok) {
            float ix, iy;
            sample_coords(a, n, ox, oy, ix, iy);
            if (ix > -1.f && ix < (float)a.W && iy > -1.f && iy < (float)a.H) {
                const float x0f = floorf(ix), y0f = floorf(iy);
                const float tx = ix - x0f, ty = iy - y0f, ex = 1.f - tx, ey = 1.f - ty;
                x0 = (int)x0f; y0 = (int)y0f;
                const bool vx0 = x0 >= 0, vx1 = x0 + 1 <= a.W - 1, vy0 = y0 >= 0, vy1 = y0 + 1 <= a.H - 1;
                t.w00 = (vx0 && vy0) ? ey * ex : 0.f;      // zero padding: out-of-image taps weigh nothing, so the patch
                t.w01 = (vx1 && vy0) ? ey * tx : 0.f;      // may hold anything at those positions (clamped loads below)
                t.w10 = (vx0 && vy1) ? ty * ex : 0.f;
                t.w11 = (vx1 && vy1) ? ty * tx : 0.f;
                live = true;
            }
        }
        const int big = 1 << 28;
        const int xmin = wave_min(live ? x0 : big), xmax = wave_max(live ? x0 + 1 : -big);
        const int ymin = wave_min(live ? y0 : big), ymax = wave_max(live ? y0 + 1 : -big);
        Origin o;
        o.ax0 = 0; o.py0 = 0; o.nc4 = 1; o.nrows = 0; o.fit = 1;
        if (xmin != big) {
            o.ax0 = (xmin >> 2) << 2;                  // floor to a multiple of 4 (also for -1)
            o.py0 = ymin;
            o.nc4 = ((xmax - o.ax0) >> 2) + 1;         // float4 columns actually touched (1..4 when it fits)
            o.nrows = ymax - ymin + 1;
            o.fit = (o.nc4 <= 4) && (o.nrows <= kPatchRows);
        }
        if (live && o.fit) t.idx = (y0 - o.py0) * kPatchStride + (x0 - o.ax0);   // else: weights 0 or fallback route
        taps[n * 64 + px] = t;
        if (px == 0) origins[n] = o;
    }
    __syncthreads();
    bool fast = a.vec_ok != 0;
    for (int n = 0; n < a.n; ++n) fast = fast && origins[n].fit;

    if (fast) {
        // ------------------------------------------------------------ phase 1: LDS-staged gather
        // Work items j = (agent n, channel k) are walked in order with a rolling prefetch ring: the 16 B/lane patch
        // load of item j + D is issued when item j is consumed, so every wave keeps D loads (D KB) in flight -- the
        // kernel is latency bound otherwise (PMC: 79 % of wave cycles waiting on vmcnt with 4 loads in flight).
        constexpr int D = PD;                             // prefetch distance (patch loads in flight per wave)
        constexpr int ITEMS = NA * CPT;
        // Only the bounding box of the tile's footprint is fetched: the lanes are laid over it row-major with nc4 float4 per
        // row (nc4 * nrows <= 64).  Every lane issues the load -- box lanes at their (clamped into the image) position,
        // the others at byte 0 of the plane (one broadcast line) -- and writes LDS, the others into the unused columns
        // 16..23 of the slab: no per-lane branch anywhere in the item loop.
        unsigned voff[NA];   // byte offset of this lane's patch slot inside a channel plane
        int loff[NA];        // LDS offset (floats) of the slot
#pragma unroll
        for (int n = 0; n < NA; ++n) {
            voff[n] = 0u; loff[n] = (px & 15) * kPatchStride + 16 + ((px >> 4) & 1) * 4;
            if (n < a.n) {
                const Origin o = origins[n];
                const int prow = (o.nc4 == 4) ? (px >> 2) : (o.nc4 == 3) ? (px * 43 >> 7) : (o.nc4 == 2) ? (px >> 1) : px;
                const int pc4 = (px - prow * o.nc4) << 2;
                if (prow < o.nrows) {
                    loff[n] = prow * kPatchStride + pc4;
                    const int gy = min(max(o.py0 + prow, 0), a.H - 1), gx = min(max(o.ax0 + pc4, 0), a.W - 4);
                    voff[n] = (unsigned)(gy * a.W + gx) * 4u;
                }
            }
        }
        const char *xb = reinterpret_cast<const char *>(a.x);
        auto fetch = [&](int n, int k) -> float4 {       // n, k are compile-time after unrolling
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            const int c = c_base + k;
            if (n < a.n && c < a.C) {                      // wave-uniform: scalar branch, scalar plane address
                const char *plane = xb + ((size_t)a.rows[n] * a.C + c) * HW * sizeof(float);
                v = *reinterpret_cast<const float4 *>(plane + voff[n]);
            }
            return v;
        };
        auto resample = [&](float4 v, const Tap &t, int lo) -> float {   // patch slice -> LDS slab -> this pixel's 4 taps
            *reinterpret_cast<float4 *>(patch + lo) = v;
            coalign::wave_lds_sync();
            const float *pr = patch + t.idx;
            const float v00 = pr[0], v01 = pr[1], v10 = pr[kPatchStride], v11 = pr[kPatchStride + 1];
            const float r = v00 * t.w00 + v01 * t.w01 + v10 * t.w10 + v11 * t.w11;
            coalign::wave_lds_sync();
            return r;
        };

        if constexpr (KEEP) {
            float X[NA][CPT];                              // warped values stay in registers
            if (a.n == NA && c_base + CPT <= a.C) {
                // Every (agent, channel) slot of the template exists for this wave: a straight-line stream of
                // ITEMS x {16 B/lane load, ds_write, 2 ds_read2, 7 VALU}.  The kernel is instruction-issue bound (removing
                // every load AND the LDS round trip only took it from 31 to 18 us), so the per-item overhead is stripped:
                // no per-item conditions, the plane pointer advances by a scalar add.  The empty asm pins program order --
                // without it hipcc hoists dozens of the (now unconditional) loads to the top and spills.
                const size_t stride = (size_t)HW * sizeof(float);
                const char *agent0 = xb + (size_t)c_base * stride;
                const size_t agent_stride = (size_t)a.C * stride;
                const char *pf = agent0 + (size_t)a.rows[0] * agent_stride;   // plane of the next item to prefetch (uniform)
                float4 ring[D];
#pragma unroll
                for (int j = 0; j < D; ++j) {
                    ring[j] = *reinterpret_cast<const float4 *>(pf + voff[j / CPT]);
                    pf = ((j + 1) % CPT == 0) ? agent0 + (size_t)a.rows[((j + 1) / CPT) % NA] * agent_stride : pf + stride;
                }
                Tap t = taps[px];
#pragma unroll
                for (int j = 0; j < ITEMS; ++j) {
                    const int n = j / CPT, k = j % CPT;
                    if (k == 0 && n > 0) t = taps[n * 64 + px];
                    const float4 v = ring[j % D];
                    if (j + D < ITEMS) {
                        ring[j % D] = *reinterpret_cast<const float4 *>(pf + voff[(j + D) / CPT]);
                        pf = ((j + D + 1) % CPT == 0) ? agent0 + (size_t)a.rows[((j + D + 1) / CPT) % NA] * agent_stride : pf + stride;
                    }
                    X[n][k] = resample(v, t, loff[n]);
#pragma unroll
                    for (int q = 0; q < NA; ++q) asm volatile("" : "+v"(voff[q]));   // later loads may not move above here
                }
            } else {
            float4 ring[D];
#pragma unroll
            for (int j = 0; j < D; ++j) ring[j] = fetch(j / CPT, j % CPT);
            Tap t = taps[px];
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) {
                const int n = j / CPT, k = j % CPT;
                if (k == 0 && n > 0 && n < a.n) t = taps[n * 64 + px];
                const float4 v = ring[j % D];
                if (j + D < ITEMS) ring[j % D] = fetch((j + D) / CPT, (j + D) % CPT);
                X[n][k] = (n < a.n && c_base + k < a.C) ? resample(v, t, loff[n]) : 0.f;
                __builtin_amdgcn_sched_barrier(0);   // keep the prefetch of item j + D inside iteration j
            }
            }
            if (a.mode == COALIGN_FUSE_ATT) {
#pragma unroll
                for (int n = 0; n < NA; ++n) {
                    float p = 0.f;
#pragma unroll
                    for (int k = 0; k < CPT; ++k) p = fmaf(X[0][k], X[n][k], p);
                    red[(g * NA + n) * 64 + px] = p;
                }
                __syncthreads();
                float s[NA];
                softmax_weights<NA>(a, red, G, px, s);
#pragma unroll
                for (int k = 0; k < CPT; ++k) {
                    const int c = c_base + k;
                    float o = 0.f;
#pragma unroll
                    for (int n = 0; n < NA; ++n) o = fmaf(s[n], X[n][k], o);
                    if (pix_ok && c < a.C) a.out[(size_t)c * HWo + opix] = o;
                }
            } else if (a.mode == COALIGN_FUSE_MAX) {
#pragma unroll
                for (int k = 0; k < CPT; ++k) {
                    const int c = c_base + k;
                    float m = -INFINITY;
#pragma unroll
                    for (int n = 0; n < NA; ++n)
                        if (n < a.n) m = fmaxf(m, X[n][k]);
                    if (pix_ok && c < a.C) a.out[(size_t)c * HWo + opix] = m;
                }
            } else {
#pragma unroll
                for (int n = 0; n < NA; ++n)
#pragma unroll
                    for (int k = 0; k < CPT; ++k) {
                        const int c = c_base + k;
                        if (pix_ok && n < a.n && c < a.C) a.out[((size_t)n * a.C + c) * HWo + opix] = X[n][k];
                    }
            }
        } else {
            // register-light variant for many agents x many channels: only the ego values and the output
            // accumulators are resident (2 x CPT registers); the staged gather runs twice for the attention mode.
            float X0[CPT], acc[CPT];
#pragma unroll
            for (int k = 0; k < CPT; ++k) { X0[k] = 0.f; acc[k] = (a.mode == COALIGN_FUSE_MAX) ? -INFINITY : 0.f; }
            float s[NA];
#pragma unroll
            for (int n = 0; n < NA; ++n) s[n] = 0.f;
            float4 ring[D];
            Tap t = taps[px];
            if (a.mode == COALIGN_FUSE_ATT) {
                float p = 0.f;
#pragma unroll
                for (int j = 0; j < D; ++j) ring[j] = fetch(j / CPT, j % CPT);
#pragma unroll
                for (int j = 0; j < ITEMS; ++j) {
                    const int n = j / CPT, k = j % CPT;
                    if (k == 0) { p = 0.f; if (n > 0 && n < a.n) t = taps[n * 64 + px]; }
                    const float4 v4 = ring[j % D];
                    if (j + D < ITEMS) ring[j % D] = fetch((j + D) / CPT, (j + D) % CPT);
                    const float v = (n < a.n && c_base + k < a.C) ? resample(v4, t, loff[n]) : 0.f;
                    if (n == 0) X0[k] = v;
                    p = fmaf(X0[k], v, p);
                    if (k == CPT - 1) red[(g * NA + n) * 64 + px] = p;
                    __builtin_amdgcn_sched_barrier(0);
                }
                __syncthreads();
                softmax_weights<NA>(a, red, G, px, s);
                t = taps[px];
            }
#pragma unroll
            for (int j = 0; j < D; ++j) ring[j] = fetch(j / CPT, j % CPT);
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) {
                const int n = j / CPT, k = j % CPT;
                if (k == 0 && n > 0 && n < a.n) t = taps[n * 64 + px];
                const float4 v4 = ring[j % D];
                if (j + D < ITEMS) ring[j % D] = fetch((j + D) / CPT, (j + D) % CPT);
                const int c = c_base + k;
                if (n < a.n && c < a.C) {
                    const float v = resample(v4, t, loff[n]);
                    if (a.mode == COALIGN_FUSE_ATT) acc[k] = fmaf(s[n], v, acc[k]);
                    else if (a.mode == COALIGN_FUSE_MAX) acc[k] = fmaxf(acc[k], v);
                    else if (pix_ok) a.out[((size_t)n * a.C + c) * HWo + opix] = v;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (a.mode != COALIGN_FUSE_NONE) {
#pragma unroll
                for (int k = 0; k < CPT; ++k) {
                    const int c = c_base + k;
                    if (pix_ok && c < a.C) a.out[(size_t)c * HWo + opix] = acc[k];
                }
            }
        }
        return;
    }

    // ---------------------------------------------------------------- generic fallback: direct gather, two passes
    float ixs[NA], iys[NA];
#pragma unroll
    for (int n = 0; n < NA; ++n) {
        ixs[n] = -2.f; iys[n] = -2.f;
        if (n < a.n && pix_ok) sample_coords(a, n, ox, oy, ixs[n], iys[n]);
    }
    float s[NA];
    if (a.mode == COALIGN_FUSE_ATT) {
        float part[NA];
#pragma unroll
        for (int n = 0; n < NA; ++n) part[n] = 0.f;
        for (int k = 0; k < CPT; ++k) {
            const int c = c_base + k;
            if (c >= a.C) break;
            const float v0 = sample_direct(a, a.x + ((size_t)a.rows[0] * a.C + c) * HW, ixs[0], iys[0]);
#pragma unroll
            for (int n = 0; n < NA; ++n) {
                if (n < a.n) {
                    const float v = n == 0 ? v0 : sample_direct(a, a.x + ((size_t)a.rows[n] * a.C + c) * HW, ixs[n], iys[n]);
                    part[n] = fmaf(v0, v, part[n]);
                }
            }
        }
#pragma unroll
        for (int n = 0; n < NA; ++n) red[(g * NA + n) * 64 + px] = part[n];
        __syncthreads();
        softmax_weights<NA>(a, red, G, px, s);
    }
    for (int k = 0; k < CPT; ++k) {
        const int c = c_base + k;
        if (c >= a.C) break;
        float o = 0.f, m = -INFINITY;
#pragma unroll
        for (int n = 0; n < NA; ++n) {
            if (n < a.n) {
                const float v = sample_direct(a, a.x + ((size_t)a.rows[n] * a.C + c) * HW, ixs[n], iys[n]);
                if (a.mode == COALIGN_FUSE_ATT) o = fmaf(s[n], v, o);
                else if (a.mode == COALIGN_FUSE_MAX) m = fmaxf(m, v);
                else if (pix_ok) a.out[((size_t)n * a.C + c) * HWo + opix] = v;
            }
        }
        if (pix_ok && a.mode == COALIGN_FUSE_ATT) a.out[(size_t)c * HWo + opix] = o;
        if (pix_ok && a.mode == COALIGN_FUSE_MAX) a.out[(size_t)c * HWo + opix] = m;
    }
}

template <int NA, int CPT, bool KEEP, int MAXT, int PD>
int launch(const WarpArgs &a, int G, hipStream_t stream) {
    const size_t lds = ((size_t)G * kPatchFloats + (size_t)G * NA * 64) * sizeof(float) + (size_t)NA * 64 * sizeof(Tap) +
                       (size_t)NA * sizeof(Origin);
    hipLaunchKernelGGL((warp_fuse_kernel<NA, CPT, KEEP, MAXT, PD>), dim3(a.ntiles), dim3(G * 64), lds, stream, a);
    return coalign::check_launch();
}

// Channels per thread: 8 while that keeps the block within 16 waves (more, smaller waves = more loads in flight and a
// finer spread over the 256 CUs), 16 above.  The register-resident variant (KEEP) needs NA * CPT registers for the
// warped values plus the prefetch ring, so it is used while that fits the per-thread budget the block size leaves
// (512-entry file per SIMD lane / waves per SIMD: <= 4 waves -> 512, 8 -> 256, 16 -> 128); otherwise two-pass.
template <int NA>
int dispatch(const WarpArgs &a, hipStream_t stream) {
    const int G8 = (a.C + 7) / 8, G16 = (a.C + 15) / 16;
    if (G16 > 16) return COALIGN_ERR_UNSUPPORTED;
    if (G8 <= 8) {
        if constexpr (NA <= 5) return launch<NA, 8, true, 512, 8>(a, G8, stream);
        else return launch<NA, 8, true, 512, 4>(a, G8, stream);
    }
    if (G8 <= 16) {
        if constexpr (NA <= 5) return launch<NA, 8, true, 1024, 8>(a, G8, stream);
        else return launch<NA, 8, true, 1024, 4>(a, G8, stream);
    }
    if constexpr (NA <= 3) return launch<NA, 16, true, 1024, 8>(a, G16, stream);
    else if constexpr (NA <= 5) return launch<NA, 16, true, 1024, 4>(a, G16, stream);
    else return launch<NA, 16, false, 1024, 4>(a, G16, stream);
}

}  // namespace

// normalize_pairwise_tfm (opencood/utils/transformation_utils.py:69-91) in one launch: rows {0, 1} x columns {0, 1, 3} of every 4 x 4
// matrix, [0,1] *= H / W, [1,0] *= W / H, [0,2] /= den_x, *= 2, [1,2] /= den_y, *= 2 -- float64, the reference's operation order
// (the library is built with -ffp-contract=off), so the result is bit-identical to the torch expression it replaces (~14 tiny
// element-wise launches per frame, 18-35 us each inside the pipeline).
__global__ __launch_bounds__(256) void normalize_affine_kernel(const double *__restrict__ T, double *__restrict__ out, int n, double H, double W,
                                                               double den_x, double den_y) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double *t = T + (size_t)i * 16;
    double *o = out + (size_t)i * 6;
    o[0] = t[0];
    o[1] = t[1] * H / W;
    o[2] = t[3] / den_x * 2.0;
    o[3] = t[4] * W / H;
    o[4] = t[5];
    o[5] = t[7] / den_y * 2.0;
}

extern "C" int coalign_normalize_pairwise(const double *pairwise, int n_matrices, int H, int W, double den_x, double den_y, double *out,
                                          void *stream_) {
    if (n_matrices < 0 || H < 1 || W < 1) return COALIGN_ERR_BAD_SHAPE;
    if (n_matrices == 0) return COALIGN_OK;
    if (!pairwise || !out) return COALIGN_ERR_NULL_POINTER;
    hipLaunchKernelGGL(normalize_affine_kernel, dim3((n_matrices + 255) / 256), dim3(256), 0, (hipStream_t)stream_, pairwise, out, n_matrices,
                       (double)H, (double)W, den_x, den_y);
    return coalign::check_launch();
}

extern "C" int coalign_warp_fuse(const float *x, int n_total, int C, int H, int W, const double *theta,
                                 const int32_t *group_len, int n_groups, int mode, float *out, int Ho, int Wo,
                                 void *stream_) {
    return coalign_warp_fuse_rows(x, n_total, C, H, W, theta, group_len, n_groups, nullptr, mode, out, Ho, Wo, stream_);
}

extern "C" int coalign_warp_fuse_rows(const float *x, int n_total, int C, int H, int W, const double *theta,
                                      const int32_t *group_len, int n_groups, const int32_t *rows, int mode, float *out, int Ho, int Wo,
                                      void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_total < 0 || C <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0 || n_groups < 0) return COALIGN_ERR_BAD_SHAPE;
    if (mode != COALIGN_FUSE_ATT && mode != COALIGN_FUSE_MAX && mode != COALIGN_FUSE_NONE) return COALIGN_ERR_UNSUPPORTED;
    if (n_total == 0 || n_groups == 0) return n_total == 0 && n_groups == 0 ? COALIGN_OK : COALIGN_ERR_BAD_SHAPE;
    if (!x || !theta || !group_len || !out) return COALIGN_ERR_NULL_POINTER;
    if ((size_t)C * H * W > (size_t)INT32_MAX) return COALIGN_ERR_BAD_SHAPE;
    if (C > 256) return COALIGN_ERR_UNSUPPORTED;
    long sum = 0;
    for (int b = 0; b < n_groups; ++b) {
        if (group_len[b] < 1) return COALIGN_ERR_BAD_SHAPE;
        if (group_len[b] > 8) return COALIGN_ERR_UNSUPPORTED;
        sum += group_len[b];
    }
    if (sum != n_total) return COALIGN_ERR_BAD_SHAPE;
    if (rows) {                              // a permutation inside every group
        int o = 0;
        for (int b = 0; b < n_groups; ++b) {
            unsigned seen = 0;
            for (int i = 0; i < group_len[b]; ++i) {
                const int r = rows[o + i];
                if (r < 0 || r >= group_len[b] || ((seen >> r) & 1u)) return COALIGN_ERR_BAD_SHAPE;
                seen |= 1u << r;
            }
            o += group_len[b];
        }
    }

    WarpArgs a;
    a.C = C; a.H = H; a.W = W; a.Ho = Ho; a.Wo = Wo; a.mode = mode;
    a.tiles_x = (Wo + 7) / 8;
    a.ntiles = a.tiles_x * ((Ho + 7) / 8);
    a.sqrt_dim = (float)sqrt((double)C);
    a.vec_ok = (W % 4 == 0) && (((uintptr_t)x & 15) == 0);   // 16 B patch loads need aligned rows
    int off = 0;
    for (int b = 0; b < n_groups; ++b) {
        const int n = group_len[b];
        a.n = n;
        a.x = x + (size_t)off * C * H * W;
        a.theta = theta + (size_t)off * 6;
        a.out = out + (size_t)(mode == COALIGN_FUSE_NONE ? off : b) * C * Ho * Wo;
        for (int i = 0; i < 8; ++i) a.rows[i] = (rows && i < n) ? rows[off + i] : (i < n ? i : 0);
        int rc;
        if (n == 1) rc = dispatch<1>(a, stream);
        else if (n == 2) rc = dispatch<2>(a, stream);
        else if (n == 3) rc = dispatch<3>(a, stream);
        else if (n <= 5) rc = dispatch<5>(a, stream);
        else rc = dispatch<8>(a, stream);
        if (rc) return rc;
        off += n;
    }
    return COALIGN_OK;
}
