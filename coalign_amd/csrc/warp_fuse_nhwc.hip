// Pose-aware affine warp + multi-agent fusion on CHANNELS-LAST feature maps, all scales in one launch, gfx950.
//
// Same reference semantics as warp_fuse.hip (warp_affine_simple, torch_transformation_utils.py:322-331; AttFusion / MaxFusion,
// fusion_in_one.py:51-136; ScaledDotProductAttention, att_fuse.py:43-47) -- only the memory layout differs: x is [n][H][W][C],
// the layout the last convolution of every ResNet stage writes (conv3x3_emu.hip, LAYOUT_OUT_NHWC).
//
// Why a second kernel.  In NCHW a bilinear tap of one (agent, channel) is 4 scattered dwords, so warp_fuse.hip stages 16 x 16 source
// patches per (agent, channel) through LDS: one 16 B/lane load + ds_write + two ds_read2 + a wave-level LDS round trip per 64
// values, and it is bound by exactly that chain (profiles/round1: 0.14 of the HBM roofline over the three scales, LDS bank conflicts
// 35 %, the 25 x 88 scale only 44 workgroups).  Channels-last makes the gather trivially coalesced: a tap is C contiguous floats, a
// lane reads 2 x 16 B of it, the lanes of a pixel cover it with full 128-byte lines.  No LDS, no barriers, no patch over-fetch:
//   lane  = (pixel of the wave's group, channel slice): LPP = C / 8 lanes per pixel, 8 channels per lane as two float4 groups
//           [4 li, 4 li + 4) and [C / 2 + 4 li, ...) so that each load instruction covers contiguous 16 * LPP bytes per pixel
//   per agent: float64 grid -> float32 taps + masked weights once per lane, 8 x 16 B loads (4 taps x 2 groups) issued one agent
//           ahead of their use, 32 FMAs; the warped values of all agents stay in registers (X[NA][8])
//   scores: <X0, Xn> partials reduced over the LPP lanes of the pixel with DPP butterflies (quad_perm, row_half_mirror, row_mirror;
//           one ds_swizzle for C = 256), softmax + weighted sum per lane, 2 x 16 B stores (channels-last output)
// One launch serves the three scales of a frame: a workgroup (4 waves) owns a (512 / C) x 4 pixel tile of one scale -- 8 x 4 at
// C = 64, 4 x 4 at 128, 2 x 4 at 256 -- so every wave of every scale does the same amount of work (N agents x 8 loads of 1 KB); tile
// order is XCD-aware inside each scale.  1980 equal workgroups at the OPV2V sizes instead of 1100 + 154 + 44 in three launches.
#include "common.h"

namespace {

constexpr int kMaxScales = 3;

struct ScaleArgs {
    const float *x;       // [n, H, W, C] of this frame
    float *out;           // ATT / MAX: [Ho, Wo, C]; NONE: [n, Ho, Wo, C]
    int C, H, W, Ho, Wo, tiles_x, ntiles, first_block, n_blocks;   // n_blocks = ntiles rounded up to a multiple of 8 (XCD remap)
    float sqrt_dim, inv_sqrt_dim;   // inv_sqrt_dim is used only where sqrt_dim is a power of two (C = 64, 256)
};

struct FuseArgs {
    ScaleArgs s[kMaxScales];
    const double *theta;  // [n, 2, 3]
    int n_scales, n, mode;
    int rows[8];
};

template <int CTRL>
__device__ __forceinline__ float dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}

// sum over the LPP lanes that share a pixel (aligned groups of 8 / 16 / 32 lanes); every lane of the group gets the same value
template <int LPP>
__device__ __forceinline__ float group_sum(float v) {
    v += dpp<0xB1>(v);                         // quad_perm [1, 0, 3, 2]: xor 1
    v += dpp<0x4E>(v);                         // quad_perm [2, 3, 0, 1]: xor 2
    v += dpp<0x141>(v);                        // row_half_mirror: lane i <- 7 - i of its 8-lane group (both quads now hold their sums)
    if constexpr (LPP >= 16) v += dpp<0x140>(v);   // row_mirror: lane i <- 15 - i
    if constexpr (LPP >= 32) v += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), (16 << 10) | 0x1f));
    return v;
}

struct Taps {
    int o00, o01, o10, o11;                // element offsets of the four (clamped) taps inside the agent's plane (channel 0 of the pixel)
    float w00, w01, w10, w11;              // masked bilinear weights (zero padding)
};

// grid_sample geometry of output pixel (ox, oy) in agent n's plane, reference arithmetic (identical to warp_fuse.hip):
// F.affine_grid on a float64 theta -> .to(float32) -> (g + 1) * (size / 2) - 0.5 -> floor / floor + 1 taps, masked weights
__device__ __forceinline__ Taps make_taps(const ScaleArgs &a, const double *theta, int n, int ox, int oy) {
    const double xn = (2.0 * ox + 1.0) / a.Wo - 1.0;
    const double yn = (2.0 * oy + 1.0) / a.Ho - 1.0;
    const double *th = theta + n * 6;
    const float gx = (float)(th[0] * xn + th[1] * yn + th[2]);
    const float gy = (float)(th[3] * xn + th[4] * yn + th[5]);
    const float ix = (gx + 1.f) * ((float)a.W / 2) - 0.5f;
    const float iy = (gy + 1.f) * ((float)a.H / 2) - 0.5f;
    Taps t;
    t.w00 = t.w01 = t.w10 = t.w11 = 0.f;
    int x0 = 0, y0 = 0;
    if (ix > -1.f && ix < (float)a.W && iy > -1.f && iy < (float)a.H) {
        const float x0f = floorf(ix), y0f = floorf(iy);
        const float tx = ix - x0f, ty = iy - y0f, ex = 1.f - tx, ey = 1.f - ty;
        x0 = (int)x0f; y0 = (int)y0f;
        const bool vx0 = x0 >= 0, vx1 = x0 + 1 <= a.W - 1, vy0 = y0 >= 0, vy1 = y0 + 1 <= a.H - 1;
        t.w00 = (vx0 && vy0) ? ey * ex : 0.f;
        t.w01 = (vx1 && vy0) ? ey * tx : 0.f;
        t.w10 = (vx0 && vy1) ? ty * ex : 0.f;
        t.w11 = (vx1 && vy1) ? ty * tx : 0.f;
    }
    const int xc0 = min(max(x0, 0), a.W - 1), xc1 = min(max(x0 + 1, 0), a.W - 1);
    const int yc0 = min(max(y0, 0), a.H - 1), yc1 = min(max(y0 + 1, 0), a.H - 1);
    t.o00 = (yc0 * a.W + xc0) * a.C;       // (C * H * W <= INT32_MAX is checked by the entry point)
    t.o01 = (yc0 * a.W + xc1) * a.C;
    t.o10 = (yc1 * a.W + xc0) * a.C;
    t.o11 = (yc1 * a.W + xc1) * a.C;
    return t;
}

// Round 4: the LPP >= 8 lanes of a pixel used to compute the SAME taps for every agent (float64 grid + clamps: ~60 VALU instructions per agent and lane, a
// third of the kernel's instructions, and the kernel is half issue-bound).  Now lane li of a pixel computes the taps of agent li once (N <= 8 <= LPP), and
// agent n's taps reach the pixel's other lanes through the LDS crossbar: ds_swizzle in bit-mask mode, new lane = (lane & ~(LPP - 1)) | n inside each half.
template <int LPP, int N>
__device__ __forceinline__ int from_lane(int v) {
    return __builtin_amdgcn_ds_swizzle(v, ((~(LPP - 1)) & 0x1f) | (N << 5));
}
template <int LPP>
__device__ __forceinline__ int from_lane_n(int v, int n) {          // n is a constant after unrolling: the switch folds
    switch (n) {
        case 0: return from_lane<LPP, 0>(v);
        case 1: return from_lane<LPP, 1>(v);
        case 2: return from_lane<LPP, 2>(v);
        case 3: return from_lane<LPP, 3>(v);
        case 4: return from_lane<LPP, 4>(v);
        case 5: return from_lane<LPP, 5>(v);
        case 6: return from_lane<LPP, 6>(v);
        default: return from_lane<LPP, 7>(v);
    }
}
template <int LPP>
__device__ __forceinline__ Taps taps_of_agent(const Taps &mine, int n) {
    Taps t;
    t.o00 = from_lane_n<LPP>(mine.o00, n); t.o01 = from_lane_n<LPP>(mine.o01, n); t.o10 = from_lane_n<LPP>(mine.o10, n); t.o11 = from_lane_n<LPP>(mine.o11, n);
    t.w00 = __builtin_bit_cast(float, from_lane_n<LPP>(__builtin_bit_cast(int, mine.w00), n));
    t.w01 = __builtin_bit_cast(float, from_lane_n<LPP>(__builtin_bit_cast(int, mine.w01), n));
    t.w10 = __builtin_bit_cast(float, from_lane_n<LPP>(__builtin_bit_cast(int, mine.w10), n));
    t.w11 = __builtin_bit_cast(float, from_lane_n<LPP>(__builtin_bit_cast(int, mine.w11), n));
    return t;
}

// base: channel c_lo of pixel (0, 0) of the agent's plane (wave-uniform pointer + this lane's channel slice); hi4 = C / 8: float4 index of the high channel group
__device__ __forceinline__ void issue(const Taps &t, const float *base, int hi4, float4 (&v)[8]) {
    const float4 *p00 = reinterpret_cast<const float4 *>(base + t.o00), *p01 = reinterpret_cast<const float4 *>(base + t.o01);
    const float4 *p10 = reinterpret_cast<const float4 *>(base + t.o10), *p11 = reinterpret_cast<const float4 *>(base + t.o11);
    v[0] = p00[0]; v[1] = p01[0]; v[2] = p10[0]; v[3] = p11[0];
    v[4] = p00[hi4]; v[5] = p01[hi4]; v[6] = p10[hi4]; v[7] = p11[hi4];
}

__device__ __forceinline__ void blend(const Taps &t, const float4 (&v)[8], float (&X)[8]) {
    // v00 * w00 + v01 * w01 + v10 * w10 + v11 * w11, evaluated left to right, every product and sum rounded (-ffp-contract=off)
#define COALIGN_TAP(j, f) X[j] = v[(j / 4) * 4 + 0].f * t.w00 + v[(j / 4) * 4 + 1].f * t.w01 + v[(j / 4) * 4 + 2].f * t.w10 + v[(j / 4) * 4 + 3].f * t.w11
    COALIGN_TAP(0, x); COALIGN_TAP(1, y); COALIGN_TAP(2, z); COALIGN_TAP(3, w);
    COALIGN_TAP(4, x); COALIGN_TAP(5, y); COALIGN_TAP(6, z); COALIGN_TAP(7, w);
#undef COALIGN_TAP
}

template <int NA, int LPP>
__device__ __forceinline__ void fuse_tile(const FuseArgs &f, const ScaleArgs &a, int tile, int wave, int lane) {
    constexpr int PPW = 64 / LPP;                 // pixels per wave: a workgroup's tile is PPW x 4 pixels, one row per wave, so every
    const int li = lane % LPP, pi = lane / LPP;   // wave of every scale does the same work (N agents x 8 loads): no heavy tiles
    const int c_lo = 4 * li, hi4 = a.C / 8;
    const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
    const int oy = ty * 4 + wave;
    if (oy >= a.Ho) return;                        // wave-uniform
    {
        const int ox_raw = tx * PPW + pi;
        const bool pix_ok = ox_raw < a.Wo;
        const int ox = pix_ok ? ox_raw : a.Wo - 1;
        float X[NA][8];
        const size_t plane = (size_t)a.H * a.W * a.C;
        const Taps mine = make_taps(a, f.theta, min(li, f.n - 1), ox, oy);           // lane li of the pixel: agent li's taps
        Taps cur = taps_of_agent<LPP>(mine, 0), nxt;
        float4 vc[8], vn[8];
        issue(cur, a.x + f.rows[0] * plane + c_lo, hi4, vc);
#pragma unroll
        for (int n = 0; n < NA; ++n) {
            if (n < f.n) {
                if (n + 1 < NA && n + 1 < f.n) {   // the next agent's taps are in flight while this one is blended
                    nxt = taps_of_agent<LPP>(mine, n + 1);
                    issue(nxt, a.x + f.rows[(n + 1) % 8] * plane + c_lo, hi4, vn);
                }
                blend(cur, vc, X[n]);
                if (n + 1 < NA && n + 1 < f.n) {
                    cur = nxt;
#pragma unroll
                    for (int j = 0; j < 8; ++j) vc[j] = vn[j];
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) X[n][j] = 0.f;
            }
        }
        const size_t opix = ((size_t)oy * a.Wo + ox) * a.C + c_lo;
        if (f.mode == COALIGN_FUSE_ATT) {
            float s[NA];
            float smax = -INFINITY;
#pragma unroll
            for (int n = 0; n < NA; ++n) {
                float p = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) p = fmaf(X[0][j], X[n][j], p);
                // score / np.sqrt(C) (att_fuse.py:44).  C = 64 and C = 256: sqrt(C) is a power of two, the division is exact and equals the product with
                // 1 / sqrt(C) bit for bit (one instruction instead of the ten of a correctly rounded division); C = 128: the division
                s[n] = (LPP == 16) ? group_sum<LPP>(p) / a.sqrt_dim : group_sum<LPP>(p) * a.inv_sqrt_dim;
                if (n < f.n) smax = fmaxf(smax, s[n]);
            }
            // Round 6: softmax with the hardware exponential (v_exp_f32 on (s - max) * log2 e: ~1 ulp, like the vectorised exp behind F.softmax on the CPU) and ONE
            // division for the pixel instead of one per agent -- a fifth of the kernel's vector instructions were the library expf and the N divisions
            // (weights within 2 ulp of exp(s - max) / sum: 2e-7 of a fused value, against the 1e-4 the parity tests hold).
            float den = 0.f;
#pragma unroll
            for (int n = 0; n < NA; ++n) {
                s[n] = (n < f.n) ? __builtin_amdgcn_exp2f((s[n] - smax) * 1.44269504088896341f) : 0.f;
                den += s[n];
            }
            const float inv_den = 1.0f / den;
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = 0.f;
#pragma unroll
            for (int n = 0; n < NA; ++n) {
                const float sn = s[n] * inv_den;
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = fmaf(sn, X[n][j], o[j]);
            }
            if (pix_ok) {
                coalign::store_stream(reinterpret_cast<float4 *>(a.out + opix), make_float4(o[0], o[1], o[2], o[3]));
                coalign::store_stream(reinterpret_cast<float4 *>(a.out + opix + a.C / 2), make_float4(o[4], o[5], o[6], o[7]));
            }
        } else if (f.mode == COALIGN_FUSE_MAX) {
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = -INFINITY;
#pragma unroll
            for (int n = 0; n < NA; ++n)
                if (n < f.n) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] = fmaxf(o[j], X[n][j]);
                }
            if (pix_ok) {
                coalign::store_stream(reinterpret_cast<float4 *>(a.out + opix), make_float4(o[0], o[1], o[2], o[3]));
                coalign::store_stream(reinterpret_cast<float4 *>(a.out + opix + a.C / 2), make_float4(o[4], o[5], o[6], o[7]));
            }
        } else {
            const size_t per_agent = (size_t)a.Ho * a.Wo * a.C;
#pragma unroll
            for (int n = 0; n < NA; ++n)
                if (n < f.n && pix_ok) {
                    *reinterpret_cast<float4 *>(a.out + n * per_agent + opix) = make_float4(X[n][0], X[n][1], X[n][2], X[n][3]);
                    *reinterpret_cast<float4 *>(a.out + n * per_agent + opix + a.C / 2) = make_float4(X[n][4], X[n][5], X[n][6], X[n][7]);
                }
        }
    }
}

template <int NA>
__global__ __launch_bounds__(256) void warp_fuse_nhwc_kernel(const FuseArgs f) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int bid = blockIdx.x;
#pragma unroll
    for (int k = 0; k < kMaxScales; ++k) {
        if (k < f.n_scales && bid >= f.s[k].first_block && bid < f.s[k].first_block + f.s[k].n_blocks) {
            const ScaleArgs &a = f.s[k];
            const int tile = coalign::xcd_remap(bid - a.first_block, a.n_blocks);   // first_block and n_blocks are multiples of 8
            if (tile >= a.ntiles) return;
            const int lpp = a.C / 8;
            if (lpp == 8) fuse_tile<NA, 8>(f, a, tile, wave, lane);
            else if (lpp == 16) fuse_tile<NA, 16>(f, a, tile, wave, lane);
            else fuse_tile<NA, 32>(f, a, tile, wave, lane);
            return;
        }
    }
}

}  // namespace

extern "C" int coalign_warp_fuse_nhwc(int n_scales, const float *const *x, const int32_t *C, const int32_t *H, const int32_t *W,
                                      float *const *out, const int32_t *Ho, const int32_t *Wo, int n, const double *theta,
                                      const int32_t *rows, int mode, void *stream_) {
    using namespace coalign;
    hipStream_t stream = (hipStream_t)stream_;
    if (n_scales < 1 || n_scales > kMaxScales || n < 1) return COALIGN_ERR_BAD_SHAPE;
    if (n > 8) return COALIGN_ERR_UNSUPPORTED;
    if (mode != COALIGN_FUSE_ATT && mode != COALIGN_FUSE_MAX && mode != COALIGN_FUSE_NONE) return COALIGN_ERR_UNSUPPORTED;
    if (!x || !C || !H || !W || !out || !Ho || !Wo || !theta) return COALIGN_ERR_NULL_POINTER;
    FuseArgs f;
    f.theta = theta; f.n = n; f.mode = mode; f.n_scales = n_scales;
    unsigned seen = 0;
    for (int i = 0; i < 8; ++i) {
        const int r = (rows && i < n) ? rows[i] : (i < n ? i : 0);
        if (i < n) {
            if (r < 0 || r >= n || ((seen >> r) & 1u)) return COALIGN_ERR_BAD_SHAPE;
            seen |= 1u << r;
        }
        f.rows[i] = r;
    }
    // heaviest tiles first: order the scales by channel count, descending
    int order[kMaxScales] = {0, 1, 2};
    for (int i = 0; i < n_scales; ++i)
        for (int j = i + 1; j < n_scales; ++j)
            if (C[order[j]] > C[order[i]]) { const int t = order[i]; order[i] = order[j]; order[j] = t; }
    int next_block = 0;
    for (int k = 0; k < n_scales; ++k) {
        const int i = order[k];
        if (!x[i] || !out[i]) return COALIGN_ERR_NULL_POINTER;
        if (C[i] < 1 || H[i] < 1 || W[i] < 1 || Ho[i] < 1 || Wo[i] < 1) return COALIGN_ERR_BAD_SHAPE;
        if (C[i] != 64 && C[i] != 128 && C[i] != 256) return COALIGN_ERR_UNSUPPORTED;
        if ((size_t)C[i] * H[i] * W[i] > (size_t)INT32_MAX) return COALIGN_ERR_BAD_SHAPE;
        if ((reinterpret_cast<uintptr_t>(x[i]) | reinterpret_cast<uintptr_t>(out[i])) & 15) return COALIGN_ERR_UNSUPPORTED;
        ScaleArgs &a = f.s[k];
        a.x = x[i]; a.out = out[i]; a.C = C[i]; a.H = H[i]; a.W = W[i]; a.Ho = Ho[i]; a.Wo = Wo[i];
        a.tiles_x = (Wo[i] + (512 / C[i]) - 1) / (512 / C[i]);       // 64 lanes / (C / 8 lanes per pixel) pixels per wave
        a.ntiles = a.tiles_x * ((Ho[i] + 3) / 4);
        a.n_blocks = (a.ntiles + 7) / 8 * 8;
        a.first_block = next_block;
        a.sqrt_dim = (float)sqrt((double)C[i]);
        a.inv_sqrt_dim = 1.0f / a.sqrt_dim;
        next_block += a.n_blocks;
    }
    for (int k = n_scales; k < kMaxScales; ++k) f.s[k] = f.s[0];
    const dim3 grid(next_block), block(256);
    if (n == 1) hipLaunchKernelGGL(warp_fuse_nhwc_kernel<1>, grid, block, 0, stream, f);
    else if (n == 2) hipLaunchKernelGGL(warp_fuse_nhwc_kernel<2>, grid, block, 0, stream, f);
    else if (n == 3) hipLaunchKernelGGL(warp_fuse_nhwc_kernel<3>, grid, block, 0, stream, f);
    else if (n <= 5) hipLaunchKernelGGL(warp_fuse_nhwc_kernel<5>, grid, block, 0, stream, f);
    else hipLaunchKernelGGL(warp_fuse_nhwc_kernel<8>, grid, block, 0, stream, f);
    return check_launch();
}
