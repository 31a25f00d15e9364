// Pose-aware affine warp + multi-agent fusion (attention / max / none), gfx950.
//
// Reference semantics (see include/coalign_amd.h): warp_affine_simple
// (opencood/models/sub_modules/torch_transformation_utils.py:322-331) feeding AttFusion / MaxFusion
// (opencood/models/fuse_modules/fusion_in_one.py:51-136).  The reference materialises the warped copy of every
// agent, permutes it to (H*W, N, C), runs two bmm + softmax over all N rows and throws N-1 of them away.
//
// This kernel fuses everything and computes the ego row only.  One workgroup owns an 8x8 tile of output pixels
// and ALL channels of ALL agents of the frame:
//     thread = (pixel = tid & 63, channel group g = tid >> 6), each thread owns CPT consecutive channels;
//     a wavefront therefore gathers a compact 8x8 footprint of one channel plane per load (bounded by a
//     ~13x13 source patch for any yaw), served by L1/L2;
//     the sampling geometry (float64 grid -> float32, 4 taps, zero padding) is computed once per (pixel, agent)
//     and reused for every channel;
//     the warped values stay in REGISTERS (KEEP variant), the only cross-wave traffic is the partial
//     <X0, Xn> dot products (G x N x 64 floats of LDS), then softmax and the weighted sum are per-thread.
// The QK^T / AV contractions are per-pixel dot products of *different* vectors for every pixel (no operand is
// shared between pixels), i.e. a batch of 1xC . CxN GEMVs with N <= 8 -- there is no tile to feed an MFMA
// with, and the f32 MFMA rate equals the f32 VALU rate on gfx950 anyway; the kernel is bound by the gather.
// Workgroup -> tile mapping is XCD-aware so horizontally adjacent tiles (which share 128-B source lines) hit
// the same L2.
#include "common.h"

namespace {

struct WarpArgs {
    const float *x;       // [NA, C, H, W] of this frame
    const double *theta;  // [NA, 2, 3]
    float *out;           // ATT/MAX: [C, Ho, Wo]; NONE: [NA, C, Ho, Wo]
    int n, C, H, W, Ho, Wo, tiles_x, ntiles, mode;
    float sqrt_dim;
};

template <int NA>
struct Taps {
    int o00[NA], dx[NA], dy[NA];  // clamped top-left offset and the +x / +y steps (0 when clamped)
    float w00[NA], w01[NA], w10[NA], w11[NA];
};

template <int NA>
__device__ __forceinline__ void setup_taps(const WarpArgs &a, int ox, int oy, bool pix_ok, Taps<NA> &t) {
    // F.affine_grid(theta_f64, align_corners=False): x_n = (2j+1)/W - 1 evaluated in float64, then .to(float32)
    const double xn = (2.0 * ox + 1.0) / a.Wo - 1.0;
    const double yn = (2.0 * oy + 1.0) / a.Ho - 1.0;
    const float half_w = (float)a.W / 2, half_h = (float)a.H / 2;
#pragma unroll
    for (int n = 0; n < NA; ++n) {
        t.o00[n] = 0; t.dx[n] = 0; t.dy[n] = 0;
        t.w00[n] = t.w01[n] = t.w10[n] = t.w11[n] = 0.f;
        if (n < a.n && pix_ok) {
            const double *th = a.theta + n * 6;
            const float gx = (float)(th[0] * xn + th[1] * yn + th[2]);
            const float gy = (float)(th[3] * xn + th[4] * yn + th[5]);
            // grid_sample un-normalisation, align_corners=False (CPU kernel form)
            const float ix = (gx + 1.f) * half_w - 0.5f;
            const float iy = (gy + 1.f) * half_h - 0.5f;
            if (ix > -1.f && ix < (float)a.W && iy > -1.f && iy < (float)a.H) {
                const float x0f = floorf(ix), y0f = floorf(iy);
                const float tx = ix - x0f, ty = iy - y0f;
                const float ex = 1.f - tx, ey = 1.f - ty;
                const int x0 = (int)x0f, y0 = (int)y0f;
                const bool vx0 = x0 >= 0, vx1 = x0 + 1 <= a.W - 1, vy0 = y0 >= 0, vy1 = y0 + 1 <= a.H - 1;
                t.w00[n] = (vx0 && vy0) ? ey * ex : 0.f;
                t.w01[n] = (vx1 && vy0) ? ey * tx : 0.f;
                t.w10[n] = (vx0 && vy1) ? ty * ex : 0.f;
                t.w11[n] = (vx1 && vy1) ? ty * tx : 0.f;
                const int xc0 = vx0 ? x0 : 0, yc0 = vy0 ? y0 : 0;
                const int xc1 = vx1 ? x0 + 1 : a.W - 1, yc1 = vy1 ? y0 + 1 : a.H - 1;
                t.o00[n] = yc0 * a.W + xc0;
                t.dx[n] = xc1 - xc0;
                t.dy[n] = (yc1 - yc0) * a.W;
            }
        }
    }
}

template <int NA>
__device__ __forceinline__ float sample(const float *__restrict__ plane, const Taps<NA> &t, int n) {
    const float *p = plane + t.o00[n];
    const float v00 = p[0], v01 = p[t.dx[n]], v10 = p[t.dy[n]], v11 = p[t.dy[n] + t.dx[n]];
    return v00 * t.w00[n] + v01 * t.w01[n] + v10 * t.w10[n] + v11 * t.w11[n];
}

template <int NA, int CPT, bool KEEP, int MAXT>
__global__ __launch_bounds__(MAXT) void warp_fuse_kernel(WarpArgs a) {
    extern __shared__ __attribute__((aligned(16))) float red[];  // [G][NA][64]
    const int tile = coalign::xcd_remap(blockIdx.x, a.ntiles);
    const int px = threadIdx.x & 63, g = threadIdx.x >> 6, G = blockDim.x >> 6;
    const int oy = (tile / a.tiles_x) * 8 + (px >> 3), ox = (tile % a.tiles_x) * 8 + (px & 7);
    const bool pix_ok = oy < a.Ho && ox < a.Wo;
    const int HW = a.H * a.W;
    const size_t HWo = (size_t)a.Ho * a.Wo;
    const int c_base = g * CPT;

    Taps<NA> t;
    setup_taps<NA>(a, ox, oy, pix_ok, t);

    if (a.mode != COALIGN_FUSE_ATT) {  // max / none: no cross-channel dependency -> stream the channels
        for (int k = 0; k < CPT; ++k) {
            const int c = c_base + k;
            if (c >= a.C) break;
            float m = -INFINITY;
#pragma unroll
            for (int n = 0; n < NA; ++n) {
                if (n < a.n) {
                    const float v = sample<NA>(a.x + ((size_t)n * a.C + c) * HW, t, n);
                    if (a.mode == COALIGN_FUSE_NONE) {
                        if (pix_ok) a.out[((size_t)n * a.C + c) * HWo + (size_t)oy * a.Wo + ox] = v;
                    } else {
                        m = fmaxf(m, v);
                    }
                }
            }
            if (a.mode == COALIGN_FUSE_MAX && pix_ok) a.out[(size_t)c * HWo + (size_t)oy * a.Wo + ox] = m;
        }
        return;
    }

    float X[KEEP ? NA : 1][KEEP ? CPT : 1];
    float part[NA];
#pragma unroll
    for (int n = 0; n < NA; ++n) part[n] = 0.f;

#pragma unroll(KEEP ? CPT : 1)
    for (int k = 0; k < CPT; ++k) {
        const int c = c_base + k;
        float v[NA];
#pragma unroll
        for (int n = 0; n < NA; ++n) {
            v[n] = 0.f;
            if (n < a.n && c < a.C) v[n] = sample<NA>(a.x + ((size_t)n * a.C + c) * HW, t, n);
            if constexpr (KEEP) X[n][k] = v[n];
        }
#pragma unroll
        for (int n = 0; n < NA; ++n) part[n] = fmaf(v[0], v[n], part[n]);
    }

#pragma unroll
    for (int n = 0; n < NA; ++n) red[(g * NA + n) * 64 + px] = part[n];
    __syncthreads();

    // scores of the ego row, softmax over the frame's agents (att_fuse.py:43-47)
    float s[NA], smax = -INFINITY;
#pragma unroll
    for (int n = 0; n < NA; ++n) {
        float acc = 0.f;
        for (int gg = 0; gg < G; ++gg) acc += red[(gg * NA + n) * 64 + px];
        s[n] = acc / a.sqrt_dim;
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

#pragma unroll(KEEP ? CPT : 1)
    for (int k = 0; k < CPT; ++k) {
        const int c = c_base + k;
        float o = 0.f;
#pragma unroll
        for (int n = 0; n < NA; ++n) {
            float v;
            if constexpr (KEEP) v = X[n][k];
            else v = (n < a.n && c < a.C) ? sample<NA>(a.x + ((size_t)n * a.C + c) * HW, t, n) : 0.f;
            o = fmaf(s[n], v, o);
        }
        if (pix_ok && c < a.C) a.out[(size_t)c * HWo + (size_t)oy * a.Wo + ox] = o;
    }
}

template <int NA, int CPT, bool KEEP, int MAXT>
int launch(const WarpArgs &a, int G, hipStream_t stream) {
    const size_t lds = (size_t)G * NA * 64 * sizeof(float);
    hipLaunchKernelGGL((warp_fuse_kernel<NA, CPT, KEEP, MAXT>), dim3(a.ntiles), dim3(G * 64), lds, stream, a);
    return coalign::check_launch();
}

// Channels-per-thread / register residency are picked from the register budget the block size leaves per
// thread (512-entry file per SIMD lane / waves per SIMD): 4 waves -> 512, 8 -> 256, 16 -> 128.
template <int NA>
int dispatch(const WarpArgs &a, hipStream_t stream) {
    const int G16 = (a.C + 15) / 16;
    if (G16 <= 4) return launch<NA, 16, true, 256>(a, G16, stream);
    if (G16 <= 8) return launch<NA, 16, true, 512>(a, G16, stream);
    if (G16 > 16) return COALIGN_ERR_UNSUPPORTED;
    if constexpr (NA <= 3) {
        return launch<NA, 16, true, 1024>(a, G16, stream);
    } else if constexpr (NA <= 5) {
        return launch<NA, 32, true, 512>(a, (a.C + 31) / 32, stream);
    } else {
        return launch<NA, 32, false, 512>(a, (a.C + 31) / 32, stream);  // two-pass: scores first, recompute for the output
    }
}

}  // namespace

extern "C" int coalign_warp_fuse(const float *x, int n_total, int C, int H, int W, const double *theta,
                                 const int32_t *group_len, int n_groups, int mode, float *out, int Ho, int Wo,
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

    WarpArgs a;
    a.C = C; a.H = H; a.W = W; a.Ho = Ho; a.Wo = Wo; a.mode = mode;
    a.tiles_x = (Wo + 7) / 8;
    a.ntiles = a.tiles_x * ((Ho + 7) / 8);
    a.sqrt_dim = (float)sqrt((double)C);
    int off = 0;
    for (int b = 0; b < n_groups; ++b) {
        const int n = group_len[b];
        a.n = n;
        a.x = x + (size_t)off * C * H * W;
        a.theta = theta + (size_t)off * 6;
        a.out = out + (size_t)(mode == COALIGN_FUSE_NONE ? off : b) * C * Ho * Wo;
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
