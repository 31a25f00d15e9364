// Pointwise (per-pixel GEMM) layers of the BEV backbone on the fp32 matrix cores, NCHW, gfx950:
//   * the up-sampling heads  ConvTranspose2d(kernel = stride = k in {1, 2, 4}) + BatchNorm + ReLU  whose outputs are
//     concatenated along the channels (opencood/models/sub_modules/base_bev_backbone_resnet.py:47-87, 121-138): every input
//     pixel produces a k x k patch of every output channel and nothing overlaps, so the layer is one GEMM
//     D[(co, ky, kx), pixel] = sum_ci W[ci, co, ky, kx] * X[ci, pixel]; the result is written straight into its channel slice
//     of the concatenated tensor (no torch.cat pass);
//   * the 1 x 1 / stride-2 down-sampling convolution + BatchNorm on the skip path of the first block of a ResNet stage
//     (opencood/models/sub_modules/resblock.py:53-69, 165-174): the same GEMM reading every second pixel of every second row.
// MIOpen serves these through NHWC implicit-GEMM / rocBLAS kernels wrapped in NCHW<->NHWC transposes, a col2im pass and a
// separate bias / ReLU pass (profiles/round1: ~0.4 ms per frame for 5.3 GFLOP); here each is a single launch.
//
// A workgroup (4 wavefronts) owns 32 GEMM pixels and up to 256 GEMM rows (m = co * k * k + ky * k + kx) -- or, for layers with <= 128 / <= 64
// rows, 64 / 128 pixels and 128 / 64 rows, so that no wavefront idles --; it stages the X tile
// [Cin x 32 pixels] in LDS once, each wavefront keeps two 32 x 32 accumulator tiles and streams its weight rows from L2
// (the weight layout [Cin][M] of ConvTranspose2d is already the A operand: 32 consecutive m for one input channel).
#include "common.h"
#include <cstdlib>

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int kMaxCin = 256;

struct PwArgs {
    const float *__restrict__ x, *__restrict__ w, *__restrict__ bias;
    float *__restrict__ y;
    int N, Cin, Hin, Win, in_stride, Hp, Wp, M, up, Cout, Ctot, c_off, relu;
    int in_nhwc;          // x is [N, Hin, Win, Cin] (channels-last: the fused maps / stage outputs of the NHWC route)
    int pb;               // pixel blocks of 32 per workgroup: 1, 2 or 4 (see the kernel)
    int out_nhwc;         // y is [N, Hp, Wp, Ctot] (up = 1 only): the skip convolution feeding a channels-last residual add
    // round 4, sparse canvas (csrc/pillar_sparse.hip; in_nhwc only): x = feature rows [M][Cin], pixel (n, y, x) = row (stamp & 0xffffffff) if
    // stamps[(n * Hin + y) * Win + x] >> 32 == *tag_ptr, else zero
    const unsigned long long *stamps;
    const int *tag_ptr;
    unsigned sparse_rows;  // round 6: rows behind x; a stamp naming a row beyond them reads as empty
    // round 5: y is an SP map (csrc/conv3x3_sp.hip: [N][Ctot / 16][4 planes][Ho][Wo][8 x fp16]) -- the up-sampling heads write the input of the shrink header's
    // first convolution already split; range_flag as there (bit 0: a value beyond the pair's range)
    int out_sp;
    int *range_flag;
};

// Epilogue shared by both kernels: accumulator r of lane l is GEMM row 8 * (r / 4) + 4 * (l / 32) + r % 4 of its 32-row tile, pixel l % 32
template <int UP>
__device__ __forceinline__ void pw_store(const PwArgs &a, int n, int m0, bool second, int px, int half, const floatx16 &acc0, const floatx16 &acc1) {
    const int pixels = a.Hp * a.Wp;
    if (px >= pixels) return;
    const int hp = px / a.Wp, wp = px - hp * a.Wp;
    const int Ho = a.Hp * UP, Wo = a.Wp * UP;
    const size_t out_plane = (size_t)Ho * Wo;
    float *yout = a.y + ((size_t)n * a.Ctot + a.c_off) * out_plane;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        if (t == 1 && !second) break;
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const int m = m0 + t * 32 + 8 * r4 + 4 * half;            // rows m .. m + 3 are accumulators 4 * r4 .. 4 * r4 + 3
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = t == 0 ? acc0[4 * r4 + j] : acc1[4 * r4 + j];
            if (UP == 4) {                 // m % 16 = ky * 4 + kx: the four rows are kx = 0..3 of one (co, ky): one 16-byte store
                const int co = m >> 4, ky = (m >> 2) & 3;
                const float bb = a.bias[co];
                float4 o;
                o.x = v[0] + bb; o.y = v[1] + bb; o.z = v[2] + bb; o.w = v[3] + bb;
                if (a.relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
                *reinterpret_cast<float4 *>(yout + (size_t)co * out_plane + (size_t)(hp * 4 + ky) * Wo + wp * 4) = o;
            } else if (UP == 2) {          // m % 4 = ky * 2 + kx: the four rows are the 2 x 2 patch of one co: two 8-byte stores
                const int co = m >> 2;
                const float bb = a.bias[co];
#pragma unroll
                for (int ky = 0; ky < 2; ++ky) {
                    float2 o;
                    o.x = v[ky * 2] + bb; o.y = v[ky * 2 + 1] + bb;
                    if (a.relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); }
                    *reinterpret_cast<float2 *>(yout + (size_t)co * out_plane + (size_t)(hp * 2 + ky) * Wo + wp * 2) = o;
                }
            } else if (a.out_nhwc) {       // channels-last output: four consecutive channels of one pixel = one 16-byte store
                if (m + 3 < a.Cout) {
                    const float4 bb = *reinterpret_cast<const float4 *>(a.bias + m);
                    float4 o;
                    o.x = v[0] + bb.x; o.y = v[1] + bb.y; o.z = v[2] + bb.z; o.w = v[3] + bb.w;
                    if (a.relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
                    *reinterpret_cast<float4 *>(a.y + ((size_t)n * pixels + px) * a.Ctot + a.c_off + m) = o;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (m + j < a.Cout) {
                            const float o = v[j] + a.bias[m + j];
                            a.y[((size_t)n * pixels + px) * a.Ctot + a.c_off + m + j] = a.relu ? fmaxf(o, 0.f) : o;
                        }
                }
            } else {                       // four consecutive output channels
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (m + j < a.Cout) {
                        const float o = v[j] + a.bias[m + j];
                        yout[(size_t)(m + j) * out_plane + (size_t)hp * Wo + wp] = a.relu ? fmaxf(o, 0.f) : o;
                    }
                }
            }
        }
    }
}

template <int UP>
__global__ __launch_bounds__(256) void pointwise_kernel(const PwArgs a) {
    __shared__ float xt[kMaxCin * 32];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, half = lane >> 5, p = lane & 31;
    const int pixels = a.Hp * a.Wp;
    // PB pixel blocks of 32 per workgroup: a layer with few GEMM rows (M <= 64: one wavefront's two tiles; <= 128: two wavefronts) would
    // leave three (two) of the four wavefronts idle after the staging -- they take further pixel blocks instead (a.pb in {1, 2, 4},
    // Cin * 32 * pb floats of LDS)
    const int PB = a.pb, WPB = 4 / PB;                    // wavefronts per pixel block = row tiles of 64 per workgroup
    const int TP = 32 * PB;                               // pixels per workgroup
    const int p0 = blockIdx.x * TP, n = blockIdx.z;
    const int m_base = blockIdx.y * (64 * WPB);
    const size_t in_plane = (size_t)a.Hin * a.Win;
    const float *xin = a.stamps ? a.x : a.x + (size_t)n * a.Cin * in_plane;
    // stage X[ci][TP pixels]: thread t loads pixel t % TP of channels t / TP, t / TP + 256 / TP, ...
    {
        const int pl = tid & (TP - 1), c0 = tid / TP, cstep = 256 / TP;
        const int px = p0 + pl;
        const bool ok0 = px < pixels;
        const int hp = ok0 ? px / a.Wp : 0, wp = ok0 ? px - hp * a.Wp : 0;
        size_t off = (size_t)(hp * a.in_stride) * a.Win + (size_t)wp * a.in_stride;
        bool ok = ok0;
        if (a.stamps) {        // sparse canvas: the pixel's feature row, or nothing
            const unsigned long long st = a.stamps[(size_t)n * in_plane + off];
            ok = ok0 && (unsigned)(st >> 32) == (unsigned)*a.tag_ptr && (unsigned)st < a.sparse_rows;
            off = ok ? (size_t)(unsigned)st : 0;
        }
        if (a.in_nhwc) {       // a pixel's channels are contiguous: 16 B per lane, thread t takes channel quads c0, c0 + cstep, ...
            const float4 *xp = reinterpret_cast<const float4 *>(xin + off * a.Cin);
            for (int c4 = c0; c4 < (a.Cin >> 2); c4 += cstep) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ok) v = xp[c4];
                xt[(4 * c4) * TP + pl] = v.x; xt[(4 * c4 + 1) * TP + pl] = v.y;
                xt[(4 * c4 + 2) * TP + pl] = v.z; xt[(4 * c4 + 3) * TP + pl] = v.w;
            }
        } else {
            for (int ci = c0; ci < a.Cin; ci += cstep) {
                const float v = xin[(size_t)ci * in_plane + off];
                xt[ci * TP + pl] = ok ? v : 0.f;
            }
        }
    }
    __syncthreads();
    // this wavefront's pixel block and its two m-tiles
    const int pbk = wave / WPB, rt = wave - pbk * WPB;
    const int m0 = m_base + rt * 64;
    const bool active = m0 < a.M;
    if (!active && !a.out_sp) return;                     // (the SP epilogue below has workgroup barriers: idle wavefronts stay for them)
    const bool second = m0 + 32 < a.M;
    floatx16 acc0 = {0}, acc1 = {0};
    const float *w0 = a.w + m0 + p;
    const float *xb = xt + pbk * 32 + p;
#pragma unroll 8
    for (int k0 = 0; k0 < a.Cin; k0 += 2) {
        const float b = xb[(k0 + half) * TP];
        const float a0 = w0[(size_t)(k0 + half) * a.M];
        const float a1 = second ? w0[(size_t)(k0 + half) * a.M + 32] : 0.f;
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b, acc1, 0, 0, 0);
    }
    pw_store<UP>(a, n, m0, second, p0 + pbk * 32 + p, half, acc0, acc1);
}

// SP-map epilogue (round 5).  The 8 consecutive channels of one 16-byte group of an SP map are spread over the accumulators of both lane halves (UP = 1, 2)
// or of two wavefronts (UP = 4: a wavefront's 64 GEMM rows are 4 output channels): the workgroup parks its [rows][32 pixels] float tile in LDS (the X tile is
// dead by then) and every thread then assembles whole groups: item = (pixel block, 8-channel group, ky, kx, pixel), kx and the pixel fastest, so that a
// wavefront's 16-byte stores run along output rows.  Bias, ReLU and the sp16 split (common.h) happen on the way out: the same values, bit for bit, as
// coalign_sp_pack of the float32 result.
template <int UP>
__device__ __forceinline__ void pw_store_sp(const PwArgs &a, float *stage, int n, int m_base, int p0, int RW, int PB, int tid, int pbk, int rt, int lane, bool active, bool second,
                                            const floatx16 &acc0, const floatx16 &acc1) {
    constexpr int UU = UP * UP;
    const int half = lane >> 5, p = lane & 31;
    __syncthreads();                                       // every wavefront has read its last X operands
    if (active) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (t == 1 && !second) break;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rt * 64 + t * 32 + 8 * (r / 4) + 4 * half + r % 4;
                stage[(pbk * RW + row) * 33 + p] = t == 0 ? acc0[r] : acc1[r];
            }
        }
    }
    __syncthreads();
    const int pixels = a.Hp * a.Wp, Ho = a.Hp * UP, Wo = a.Wp * UP;
    const size_t HWo = (size_t)Ho * Wo;
    const int c8n = RW / (8 * UU);                         // 8-channel groups of the workgroup's rows
    const int items = PB * c8n * UU * 32;
    uint4 *ysp = reinterpret_cast<uint4 *>(a.y);
    bool big = false;
    for (int it = tid; it < items; it += 256) {
        const int kx = it % UP, pp = (it / UP) & 31;
        int rest = it / (32 * UP);
        const int ky = rest % UP;
        rest /= UP;
        const int c8 = rest % c8n, pb = rest / c8n;
        const int co0 = m_base / UU + 8 * c8;              // first of the 8 output channels (layer-local)
        const int px = p0 + pb * 32 + pp;
        if (co0 >= a.Cout || px >= pixels) continue;
        const int hp = px / a.Wp, wp = px - hp * a.Wp;
        const float4 b0 = *reinterpret_cast<const float4 *>(a.bias + co0), b1 = *reinterpret_cast<const float4 *>(a.bias + co0 + 4);
        const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            v[j] = stage[(pb * RW + (8 * c8 + j) * UU + ky * UP + kx) * 33 + pp] + bb[j];
            if (a.relu) v[j] = fmaxf(v[j], 0.f);
            big = big || fabsf(v[j]) > 65504.f;
        }
        unsigned h[4], l[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) coalign::sp16_split2(v[2 * k], v[2 * k + 1], h[k], l[k]);
        const int cg = a.c_off + co0;                      // channel inside the concatenated map
        const size_t idx = ((size_t)(n * (a.Ctot / 16) + cg / 16) * 4 + ((cg >> 3) & 1) * 2) * HWo + (size_t)(hp * UP + ky) * Wo + wp * UP + kx;
        ysp[idx] = uint4{h[0], h[1], h[2], h[3]};
        ysp[idx + HWo] = uint4{l[0], l[1], l[2], l[3]};
    }
    if (a.range_flag && big) atomicOr(a.range_flag, 1);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The same layers on the bf16 matrix cores by error-free 3-way operand splitting (round 3), as conv3x3_emu.hip does for the 3 x 3 layers:
// w = w_h + w_m + w_l, x = x_h + x_m + x_l (bf16 terms, round-to-nearest-even), the six products down to 2^-16 of the leading one
// (hl, mm, lh, hm, mh, hh, smallest first) accumulated in fp32 by v_mfma_f32_32x32x16_bf16.  The fp32 instruction above moves K = 2 per
// issue and needs one weight word per lane for it; here one issue covers K = 16, the pixel tile is split ONCE while it is staged
// ([term][8-channel group][pixel] x 16 B in LDS: a B operand is one ds_read_b128), and the weights come pre-split in operand order
// (image below), one 16-byte load per lane, tile, step and term, a step ahead of the matrix instructions that use them.
// Weight image (ops.pack_pointwise_emu_weight): uint4 [M / 32 row tiles][Cin / 16 steps][3 terms][64 lanes]; lane l of (tile, step, term) holds
// term `term` of W[k = 16 step + 8 (l / 32) + 0..7][m = 32 tile + l % 32] as 8 bf16.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split8(const float (&v)[8], bf16x8 (&out)[3]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const __bf16 h = (__bf16)v[i];
        const float r = v[i] - (float)h;
        const __bf16 m = (__bf16)r;
        out[0][i] = h; out[1][i] = m; out[2][i] = (__bf16)(r - (float)m);
    }
}

template <int UP>
__device__ __forceinline__ void pointwise_emu_body(const PwArgs &a, uint4 *xs, int bx, int by, int bz) {      // xs: [3][Cin / 8][TP] (dynamic LDS)
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, half = lane >> 5, p = lane & 31;
    const int pixels = a.Hp * a.Wp;
    const int PB = a.pb, WPB = 4 / PB, TP = 32 * PB;
    const int p0 = bx * TP, n = bz;
    const int m_base = by * (64 * WPB);
    const size_t in_plane = (size_t)a.Hin * a.Win;
    const float *xin = a.stamps ? a.x : a.x + (size_t)n * a.Cin * in_plane;
    const int G = a.Cin >> 3;
    {
        const int pl = tid & (TP - 1), g0 = tid / TP, gstep = 256 / TP;
        const int px = p0 + pl;
        const bool ok0 = px < pixels;
        const int hp = ok0 ? px / a.Wp : 0, wp = ok0 ? px - hp * a.Wp : 0;
        size_t off = (size_t)(hp * a.in_stride) * a.Win + (size_t)wp * a.in_stride;
        bool ok = ok0;
        if (a.stamps) {        // sparse canvas: the pixel's feature row, or nothing
            const unsigned long long st = a.stamps[(size_t)n * in_plane + off];
            ok = ok0 && (unsigned)(st >> 32) == (unsigned)*a.tag_ptr && (unsigned)st < a.sparse_rows;
            off = ok ? (size_t)(unsigned)st : 0;
        }
        for (int g = g0; g < G; g += gstep) {
            float u[8];
            if (a.in_nhwc) {
                const float4 *xp = reinterpret_cast<const float4 *>(xin + off * a.Cin) + 2 * g;
                const float4 q0 = xp[0], q1 = xp[1];
                u[0] = q0.x; u[1] = q0.y; u[2] = q0.z; u[3] = q0.w; u[4] = q1.x; u[5] = q1.y; u[6] = q1.z; u[7] = q1.w;
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) u[j] = xin[(size_t)(8 * g + j) * in_plane + off];
            }
            if (!ok) {
#pragma unroll
                for (int j = 0; j < 8; ++j) u[j] = 0.f;
            }
            bf16x8 o[3];
            split8(u, o);
#pragma unroll
            for (int t = 0; t < 3; ++t) xs[(t * G + g) * TP + pl] = __builtin_bit_cast(uint4, o[t]);
        }
    }
    __syncthreads();
    const int pbk = wave / WPB, rt = wave - pbk * WPB;
    const int m0 = m_base + rt * 64;
    const bool active = m0 < a.M;
    if (!active && !a.out_sp) return;                     // (the SP epilogue below has workgroup barriers: idle wavefronts stay for them)
    const bool second = m0 + 32 < a.M;
    const int S = a.Cin >> 4;
    const uint4 *w0 = reinterpret_cast<const uint4 *>(a.w) + (size_t)((active ? m0 : 0) >> 5) * S * 192 + lane;      // 3 terms x 64 lanes per (tile, step)
    const uint4 *w1 = w0 + (second ? (size_t)S * 192 : 0);
    const uint4 *xb = xs + pbk * 32 + p;
    floatx16 acc0 = {0}, acc1 = {0};
    constexpr int wi[6] = {0, 1, 2, 0, 1, 0};             // weight term of product i   (0 = h, 1 = m, 2 = l): hl, mm, lh, hm, mh, hh
    constexpr int bi[6] = {2, 1, 0, 1, 0, 0};             // pixel term of product i
    bf16x8 wa[3], wb[3], bc[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        wa[t] = __builtin_bit_cast(bf16x8, w0[t * 64]);
        wb[t] = __builtin_bit_cast(bf16x8, w1[t * 64]);
        bc[t] = __builtin_bit_cast(bf16x8, xb[(t * G + half) * TP]);
    }
    const int steps = active ? S : 0;
#pragma unroll 2
    for (int s = 0; s < steps; ++s) {
        bf16x8 na[3], nb[3], nc[3];
        const int sn = s + 1 < S ? s + 1 : s;             // (the last step reloads itself: no branch in the stream)
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            na[t] = __builtin_bit_cast(bf16x8, w0[(sn * 3 + t) * 64]);
            nb[t] = __builtin_bit_cast(bf16x8, w1[(sn * 3 + t) * 64]);
            nc[t] = __builtin_bit_cast(bf16x8, xb[(t * G + 2 * sn + half) * TP]);
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[wi[i]], bc[bi[i]], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb[wi[i]], bc[bi[i]], acc1, 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < 3; ++t) { wa[t] = na[t]; wb[t] = nb[t]; bc[t] = nc[t]; }
    }
    if (a.out_sp) pw_store_sp<UP>(a, reinterpret_cast<float *>(xs), n, m_base, p0, 64 * WPB, PB, tid, pbk, rt, lane, active, second, acc0, acc1);
    else pw_store<UP>(a, n, m0, second, p0 + pbk * 32 + p, half, acc0, acc1);
}

template <int UP>
__global__ __launch_bounds__(256) void pointwise_emu_kernel(const PwArgs a) {
    extern __shared__ uint4 xs[];
    pointwise_emu_body<UP>(a, xs, blockIdx.x, blockIdx.y, blockIdx.z);
}

// Round 6: SEVERAL pointwise layers in ONE launch -- the up-sampling heads of the three scales (base_bev_backbone_resnet.py:121-138) are independent GEMMs writing
// disjoint channel slices of one concatenated map, each a few hundred workgroups of latency-bound work (2 200 ... 35 200 pixels of ONE fused map: 20 / 19 / 40 us
// alone on the GPU for 4 us worth of matrix work).  Back to back on a stream they leave most of the chip idle three times; as one launch their workgroups run
// side by side.  A block's layer comes from its index (the longest layer's blocks first).
constexpr int kMultiMax = 4;
struct PwMulti {
    PwArgs l[kMultiMax];
    int first[kMultiMax + 1], gx[kMultiMax], gy[kMultiMax];
    int n;
};

__global__ __launch_bounds__(256) void pointwise_emu_multi_kernel(const PwMulti m) {
    extern __shared__ uint4 xs[];
    const int bid = blockIdx.x;
    // (a struct member selected by a runtime index would live in scratch: one branch per slot keeps the arguments in scalar registers; first[k] = the grid size
    //  for every unused slot)
#define COALIGN_PW_SLOT(K)                                                                  \
    if (bid >= m.first[K] && bid < m.first[K + 1]) {                                        \
        const PwArgs &a = m.l[K];                                                           \
        const int local = bid - m.first[K], gx = m.gx[K], gy = m.gy[K];                     \
        const int bx = local % gx, by = (local / gx) % gy, bz = local / (gx * gy);          \
        if (a.up == 4) pointwise_emu_body<4>(a, xs, bx, by, bz);                            \
        else if (a.up == 2) pointwise_emu_body<2>(a, xs, bx, by, bz);                       \
        else pointwise_emu_body<1>(a, xs, bx, by, bz);                                      \
        return;                                                                             \
    }
    COALIGN_PW_SLOT(0)
    COALIGN_PW_SLOT(1)
    COALIGN_PW_SLOT(2)
    COALIGN_PW_SLOT(3)
#undef COALIGN_PW_SLOT
}

}  // namespace

extern "C" int coalign_pointwise_conv(const float *x, const float *w, const float *bias, float *y, int N, int Cin, int Hin, int Win,
                                      int in_stride, int Cout, int up, int M_padded, int Ctot, int c_off, int relu, void *stream) {
    return coalign_pointwise_conv_ex(x, w, bias, y, N, Cin, Hin, Win, in_stride, Cout, up, M_padded, Ctot, c_off, relu, 0, stream);
}

// argument checks + launch record of one layer; `grid` / `lds` as the single-layer launch would use them
static int pointwise_prepare(const float *x, const float *w, const float *bias, float *y, int N, int Cin, int Hin, int Win,
                             int in_stride, int Cout, int up, int M_padded, int Ctot, int c_off, int relu, int in_nhwc, bool emu,
                             const void *stamps, const int32_t *state, bool out_sp, int32_t *range_flag, int sparse_rows, PwArgs &a, dim3 &grid, size_t &lds) {
    using namespace coalign;
    if (!x || !w || !bias || !y) return COALIGN_ERR_NULL_POINTER;
    if (emu && ((Cin & 15) || (reinterpret_cast<uintptr_t>(w) & 15))) return COALIGN_ERR_UNSUPPORTED;
    if (N < 0 || Cin < 1 || Hin < 1 || Win < 1 || Cout < 1 || Ctot < Cout || c_off < 0 || c_off + Cout > Ctot) return COALIGN_ERR_BAD_SHAPE;
    if ((in_nhwc & 1) && ((Cin & 3) || (reinterpret_cast<uintptr_t>(x) & 15))) return COALIGN_ERR_UNSUPPORTED;
    if (Cin > (emu ? 2 * kMaxCin : kMaxCin) || (Cin & 1) || (up != 1 && up != 2 && up != 4) || (in_stride != 1 && in_stride != 2) || (up != 1 && in_stride != 1))
        return COALIGN_ERR_UNSUPPORTED;
    const int M = Cout * up * up;
    if (M_padded < M || M_padded % 32 || (up != 1 && M_padded != M)) return COALIGN_ERR_BAD_SHAPE;
    // in_nhwc: bit 0 = channels-last input, bit 1 = channels-last output (up = 1; Ctot, c_off multiples of 4, 16-byte aligned y and bias)
    const int out_nhwc = (in_nhwc >> 1) & 1;
    in_nhwc &= 1;
    if (out_nhwc && (up != 1 || (Ctot & 3) || (c_off & 3) || ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(bias)) & 15))) return COALIGN_ERR_UNSUPPORTED;
    a = PwArgs{x, w, bias, y, N, Cin, Hin, Win, in_stride, (Hin + in_stride - 1) / in_stride, (Win + in_stride - 1) / in_stride, M_padded, up, Cout, Ctot, c_off, relu, in_nhwc != 0, 1, out_nhwc,
               static_cast<const unsigned long long *>(stamps), state, (unsigned)(sparse_rows < 0 ? 0 : sparse_rows), out_sp ? 1 : 0, range_flag};
    if (out_sp && (!emu || out_nhwc || (Cout & 15) || (Ctot & 15) || (c_off & 15) || (reinterpret_cast<uintptr_t>(y) & 15) || (reinterpret_cast<uintptr_t>(bias) & 15) ||
                   M_padded != Cout * up * up || (size_t)N * Ctot * a.Hp * up * a.Wp * up >= ((size_t)1 << 33)))
        return COALIGN_ERR_UNSUPPORTED;
    if (stamps && (!in_nhwc || !state || (reinterpret_cast<uintptr_t>(stamps) & 7))) return COALIGN_ERR_UNSUPPORTED;
    if (N > 65535) return COALIGN_ERR_UNSUPPORTED;
    const int pixels = a.Hp * a.Wp;
    static const int pb_max = coalign::lab_env("COALIGN_PW_PB", 4);      // laboratory build: 1 = the round-1 mapping
    a.pb = M_padded <= 64 ? 4 : M_padded <= 128 ? 2 : 1;
    while (a.pb > 1 && (a.pb > pb_max || Cin * 32 * a.pb > kMaxCin * 32)) a.pb >>= 1;
    if (up == 4 && ((a.Wp * 4) % 4 || (reinterpret_cast<uintptr_t>(y) & 15))) return COALIGN_ERR_UNSUPPORTED;
    if (up == 2 && (reinterpret_cast<uintptr_t>(y) & 7)) return COALIGN_ERR_UNSUPPORTED;
    const int rows_per_wg = 64 * (4 / a.pb), px_per_wg = 32 * a.pb;
    grid = dim3((pixels + px_per_wg - 1) / px_per_wg, (M_padded + rows_per_wg - 1) / rows_per_wg, N);
    lds = 0;
    if (emu) {
        lds = (size_t)Cin * px_per_wg * 6;                        // three bf16 terms of the pixel tile (<= 48 KB up to 256 input channels, 96 KB at 512)
        if (out_sp && lds < (size_t)256 * 33 * 4) lds = (size_t)256 * 33 * 4;      // the SP epilogue parks the workgroup's 256 rows x 32 pixels of floats there
    }
    return COALIGN_OK;
}

static int pointwise_big_lds(size_t lds) {                       // beyond 64 KB of dynamic LDS the kernels need the attribute -- the merged heads of a 384-channel map -- and the
    using namespace coalign;                                     // attribute belongs to the DEVICE's code object: one flag per device
    static bool big_lds_dev[16] = {false};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
    bool &big_lds = big_lds_dev[dev];
    if (lds > 48 * 1024 && !big_lds) {
        const void *fns[4] = {reinterpret_cast<const void *>(pointwise_emu_kernel<1>), reinterpret_cast<const void *>(pointwise_emu_kernel<2>),
                              reinterpret_cast<const void *>(pointwise_emu_kernel<4>), reinterpret_cast<const void *>(pointwise_emu_multi_kernel)};
        for (const void *fn : fns) {
            const int rc = hip_call(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
            if (rc != COALIGN_OK) {
                (void)hipGetLastError();
                return rc;
            }
        }
        big_lds = true;
    }
    return COALIGN_OK;
}

static int pointwise_impl(const float *x, const float *w, const float *bias, float *y, int N, int Cin, int Hin, int Win,
                          int in_stride, int Cout, int up, int M_padded, int Ctot, int c_off, int relu, int in_nhwc, bool emu,
                          void *stream, const void *stamps = nullptr, const int32_t *state = nullptr, bool out_sp = false, int32_t *range_flag = nullptr, int sparse_rows = 0) {
    using namespace coalign;
    PwArgs a;
    dim3 grid;
    size_t lds;
    int rc = pointwise_prepare(x, w, bias, y, N, Cin, Hin, Win, in_stride, Cout, up, M_padded, Ctot, c_off, relu, in_nhwc, emu, stamps, state, out_sp, range_flag, sparse_rows, a, grid, lds);
    if (rc != COALIGN_OK) return rc;
    if (N == 0) return COALIGN_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (emu) {
        rc = pointwise_big_lds(lds);
        if (rc != COALIGN_OK) return rc;
        if (up == 4) hipLaunchKernelGGL(pointwise_emu_kernel<4>, grid, dim3(256), lds, s, a);
        else if (up == 2) hipLaunchKernelGGL(pointwise_emu_kernel<2>, grid, dim3(256), lds, s, a);
        else hipLaunchKernelGGL(pointwise_emu_kernel<1>, grid, dim3(256), lds, s, a);
        return check_launch();
    }
    if (up == 4) hipLaunchKernelGGL(pointwise_kernel<4>, grid, dim3(256), 0, s, a);
    else if (up == 2) hipLaunchKernelGGL(pointwise_kernel<2>, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(pointwise_kernel<1>, grid, dim3(256), 0, s, a);
    return check_launch();
}

extern "C" int coalign_pointwise_conv_ex(const float *x, const float *w, const float *bias, float *y, int N, int Cin, int Hin, int Win,
                                         int in_stride, int Cout, int up, int M_padded, int Ctot, int c_off, int relu, int in_nhwc,
                                         void *stream) {
    return pointwise_impl(x, w, bias, y, N, Cin, Hin, Win, in_stride, Cout, up, M_padded, Ctot, c_off, relu, in_nhwc, false, stream);
}

extern "C" size_t coalign_pointwise_emu_weight_bytes(int Cin, int M_padded) {
    if (Cin < 16 || (Cin & 15) || Cin > 2 * kMaxCin || M_padded < 32 || (M_padded & 31)) return 0;
    return (size_t)M_padded * Cin * 6;
}

extern "C" int coalign_pointwise_conv_emu(const float *x, const void *w_split, const float *bias, float *y, int N, int Cin, int Hin, int Win,
                                          int in_stride, int Cout, int up, int M_padded, int Ctot, int c_off, int relu, int in_nhwc,
                                          void *stream) {
    return pointwise_impl(x, static_cast<const float *>(w_split), bias, y, N, Cin, Hin, Win, in_stride, Cout, up, M_padded, Ctot, c_off, relu, in_nhwc,
                          true, stream);
}

// Round 4: the 1 x 1 / stride-2 skip convolution of the first ResNet stage reading the SPARSE canvas of csrc/pillar_sparse.hip (feature rows + cell stamps)
// instead of the dense canvas.  out_nhwc: bit 0 = channels-last output.
extern "C" int coalign_pointwise_conv_emu_sparse(const float *feats, int M_rows, const void *stamps, const int32_t *state, const void *w_split, const float *bias, float *y,
                                                 int N, int Cin, int Hin, int Win, int Cout, int M_padded, int relu, int out_nhwc, void *stream) {
    if (!stamps || !state) return COALIGN_ERR_NULL_POINTER;
    if (M_rows < 0) return COALIGN_ERR_BAD_SHAPE;
    return pointwise_impl(feats, static_cast<const float *>(w_split), bias, y, N, Cin, Hin, Win, 2, Cout, 1, M_padded, Cout, 0, relu, 1 | (out_nhwc ? 2 : 0), true, stream,
                          stamps, state, false, nullptr, M_rows);
}

// Round 5: the up-sampling heads writing their channel slice of the concatenated map as an SP map (include/coalign_amd.h (9e)): the shrink header's first
// 3 x 3 convolution then reads it by LDS-DMA (coalign_conv3x3_sp) instead of splitting 54 MB of float32 in its own K loop.  Cout, Ctot, c_off multiples of 16.
extern "C" int coalign_pointwise_conv_emu_sp(const float *x, const void *w_split, const float *bias, void *y_sp, int N, int Cin, int Hin, int Win, int in_stride,
                                             int Cout, int up, int M_padded, int Ctot, int c_off, int relu, int in_nhwc, int32_t *range_flag, void *stream) {
    if (in_nhwc & ~1) return COALIGN_ERR_UNSUPPORTED;
    return pointwise_impl(x, static_cast<const float *>(w_split), bias, static_cast<float *>(y_sp), N, Cin, Hin, Win, in_stride, Cout, up, M_padded, Ctot, c_off, relu, in_nhwc,
                          true, stream, nullptr, nullptr, true, range_flag);
}

// Round 6: the up-sampling heads of ALL scales in one launch (10c, several layers): layer i reads x[i] ([N, Cin[i], Hin[i], Win[i]], channels-last if in_nhwc[i]) and
// writes channels [c_off[i], c_off[i] + Cout[i]) of the SP map y_sp [N, Ctot, Hin[i] * up[i], Win[i] * up[i]] (the same output size for every layer).  Results
// are those of n_layers calls of coalign_pointwise_conv_emu_sp, bit for bit.
extern "C" int coalign_pointwise_conv_emu_sp_multi(int n_layers, const void *const *x, const void *const *w_split, const void *const *bias, const int32_t *Cin, const int32_t *Hin,
                                                   const int32_t *Win, const int32_t *Cout, const int32_t *up, const int32_t *c_off, const int32_t *in_nhwc, void *y_sp, int N, int Ctot,
                                                   int relu, int32_t *range_flag, void *stream) {
    using namespace coalign;
    if (!x || !w_split || !bias || !Cin || !Hin || !Win || !Cout || !up || !c_off || !in_nhwc || !y_sp) return COALIGN_ERR_NULL_POINTER;
    if (n_layers < 1 || n_layers > kMultiMax) return COALIGN_ERR_UNSUPPORTED;
    PwMulti m{};
    m.n = n_layers;
    size_t lds = 0;
    int order[kMultiMax];
    long long work[kMultiMax];
    for (int i = 0; i < n_layers; ++i) {
        if (in_nhwc[i] & ~1) return COALIGN_ERR_UNSUPPORTED;
        if (Hin[i] * up[i] != Hin[0] * up[0] || Win[i] * up[i] != Win[0] * up[0]) return COALIGN_ERR_BAD_SHAPE;
        order[i] = i;
        work[i] = (long long)Cin[i] * Cout[i] * up[i] * up[i];                     // the longest workgroups first
    }
    for (int i = 0; i < n_layers; ++i)
        for (int j = i + 1; j < n_layers; ++j)
            if (work[order[j]] > work[order[i]]) { const int t = order[i]; order[i] = order[j]; order[j] = t; }
    for (int k = 0; k < n_layers; ++k) {
        const int i = order[k];
        dim3 grid;
        size_t l;
        const int rc = pointwise_prepare(static_cast<const float *>(x[i]), static_cast<const float *>(w_split[i]), static_cast<const float *>(bias[i]), static_cast<float *>(y_sp), N, Cin[i],
                                         Hin[i], Win[i], 1, Cout[i], up[i], Cout[i] * up[i] * up[i], Ctot, c_off[i], relu, in_nhwc[i], true, nullptr, nullptr, true, range_flag, 0, m.l[k], grid, l);
        if (rc != COALIGN_OK) return rc;
        m.gx[k] = (int)grid.x;
        m.gy[k] = (int)grid.y;
        m.first[k + 1] = m.first[k] + (int)(grid.x * grid.y * grid.z);
        if (l > lds) lds = l;
    }
    for (int k = n_layers; k < kMultiMax; ++k) m.first[k + 1] = m.first[n_layers];
    if (N == 0) return COALIGN_OK;
    const int rc = pointwise_big_lds(lds);
    if (rc != COALIGN_OK) return rc;
    hipLaunchKernelGGL(pointwise_emu_multi_kernel, dim3(m.first[n_layers]), dim3(256), lds, static_cast<hipStream_t>(stream), m);
    return check_launch();
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Round 6: the merged cls / reg / dir 1 x 1 heads (point_pillar_baseline_multiscale.py:123-133: three nn.Conv2d(C, k, 1) on the shrink header's map) reading that map AS
// AN SP MAP (csrc/conv3x3_sp.hip) -- the shrink header's last convolution writes sp16 pairs in matrix-operand order, so a B operand is ONE 16-byte global load per lane
// (32 consecutive pixels of a plane = 512 contiguous bytes): no pixel staging through LDS, no split on the VALU, and the sp16 arithmetic of the 3 x 3 layers (three fp16
// products) instead of the six bf16 ones.  The pointwise kernel above took 25 us for this layer (32 pixels per workgroup, three of four wavefronts idle behind the
// staging); here a wavefront owns 32 pixels x the (<= 32) head channels and streams 2 x 2 x 16 bytes per lane and 16 input channels, four steps ahead.
// Weight image: ops.pack_conv1x1_sp_weight of the head weights padded to 64 rows ((9g): [Cin / 16][2 terms][2 channel halves][64 rows][8 cin] + zero group + 2^-k / 2^k words).
namespace {

typedef _Float16 hd_halfx8 __attribute__((ext_vector_type(8)));

struct HeadsArgs {
    const uint4 *__restrict__ x, *__restrict__ w;
    const float *__restrict__ winv, *__restrict__ bias;
    float *__restrict__ y;
    int N, Cin, HW, M;
};

constexpr int kHeadsAhead = 4;      // steps (of 16 input channels) whose operands are in flight

__global__ __launch_bounds__(256) void heads_sp_kernel(const HeadsArgs a) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, half = lane >> 5, p = lane & 31;
    const int n = blockIdx.y, px0 = (blockIdx.x * 4 + wave) * 32;
    if (px0 >= a.HW) return;                                          // wave-uniform
    const int px = px0 + p, pix = px < a.HW ? px : a.HW - 1;
    const int CI16 = a.Cin / 16;
    const uint4 *xb = a.x + ((size_t)n * CI16 * 4 + 2 * half) * a.HW + pix;      // plane (2 * half + term) of interval c: + (4 c + term) * HW
    const uint4 *wb = a.w + half * 64 + p;                                       // term t of interval c: + (4 c + 2 t) * 64
    floatx16 acc = {0}, accl = {0};
    uint4 xh[kHeadsAhead], xl[kHeadsAhead], wh[kHeadsAhead], wl[kHeadsAhead];
    auto load = [&](int c, int slot) {
        xh[slot] = xb[(size_t)(4 * c) * a.HW];
        xl[slot] = xb[(size_t)(4 * c + 1) * a.HW];
        wh[slot] = wb[(4 * c) * 64];
        wl[slot] = wb[(4 * c + 2) * 64];
    };
#pragma unroll
    for (int k = 0; k < kHeadsAhead; ++k) load(k < CI16 ? k : CI16 - 1, k);
    for (int c0 = 0; c0 < CI16; c0 += kHeadsAhead) {
#pragma unroll
        for (int k = 0; k < kHeadsAhead; ++k) {
            const int c = c0 + k;
            if (c < CI16) {
                const hd_halfx8 bh = __builtin_bit_cast(hd_halfx8, xh[k]), bl = __builtin_bit_cast(hd_halfx8, xl[k]);
                const hd_halfx8 ah = __builtin_bit_cast(hd_halfx8, wh[k]), al = __builtin_bit_cast(hd_halfx8, wl[k]);
                const int cn = c + kHeadsAhead;
                load(cn < CI16 ? cn : CI16 - 1, k);                   // (the tail reloads the last step: no branch in the stream)
                accl = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, accl, 0, 0, 0);      // w_h x_l'
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);        // w_h x_h
                accl = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, accl, 0, 0, 0);      // w_l' x_h
            }
        }
    }
    if (px >= a.HW) return;
    float *yo = a.y + (size_t)n * a.M * a.HW + px;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int m = 8 * (e / 4) + 4 * half + (e % 4);               // accumulator e of lane (half, p): GEMM row m, pixel p
        if (m < a.M) yo[(size_t)m * a.HW] = fmaf(accl[e], coalign::kSp16LowInv, acc[e]) * a.winv[m] + a.bias[m];
    }
}

}  // namespace

extern "C" int coalign_heads_sp(const void *x_sp, const void *w_sp, const float *bias, float *y, int N, int Cin, int M, int H, int W, void *stream) {
    using namespace coalign;
    if (!x_sp || !w_sp || !bias || !y) return COALIGN_ERR_NULL_POINTER;
    if (N < 0 || Cin < 16 || M < 1 || H < 1 || W < 1) return COALIGN_ERR_BAD_SHAPE;
    if ((Cin & 15) || M > 32 || N > 65535 || ((reinterpret_cast<uintptr_t>(x_sp) | reinterpret_cast<uintptr_t>(w_sp)) & 15)) return COALIGN_ERR_UNSUPPORTED;
    if ((int64_t)N * Cin * H * W > (int64_t)1 << 32) return COALIGN_ERR_UNSUPPORTED;
    if (N == 0) return COALIGN_OK;
    HeadsArgs a{static_cast<const uint4 *>(x_sp), static_cast<const uint4 *>(w_sp),
                reinterpret_cast<const float *>(static_cast<const char *>(w_sp) + (size_t)64 * Cin * 4 + 16), bias, y, N, Cin, H * W, M};
    hipLaunchKernelGGL(heads_sp_kernel, dim3((unsigned)((H * W + 127) / 128), (unsigned)N), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    return check_launch();
}
