// PillarVFE + PointPillarScatter as ONE launch with a SPARSE canvas (round 4), gfx950.
//
// Reference semantics: opencood/models/sub_modules/pillar_vfe.py:31-53,105-155 and opencood/models/sub_modules/point_pillar_scatter.py:15-72.
//
// The dense canvas of the reference is 94 % zeros and exists only to be read by the first strided 3x3 convolution and the 1x1 skip
// convolution.  Here the "scatter" is 8 bytes per pillar:
//   feats  [M, C]            the pillar feature rows (pillar_features of the reference; every pillar writes its own row: no write races)
//   stamps [n_agents*ny*nx]  one 64-bit word per cell = (frame tag << 32) | pillar row, entered with a device-scope atomicMax: two pillars of one
//                            cell resolve to the larger row index -- the reference's sequential-indexing rule -- without a pre-pass
//   state  [528]             state[0] = tag of the last completed frame, state[1] and state[16 + 16 g] = arrival counters of the launch in flight (round 6: two levels)
// A cell is occupied in the current frame iff its stamp carries the current tag: nothing is ever cleared (the previous frames' stamps are stale by
// their tag), there is no cell-map pre-pass and no ordering problem between clearing and writing -- one launch.  The consumers
// (coalign_conv3x3_emu_sparse, coalign_pointwise_conv_emu_sparse) look a pixel up as stamps[cell] -> row of feats, or zero.
// The tag is advanced by the LAST workgroup to finish (arrival counter), i.e. after every stamp of this frame has been entered and before any consumer
// (next launch on the stream) reads it; the launch reads state[0] only at its start.
//
// Encoder: the PFN input of a point is affine in the point once the pillar is fixed:
//     W f = (w_abs + w_cluster + w_center) (p - c) + w_i intensity + [w_abs c - w_cluster (mean - c)]
// a K = 4 contraction per point and channel; the bracket is evaluated per pillar in fp32.  Round 4 ran the contraction exactly in fp32 on
// v_mfma_f32_32x32x2_f32: 8 instructions of 64 cycles per pillar pair on the port the kernel's VALU work needs too (pmc_summary.json: the kernel was
// VALU-bound at 0.21 of the HBM roofline).  Round 5: ONE v_mfma_f32_32x32x16_f16 per (pillar, 32 channels): the 16 K slots hold the three product groups of
// a 22-bit operand split,
//     lanes 0-31  (k 0-7):   A = [dh0 dh1 dh2 dh3 | dh0 dh1 dh2 dh3]    B = [wh0 wh1 wh2 wh3 | wl0 wl1 wl2 wl3]
//     lanes 32-63 (k 8-15):  A = [dl0 dl1 dl2 dl3 |  0   0   0   0 ]    B = [wh0 wh1 wh2 wh3 |  0   0   0   0 ]
// d = (p - c) * 2^6 and the intensity * 2^6 (the offsets inside a 0.4 m cell are small: scaled, the low terms stay normal fp16 numbers down to offsets of
// 2 mm, below that they keep an absolute 5e-10 m), dh = fp16(d), dl = fp16(d - dh) (exact remainder); w = the summed weights times the per-channel power of
// two that puts the largest of the four into [2^13, 2^14), wh = fp16(w), wl = fp16(w - wh).  sum_k (dh wh + dh wl + dl wh) in the fp32 accumulator drops only
// dl wl (< 2^-20 of a product); 2^-(k_c + 6) rides on the sign factor of the epilogue (exact).  The A operands are the lane's own point after two
// v_permlane32_swap.  Against float64 the features stay within 2e-6 of their scale (the bracket, evaluated in fp32 as before, carries the magnitude).
#include <stdlib.h>

#include "common.h"

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void *lptr_sp_t;

constexpr int kWaves = 4;                   // wavefronts per workgroup
constexpr int kRunPairs = 32;               // pillar pairs per wavefront at most (one lane per pillar in the prologue)
constexpr int kPointShift = 6;              // the point offsets enter the fp16 contraction times 2^6 (see the header)
constexpr int kArriveGroups = 32, kArriveBase = 16, kArriveStride = 16;      // state[16 + 16 g]: the arrival counter of workgroups b % 32 == g (see the end of the kernel)
constexpr int kRound = 4;                   // pairs per LDS-DMA round (4 KB per buffer: 10 KB of LDS per wavefront, 40 KB per workgroup; the grid puts three workgroups on a CU).
                                            // Round 6: 3 -> 4 and the first TWO rounds are issued before anything waits, so that a wavefront of the benchmarked size (6-7 pairs)
                                            // has all its points in flight at once instead of fetching its second half after the first has landed and been processed
constexpr int kPairsPerWave = 6;            // pairs per wavefront the grid is sized for
#ifndef COALIGN_PILLAR_GROUP
#define COALIGN_PILLAR_GROUP 2
#endif
constexpr int kGroup = COALIGN_PILLAR_GROUP; // pairs per straight-line scheduling region (see the round loop)
#ifdef COALIGN_LAB
constexpr bool kLab = true;
#else
constexpr bool kLab = false;
#endif

#ifdef COALIGN_LAB
__device__ long long g_sparse_trace[2048 * 4];      // laboratory build (COALIGN_SPARSE_DEBUG bit 9): per workgroup start / end wall clock, hardware id, round-0 cycles
#endif

struct SparseArgs {
    const float4 *pts;
    const int *npts;
    const int4 *coords;
    int M, P;
    const float *weight, *bias, *bn_w, *bn_b, *bn_m, *bn_v;
    float eps;
    int C, Cin, use_abs;
    float vx, vy, vz, xo, yo, zo;
    int n_agents, ny, nx;
    float *feats;
    unsigned long long *stamps;
    int *state;
    const int *M_dev;
    const coalign_pillar_frame *frame;      // non-NULL: the three input arrays and the pillar count are read from this DEVICE record (coalign_pillar_encode_sparse_frame)
    const float4 *folded;  // the folded channel parameters written by coalign_pillar_fold_params ([7][64] float4)
    int debug;            // laboratory build only (COALIGN_SPARSE_DEBUG): 1 no stamp atomics, 2 no matrix steps, 4 no arrival counter, 8 no feature stores
};

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

// sum over each 32-lane half, result in every lane of the half
__device__ __forceinline__ float half_sum_dpp(float v) {
    v += dppf<0xB1>(v);
    v += dppf<0x4E>(v);
    v += dppf<0x141>(v);
    v += dppf<0x140>(v);
    const auto q = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
    const unsigned x = q[0], y = q[1];
    return __builtin_bit_cast(float, x) + __builtin_bit_cast(float, y);
}

__device__ __forceinline__ float max16(const floatx16 &v) {
    const float t0 = fmaxf(fmaxf(v[0], v[1]), v[2]), t1 = fmaxf(fmaxf(v[3], v[4]), v[5]), t2 = fmaxf(fmaxf(v[6], v[7]), v[8]);
    const float t3 = fmaxf(fmaxf(v[9], v[10]), v[11]), t4 = fmaxf(fmaxf(v[12], v[13]), v[14]);
    return fmaxf(fmaxf(fmaxf(t0, t1), fmaxf(t2, t3)), fmaxf(t4, v[15]));
}

// per-lane channel parameters in the half layout: channel 32 g + (lane & 31)
struct F32Chan {
    unsigned wq[2][4];               // B operand of channel group g: 8 fp16 = the lane's K slots (see the header), sign and 2^k_c folded
    float wc[2][3], wen[2][3];       // weights of the centre term, negated weights of the mean-offset term
    float alpha[2], shift[2], sgn[2];
};

// raw per-lane channel parameters (loads only) ...
template <bool ABS>
struct F32Raw {
    static constexpr int CIN = ABS ? 10 : 7;
    float w[2][CIN], bw[2], bb[2], bm[2], bv[2], lb[2];
};

template <bool ABS>
__device__ __forceinline__ F32Raw<ABS> load_f32_raw(const SparseArgs &a, int lane) {
    // every load is unconditional on a clamped channel index and independent of the others: the compiler issues them back to back and waits once
    // (with the loads inside `if (c < C)` blocks this took ten dependent round trips, ~10 000 cycles per wavefront)
    F32Raw<ABS> r;
    const int col = lane & 31;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int cc = min(g * 32 + col, a.C - 1);
#pragma unroll
        for (int k = 0; k < F32Raw<ABS>::CIN; ++k) r.w[g][k] = a.weight[(size_t)cc * F32Raw<ABS>::CIN + k];
        r.bw[g] = a.bn_w ? a.bn_w[cc] : 1.f; r.bb[g] = a.bn_w ? a.bn_b[cc] : 0.f; r.bm[g] = a.bn_w ? a.bn_m[cc] : 0.f; r.bv[g] = a.bn_w ? a.bn_v[cc] : 1.f;
        r.lb[g] = (!a.bn_w && a.bias) ? a.bias[cc] : 0.f;
    }
    return r;
}

// ... and the folded form the pair loop uses
template <bool ABS>
__device__ __forceinline__ F32Chan finish_f32(const F32Raw<ABS> &r, const SparseArgs &a, int lane) {
    F32Chan fc;
    constexpr int B = ABS ? 4 : 1;
    const int half = lane >> 5, col = lane & 31;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const bool live = g * 32 + col < a.C;
        float alpha = 1.f, shift = r.lb[g];
        if (a.bn_w) {
            const float inv_std = 1.0f / sqrtf(r.bv[g] + a.eps);
            alpha = r.bw[g] * inv_std;
            shift = r.bb[g] - r.bm[g] * alpha;
        }
        if (!live) { alpha = 1.f; shift = 0.f; }
        float w4[4];
#pragma unroll
        for (int k = 0; k < 3; ++k) w4[k] = live ? ((ABS ? r.w[g][k] : 0.f) + r.w[g][B + k]) + r.w[g][B + 3 + k] : 0.f;
        w4[3] = live ? r.w[g][ABS ? 3 : 0] : 0.f;
        const float sg = alpha < 0.f ? -1.f : 1.f;       // a negative BatchNorm scale turns the max over the rows into a min: negate the weights instead
        // per-channel power of two: the largest of the four weights into [2^13, 2^14) (exact); with the 2^6 of the point offsets it leaves through `sgn`
        const float wmax = fmaxf(fmaxf(fabsf(w4[0]), fabsf(w4[1])), fmaxf(fabsf(w4[2]), fabsf(w4[3])));
        int kc = wmax > 0.f ? 13 - (int)(((__builtin_bit_cast(unsigned, wmax) >> 23) & 0xffu) - 127u) : 0;
        kc = kc < -60 ? -60 : kc > 60 ? 60 : kc;
        const float wsc = __builtin_bit_cast(float, (unsigned)(127 + kc) << 23), winv = __builtin_bit_cast(float, (unsigned)(127 - kc - kPointShift) << 23);
        fc.alpha[g] = alpha; fc.shift[g] = shift; fc.sgn[g] = sg * winv;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            fc.wc[g][k] = (ABS && live) ? r.w[g][k] : 0.f;
            fc.wen[g][k] = live ? -r.w[g][B + k] : 0.f;
        }
        _Float16 wh[4], wl[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float ws = sg * w4[k] * wsc;
            wh[k] = (_Float16)ws;
            wl[k] = (_Float16)(ws - (float)wh[k]);
        }
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        fc.wq[g][0] = __builtin_bit_cast(unsigned, h2{wh[0], wh[1]});
        fc.wq[g][1] = __builtin_bit_cast(unsigned, h2{wh[2], wh[3]});
        fc.wq[g][2] = half ? 0u : __builtin_bit_cast(unsigned, h2{wl[0], wl[1]});
        fc.wq[g][3] = half ? 0u : __builtin_bit_cast(unsigned, h2{wl[2], wl[3]});
    }
    return fc;
}

// One pillar pair (lanes 0-31: pillar A's points, lanes 32-63: pillar B's) -> y[g]: relu(BN(max over the rows)) of channel 32 g + (lane & 31) in the
// half layout (lanes 0-31 pillar A, lanes 32-63 pillar B).  q: this lane's point as stored (the mean runs over ALL P slots, pillar_vfe.py:118-120);
// qs: the same with the rows at / past the point count replaced by row 0 of their pillar (they cannot change the max; their own value,
// relu(BN(0)), is added below) -- the substitution is the address of the LDS read.  rec0 = (num_points, 1 / num_points, -, -), ctr = the cell centre.
template <bool ABS>
__device__ __forceinline__ void f32_pair_half(const SparseArgs &a, const F32Chan &fc, float4 q, float4 qs, int np_eff, float rn, float ctr_x, float ctr_y, float ctr_z, float (&y)[2]) {
    const float ex = half_sum_dpp(q.x) * rn - ctr_x, ey = half_sum_dpp(q.y) * rn - ctr_y, ez = half_sum_dpp(q.z) * rn - ctr_z;
    // this lane's point as the scaled offset d = (p - c) * 2^6, split into fp16 terms: dh = fp16(d), dl = fp16(d - dh)
    constexpr float kS = (float)(1 << kPointShift);
    const float d0 = (qs.x - ctr_x) * kS, d1 = (qs.y - ctr_y) * kS, d2 = (qs.z - ctr_z) * kS, d3 = qs.w * kS;
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    const h2 h01 = __builtin_convertvector(f2{d0, d1}, h2), h23 = __builtin_convertvector(f2{d2, d3}, h2);
    const h2 l01 = __builtin_convertvector(f2{d0 - (float)h01[0], d1 - (float)h01[1]}, h2), l23 = __builtin_convertvector(f2{d2 - (float)h23[0], d3 - (float)h23[1]}, h2);
    // A operands: lanes 0-31 carry [dh | dh] of the pillar's point l & 31, lanes 32-63 [dl | 0].  One swap per dword turns the lane's own
    // (dh, dl) = ([A.dh | B.dh], [A.dl | B.dl]) into pillar A's operand [A.dh | A.dl] and pillar B's [B.dh | B.dl].
    unsigned pa0 = __builtin_bit_cast(unsigned, h01), pb0 = __builtin_bit_cast(unsigned, l01), pa1 = __builtin_bit_cast(unsigned, h23), pb1 = __builtin_bit_cast(unsigned, l23);
    {
        const auto s0 = __builtin_amdgcn_permlane32_swap(pa0, pb0, false, false);
        const unsigned x0 = s0[0], y0 = s0[1];
        const auto s1 = __builtin_amdgcn_permlane32_swap(pa1, pb1, false, false);
        const unsigned x1 = s1[0], y1 = s1[1];
        pa0 = x0; pb0 = y0; pa1 = x1; pb1 = y1;
    }
    const bool hi = (threadIdx.x & 32) != 0;
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    const h8 opA = __builtin_bit_cast(h8, uint4{pa0, pa1, hi ? 0u : pa0, hi ? 0u : pa1}), opB = __builtin_bit_cast(h8, uint4{pb0, pb1, hi ? 0u : pb0, hi ? 0u : pb1});
    const floatx16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    floatx16 accA[2], accB[2];
    if (kLab && (a.debug & 2)) {
#pragma unroll
        for (int g = 0; g < 2; ++g) { accA[g] = zero; accB[g] = zero; accA[g][0] = d0 + d2; accB[g][0] = d1 + d3; }
    } else {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const h8 w = __builtin_bit_cast(h8, uint4{fc.wq[g][0], fc.wq[g][1], fc.wq[g][2], fc.wq[g][3]});
            accA[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(opA, w, zero, 0, 0, 0);
            accB[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(opB, w, zero, 0, 0, 0);
        }
    }
    const bool padded = np_eff < a.P, empty = np_eff == 0;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        // lane holds 16 of the 32 rows of channel 32 g + (lane & 31): rows 8 (r >> 2) + 4 (lane >> 5) + (r & 3); the two halves complete each other
        float mA = max16(accA[g]), mB = max16(accB[g]);
        swap32(mA, mB);                                     // mA = [A.lo | B.lo], mB = [A.hi | B.hi]
        const float r = fmaxf(mA, mB);                      // lanes 0-31: pillar A, lanes 32-63: pillar B
        float b = fc.wen[g][0] * ex;
        b = fmaf(fc.wen[g][1], ey, b); b = fmaf(fc.wen[g][2], ez, b);
        if constexpr (ABS) { b = fmaf(fc.wc[g][0], ctr_x, b); b = fmaf(fc.wc[g][1], ctr_y, b); b = fmaf(fc.wc[g][2], ctr_z, b); }
        float v = fmaf(fmaf(fc.sgn[g], r, b), fc.alpha[g], fc.shift[g]);      // (sgn = +-1: sgn * r is exact, the fused form rounds once as the separate one did)
        // padded rows: Linear(0) = 0 -> BN -> shift; no points at all: shift alone (documented deviation: the reference divides by zero there)
        v = fmaxf(empty ? fc.shift[g] : v, padded ? fc.shift[g] : v);
        y[g] = fmaxf(v, 0.f);
    }
}

// the folded parameters of one lane as six float4 (coalign_pillar_fold_params writes them, the kernel reads them back in the same order)
constexpr int kFoldVecs = 7;
__device__ __forceinline__ void chan_to_vec(const F32Chan &fc, float4 (&t)[kFoldVecs]) {
    auto f = [](unsigned u) { return __builtin_bit_cast(float, u); };
    t[0] = make_float4(f(fc.wq[0][0]), f(fc.wq[0][1]), f(fc.wq[0][2]), f(fc.wq[0][3]));
    t[6] = make_float4(f(fc.wq[1][0]), f(fc.wq[1][1]), f(fc.wq[1][2]), f(fc.wq[1][3]));
    t[1] = make_float4(fc.wc[0][0], fc.wc[0][1], fc.wc[0][2], fc.wc[1][0]);
    t[2] = make_float4(fc.wc[1][1], fc.wc[1][2], fc.wen[0][0], fc.wen[0][1]);
    t[3] = make_float4(fc.wen[0][2], fc.wen[1][0], fc.wen[1][1], fc.wen[1][2]);
    t[4] = make_float4(fc.alpha[0], fc.alpha[1], fc.shift[0], fc.shift[1]);
    t[5] = make_float4(fc.sgn[0], fc.sgn[1], 0.f, 0.f);
}
__device__ __forceinline__ F32Chan vec_to_chan(const float4 (&t)[kFoldVecs]) {
    F32Chan fc;
    auto u = [](float f) { return __builtin_bit_cast(unsigned, f); };
    fc.wq[0][0] = u(t[0].x); fc.wq[0][1] = u(t[0].y); fc.wq[0][2] = u(t[0].z); fc.wq[0][3] = u(t[0].w);
    fc.wq[1][0] = u(t[6].x); fc.wq[1][1] = u(t[6].y); fc.wq[1][2] = u(t[6].z); fc.wq[1][3] = u(t[6].w);
    fc.wc[0][0] = t[1].x; fc.wc[0][1] = t[1].y; fc.wc[0][2] = t[1].z; fc.wc[1][0] = t[1].w;
    fc.wc[1][1] = t[2].x; fc.wc[1][2] = t[2].y; fc.wen[0][0] = t[2].z; fc.wen[0][1] = t[2].w;
    fc.wen[0][2] = t[3].x; fc.wen[1][0] = t[3].y; fc.wen[1][1] = t[3].z; fc.wen[1][2] = t[3].w;
    fc.alpha[0] = t[4].x; fc.alpha[1] = t[4].y; fc.shift[0] = t[4].z; fc.shift[1] = t[4].w;
    fc.sgn[0] = t[5].x; fc.sgn[1] = t[5].y;
    return fc;
}

// the fold as its own one-wavefront launch (once per weight set, like the convolutions' split weight images): the same device code, so the table holds
// exactly what a wavefront would have computed for itself
template <bool ABS>
__global__ __launch_bounds__(64) void pillar_fold_kernel(SparseArgs a, float4 *out) {
    const int lane = threadIdx.x & 63;
    const F32Chan fc = finish_f32<ABS>(load_f32_raw<ABS>(a, lane), a, lane);
    float4 t[kFoldVecs];
    chan_to_vec(fc, t);
#pragma unroll
    for (int i = 0; i < kFoldVecs; ++i) out[i * 64 + lane] = t[i];
}

template <bool ABS>
__global__ __launch_bounds__(kWaves * 64) void pillar_sparse_kernel(SparseArgs a) {
    constexpr int kWaveLds = 2 * kRound * 1024 + 64 * 32;           // two point buffers | the run's pillar records
    __shared__ __attribute__((aligned(16))) char lds[kWaves * kWaveLds];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), half = lane >> 5, col = lane & 31;
    char *pbuf = lds + wv * kWaveLds;
    char *meta = pbuf + 2 * kRound * 1024;
    const int gwave = blockIdx.x * kWaves + wv, nwave = gridDim.x * kWaves;
    if (kLab && (a.debug & 128)) return;
    if (a.frame) {      // a replayed graph reads whatever arrays the frame record names NOW: no copy of the inputs into launch-time addresses
        a.pts = reinterpret_cast<const float4 *>(a.frame->voxel_features);
        a.npts = a.frame->voxel_num_points;
        a.coords = reinterpret_cast<const int4 *>(a.frame->voxel_coords);
        a.M = min(max(a.frame->M, 0), a.M);
    }
    if (a.M_dev) a.M = min(max(*a.M_dev, 0), a.M);
    long long ts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (kLab && (a.debug & 512)) { ts[0] = (long long)__builtin_amdgcn_s_memtime(); ts[7] = (long long)wall_clock64(); }
    const unsigned tag = (unsigned)a.state[0] + 1u;                  // this frame's tag (state[0] is written only when the last workgroup has finished)
    const int npairs = (a.M + 1) / 2;
    // even shares (+-1 pair): with a grid that is a whole number of workgroups per CU every SIMD gets the same work.  The grid is sized for the CAPACITY of the
    // arrays; when the device-side count is far below it (the voxeliser's route: room for 5 x 70 000 pillars, ~36 000 present) only as many wavefronts as give
    // each about kPairsPerWave pairs take part -- a wavefront's prologue is not worth one or two pairs
    const int active = max(1, min(nwave, (npairs + kPairsPerWave - 1) / kPairsPerWave));
    const int p0 = gwave < active ? (int)((long long)gwave * npairs / active) : 0, p1 = gwave < active ? (int)((long long)(gwave + 1) * npairs / active) : 0;
    if (p0 < p1) {
        const int ncell = a.ny * a.nx;
        const char *pts_b = reinterpret_cast<const char *>(a.pts);
        const char *np_b = reinterpret_cast<const char *>(a.npts), *cd_b = reinterpret_cast<const char *>(a.coords);
        char *feat_b = reinterpret_cast<char *>(a.feats);
        // a round's points go straight into LDS (global_load_lds issued from inline assembly, see pillar_scatter.hip)
        auto issue_round = [&](int r0, int buf) {
            const int nr = min(kRound, p1 - r0);
#pragma unroll
            for (int k = 0; k < kRound; ++k) {
                if (k < nr && !(kLab && (a.debug & 64))) {
                    const int m = min(2 * (r0 + k) + half, a.M - 1);
                    const char *src = pts_b + (unsigned)(m * a.P + min(col, a.P - 1)) * 16u;
                    const unsigned dst = (unsigned)(size_t)(lptr_sp_t)(pbuf + (buf * kRound + k) * 1024);
                    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(__builtin_amdgcn_readfirstlane(dst)), "v"(src) : "memory", "m0");
                }
            }
        };
        issue_round(p0, 0);
        if (p0 + kRound < p1) issue_round(p0 + kRound, 1);
        // counts, coordinates, cells of the whole run: once per wavefront (one lane per pillar); the channel parameters' loads go out before anything waits
        const bool pro = lane < 2 * (p1 - p0);
        const int m_j = 2 * p0 + lane, mj = min(m_j, a.M - 1);
        int np_j = 0;
        int4 cd_j = make_int4(0, 0, 0, 0);
        if (pro) {
            np_j = *reinterpret_cast<const int *>(np_b + (unsigned)mj * 4u);
            cd_j = *reinterpret_cast<const int4 *>(cd_b + (unsigned)mj * 16u);
        }
        float4 ft[kFoldVecs];
#pragma unroll
        for (int i = 0; i < kFoldVecs; ++i) ft[i] = a.folded[i * 64 + lane];
        if (pro) {
            const int cell = cd_j.y + cd_j.z * a.nx + cd_j.w;       // z + y * nx + x (point_pillar_scatter.py:54)
            const bool ok = m_j < a.M && cd_j.x >= 0 && cd_j.x < a.n_agents && cell >= 0 && cell < ncell;
            if (ok && !(kLab && (a.debug & 1))) atomicMax(a.stamps + (size_t)cd_j.x * ncell + cell, ((unsigned long long)tag << 32) | (unsigned)m_j);      // the larger row of a cell wins
            // the pillar's record: point count (clamped), 1 / num_points (pillar_vfe.py:118-120 divides by the count as given), cell centre
            const float rn = __builtin_amdgcn_rcpf((float)np_j);
            const float ctr_x = (float)cd_j.w * a.vx + a.xo, ctr_y = (float)cd_j.z * a.vy + a.yo, ctr_z = (float)cd_j.y * a.vz + a.zo;
            float4 *rec = reinterpret_cast<float4 *>(meta + lane * 32);
            rec[0] = make_float4(__builtin_bit_cast(float, min(max(np_j, 0), a.P)), rn, ctr_x, ctr_y);
            rec[1] = make_float4(ctr_z, 0.f, 0.f, 0.f);
        }
        const F32Chan fc = vec_to_chan(ft);
        int buf = 0;
        for (int r0 = p0; r0 < p1; r0 += kRound, buf ^= 1) {
            const int nr = min(kRound, p1 - r0);
            if (r0 != p0 + kRound) {                                 // (the second round landed under the first round's wait)
                __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0): this round's points are in LDS
                coalign::wave_lds_sync();
            }
            if (kLab && (a.debug & 512)) ts[r0 == p0 ? 3 : 5] = (long long)__builtin_amdgcn_s_memtime();
            if (r0 != p0 && r0 + kRound < p1) issue_round(r0 + kRound, buf ^ 1);      // from the second round on: the next one goes into the buffer the previous round has left
            const char *pb = pbuf + buf * (kRound * 1024);
            // Round 6: a FULL round runs as straight-line code, two pairs at a time, the stores behind them.  The kernel is latency bound, not issue bound (2.9 M
            // wave instructions in 40 k cycles on 1024 SIMDs: one instruction per ~14 cycles and SIMD): a pair is one dependent chain -- record -> point reads ->
            // DPP means -> operand split -> swaps -> matrix instruction -> max tree -> swaps -> epilogue -- of ~1100 cycles, and with one conditional block per pair
            // the scheduler never sees two chains in one region.
            if (!kLab) {
#pragma unroll
                for (int k2 = 0; k2 < kRound; k2 += kGroup) {
                    if (k2 + kGroup - 1 < nr) {                     // kGroup pairs, one scheduling region
                        float yy[kGroup][2];
                        bool hb[kGroup];
#pragma unroll
                        for (int u = 0; u < kGroup; ++u) {
                            const int k = k2 + u, pair = r0 + k;
                            hb[u] = 2 * pair + 1 < a.M;
                            const float4 *rec = reinterpret_cast<const float4 *>(meta + (2 * (pair - p0) + half) * 32);
                            const float4 rec0 = rec[0];
                            const float ctr_z = rec[1].x;
                            const int np_eff = __builtin_bit_cast(int, rec0.x);
                            float4 q = *reinterpret_cast<const float4 *>(pb + k * 1024 + lane * 16);
                            const float4 qs = *reinterpret_cast<const float4 *>(pb + k * 1024 + (col >= np_eff ? half * 512 : lane * 16));
                            if (a.P < 32 && col >= a.P) q = make_float4(0.f, 0.f, 0.f, 0.f);
                            f32_pair_half<ABS>(a, fc, q, qs, np_eff, rec0.y, rec0.z, rec0.w, ctr_z, yy[u]);
                            swap32(yy[u][0], yy[u][1]);
                        }
#pragma unroll
                        for (int u = 0; u < kGroup; ++u) {
                            const int pair = r0 + k2 + u;
                            if (lane < a.C) coalign::store_stream(reinterpret_cast<float *>(feat_b + ((unsigned)(2 * pair) * (unsigned)a.C + (unsigned)lane) * 4u), yy[u][0]);
                            if (hb[u] && lane < a.C) coalign::store_stream(reinterpret_cast<float *>(feat_b + ((unsigned)(2 * pair + 1) * (unsigned)a.C + (unsigned)lane) * 4u), yy[u][1]);
                        }
                    } else {                                        // what is left of a run: one pair at a time
#pragma unroll
                      for (int k = k2; k < k2 + kGroup - 1; ++k) {
                        if (k >= nr) break;
                        const int pair = r0 + k;
                        const bool hasB = 2 * pair + 1 < a.M;
                        const float4 *rec = reinterpret_cast<const float4 *>(meta + (2 * (pair - p0) + half) * 32);
                        const float4 rec0 = rec[0];
                        const float ctr_z = rec[1].x;
                        const int np_eff = __builtin_bit_cast(int, rec0.x);
                        float4 q = *reinterpret_cast<const float4 *>(pb + k * 1024 + lane * 16);
                        const float4 qs = *reinterpret_cast<const float4 *>(pb + k * 1024 + (col >= np_eff ? half * 512 : lane * 16));
                        if (a.P < 32 && col >= a.P) q = make_float4(0.f, 0.f, 0.f, 0.f);
                        float y[2];
                        f32_pair_half<ABS>(a, fc, q, qs, np_eff, rec0.y, rec0.z, rec0.w, ctr_z, y);
                        swap32(y[0], y[1]);
                        if (lane < a.C) coalign::store_stream(reinterpret_cast<float *>(feat_b + ((unsigned)(2 * pair) * (unsigned)a.C + (unsigned)lane) * 4u), y[0]);
                        if (hasB && lane < a.C) coalign::store_stream(reinterpret_cast<float *>(feat_b + ((unsigned)(2 * pair + 1) * (unsigned)a.C + (unsigned)lane) * 4u), y[1]);
                      }
                    }
                }
                continue;
            }
#pragma unroll
            for (int k = 0; k < kRound; ++k) {
                if (k < nr) {
                    const int pair = r0 + k;
                    const bool hasB = 2 * pair + 1 < a.M;
                    const float4 *rec = reinterpret_cast<const float4 *>(meta + (2 * (pair - p0) + half) * 32);
                    const float4 rec0 = rec[0];
                    const float ctr_z = rec[1].x;
                    const int np_eff = __builtin_bit_cast(int, rec0.x);
                    float4 q = *reinterpret_cast<const float4 *>(pb + k * 1024 + lane * 16);
                    // rows at / past the point count read row 0 of their pillar
                    float4 qs = *reinterpret_cast<const float4 *>(pb + k * 1024 + (col >= np_eff ? half * 512 : lane * 16));
                    if (a.P < 32 && col >= a.P) q = make_float4(0.f, 0.f, 0.f, 0.f);
                    float y[2];
                    f32_pair_half<ABS>(a, fc, q, qs, np_eff, rec0.y, rec0.z, rec0.w, ctr_z, y);
                    // half layout -> lane = channel: [A ch 0-31 | B ch 0-31], [A ch 32-63 | B ch 32-63] -> [A 0-63], [B 0-63]: 256-byte row stores
                    swap32(y[0], y[1]);
                    if (kLab && (a.debug & 8)) continue;
                    if (lane < a.C) coalign::store_stream(reinterpret_cast<float *>(feat_b + ((unsigned)(2 * pair) * (unsigned)a.C + (unsigned)lane) * 4u), y[0]);      // (2 pair < M: p1 ends at the last pair)
                    if (hasB && lane < a.C) coalign::store_stream(reinterpret_cast<float *>(feat_b + ((unsigned)(2 * pair + 1) * (unsigned)a.C + (unsigned)lane) * 4u), y[1]);
                }
            }
        }
    }
    // the last workgroup to arrive publishes the tag: every stamp of this frame has been entered by then
#ifdef COALIGN_LAB
    if ((a.debug & 512) && threadIdx.x == 0 && blockIdx.x < 2048) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        g_sparse_trace[blockIdx.x * 4 + 0] = ts[7];
        g_sparse_trace[blockIdx.x * 4 + 1] = (long long)wall_clock64();
        g_sparse_trace[blockIdx.x * 4 + 2] = (long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) | ((long long)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) << 32);      // HW_ID | XCC_ID << 32
        g_sparse_trace[blockIdx.x * 4 + 3] = ts[5] - ts[3];
    }
#endif
    // (No fence: nobody reads the tag inside this launch -- the consumers are later launches on the stream, and a launch boundary publishes every store.  The
    //  counter only has to order "every workgroup has read state[0]" before "state[0] changes", which arrival itself does.  An agent-scope fence per workgroup
    //  here wrote back the XCD's dirty L2 lines -- the feature rows just stored -- 500 times: 55 us instead of 12.)
    // Round 6: the arrival is counted in TWO levels.  One counter for all workgroups cost 3.8 of the launch's 18.3 us (768 returning device-scope atomics on one
    // address are served one after the other, the last workgroups to finish queue behind each other: profiles/round6/experiments/pillar_sparse_ablations.txt).
    // Now workgroup b arrives at group counter b % kArriveGroups (64 bytes apart: different channels), the last of a group at the top counter, the last of those
    // publishes the tag: at most gridDim.x / 32 + 32 atomics per address.  Every counter is left at zero for the next launch.
    __syncthreads();
    if (threadIdx.x == 0 && !(kLab && (a.debug & 4))) {
        const int grp = (int)(blockIdx.x % kArriveGroups);
        const int members = ((int)gridDim.x - grp + kArriveGroups - 1) / kArriveGroups;          // blocks b < gridDim.x with b % kArriveGroups == grp
        const int live_groups = (int)gridDim.x < kArriveGroups ? (int)gridDim.x : kArriveGroups;
        int *gc = a.state + kArriveBase + grp * kArriveStride;
        if (__hip_atomic_fetch_add(gc, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == members - 1) {
            __hip_atomic_store(gc, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__hip_atomic_fetch_add(a.state + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == live_groups - 1) {
                __hip_atomic_store(a.state + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(a.state, (int)tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

}  // namespace

extern "C" size_t coalign_sparse_canvas_state_bytes(void) { return (size_t)(kArriveBase + kArriveGroups * kArriveStride) * sizeof(int32_t); }

extern "C" size_t coalign_sparse_canvas_stamp_bytes(int n_agents, int ny, int nx) {
    if (n_agents <= 0 || ny <= 0 || nx <= 0) return 0;
    return (size_t)n_agents * ny * nx * 8;
}

static int fill_sparse_params(SparseArgs &a, int P, const float *pfn_weight, const float *pfn_bias, const float *bn_weight, const float *bn_bias, const float *bn_mean,
                              const float *bn_var, float bn_eps, int C, int use_absolute_xyz) {
    if (P <= 0 || P > 32 || C < 1 || C > 64) return COALIGN_ERR_BAD_SHAPE;
    if (!pfn_weight) return COALIGN_ERR_NULL_POINTER;
    const bool has_bn = bn_weight || bn_bias || bn_mean || bn_var;
    if (has_bn && !(bn_weight && bn_bias && bn_mean && bn_var)) return COALIGN_ERR_NULL_POINTER;
    a.P = P;
    a.weight = pfn_weight; a.bias = pfn_bias; a.bn_w = bn_weight; a.bn_b = bn_bias; a.bn_m = bn_mean; a.bn_v = bn_var;
    a.eps = bn_eps; a.C = C; a.Cin = (use_absolute_xyz ? 4 : 1) + 6; a.use_abs = use_absolute_xyz;
    return COALIGN_OK;
}

extern "C" size_t coalign_pillar_folded_param_bytes(void) { return kFoldVecs * 64 * sizeof(float4); }

// Once per weight set: the encoder's channel parameters in the form the pair loop uses (BatchNorm folded, the three weight groups summed, signs folded), one
// record per lane of a wavefront.  coalign_pillar_encode_sparse then reads 96 bytes per lane instead of 30 scalars + a square root and a division per channel.
extern "C" int coalign_pillar_fold_params(const float *pfn_weight, const float *pfn_bias, const float *bn_weight, const float *bn_bias, const float *bn_mean,
                                          const float *bn_var, float bn_eps, int C, int use_absolute_xyz, float *folded, void *stream_) {
    using namespace coalign;
    if (!folded || (reinterpret_cast<uintptr_t>(folded) & 15)) return folded ? COALIGN_ERR_UNSUPPORTED : COALIGN_ERR_NULL_POINTER;
    SparseArgs a{};
    const int rc = fill_sparse_params(a, 32, pfn_weight, pfn_bias, bn_weight, bn_bias, bn_mean, bn_var, bn_eps, C, use_absolute_xyz);
    if (rc != COALIGN_OK) return rc;
    hipStream_t stream = (hipStream_t)stream_;
    if (use_absolute_xyz) hipLaunchKernelGGL(pillar_fold_kernel<true>, dim3(1), dim3(64), 0, stream, a, reinterpret_cast<float4 *>(folded));
    else hipLaunchKernelGGL(pillar_fold_kernel<false>, dim3(1), dim3(64), 0, stream, a, reinterpret_cast<float4 *>(folded));
    return check_launch();
}

namespace {
int encode_sparse(const coalign_pillar_frame *frame, const float *voxel_features, const int32_t *voxel_num_points, const int32_t *voxel_coords, int M_capacity,
                  const int32_t *M_dev, int P, const float *folded, int C, int use_absolute_xyz, const double *voxel_size, const double *range_min, int n_agents, int ny,
                  int nx, float *pillar_features, void *stamps, int32_t *state, void *stream_) {
    using namespace coalign;
    hipStream_t stream = (hipStream_t)stream_;
    if (M_capacity < 0 || n_agents <= 0 || ny <= 0 || nx <= 0 || P <= 0 || P > 32 || C < 1 || C > 64) return COALIGN_ERR_BAD_SHAPE;
    if (!folded || !voxel_size || !range_min || !stamps || !state) return COALIGN_ERR_NULL_POINTER;
    if (M_capacity > 0 && ((!frame && (!voxel_features || !voxel_num_points || !voxel_coords)) || !pillar_features)) return COALIGN_ERR_NULL_POINTER;
    if (frame && (reinterpret_cast<uintptr_t>(frame) & 7)) return COALIGN_ERR_BAD_SHAPE;
    if ((size_t)n_agents * ny * nx > (size_t)INT32_MAX || (reinterpret_cast<uintptr_t>(stamps) & 7) || (reinterpret_cast<uintptr_t>(folded) & 15)) return COALIGN_ERR_BAD_SHAPE;
    if ((size_t)M_capacity * P * 16 >= ((size_t)1 << 32) || (size_t)M_capacity * C * 4 >= ((size_t)1 << 32)) return COALIGN_ERR_UNSUPPORTED;      // 32-bit byte offsets
    SparseArgs a{};
    a.pts = (const float4 *)voxel_features; a.npts = voxel_num_points; a.coords = (const int4 *)voxel_coords;
    a.M = M_capacity; a.P = P; a.C = C; a.use_abs = use_absolute_xyz;
    a.folded = reinterpret_cast<const float4 *>(folded);
    a.vx = (float)voxel_size[0]; a.vy = (float)voxel_size[1]; a.vz = (float)voxel_size[2];
    a.xo = (float)(voxel_size[0] / 2 + range_min[0]);
    a.yo = (float)(voxel_size[1] / 2 + range_min[1]);
    a.zo = (float)(voxel_size[2] / 2 + range_min[2]);
    a.n_agents = n_agents; a.ny = ny; a.nx = nx; a.feats = pillar_features;
    a.stamps = static_cast<unsigned long long *>(stamps); a.state = state; a.M_dev = M_dev; a.frame = frame;
    a.debug = coalign::lab_env("COALIGN_SPARSE_DEBUG", 0);
    // Grid = a WHOLE number of workgroups per CU (the dispatcher spreads resident workgroups evenly: measured 3 or 4 per CU for 834 workgroups, and the CUs with
    // 4 finish 3.5 us after those with 3 -- the kernel is bound by each SIMD's issue, so the launch ends with the busiest SIMD), pairs shared out evenly
    // (+-1): k = 1..3 workgroups per CU (40 KB of LDS each), about kPairsPerWave pairs per wavefront; larger inputs: runs of at most kRunPairs pairs, more rounds
    // of workgroups.
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    const int pairs = (M_capacity + 1) / 2;
    const int per_wave_target = coalign::lab_env("COALIGN_SPARSE_PAIRS", kPairsPerWave);      // laboratory build: pairs per wavefront the grid is sized for
    const int unit = cus * kWaves * per_wave_target;
    int k = (pairs + unit / 2) / unit;
    k = k < 1 ? 1 : k > 3 ? 3 : k;       // (four per CU fit on paper -- 128 KB of LDS -- but a grid that needs EVERY slot waits a whole workgroup lifetime for a straggler: 31 vs 20 us)
    int blocks = k * cus;
    if (pairs < blocks * kWaves) blocks = (pairs + kWaves - 1) / kWaves;                   // small inputs: one pair per wavefront
    const int need = (pairs + kWaves * kRunPairs - 1) / (kWaves * kRunPairs);
    if (blocks < need) blocks = (need + cus - 1) / cus * cus;
    if (blocks < 1) blocks = 1;                               // (M = 0: the launch still advances the frame tag)
    if (use_absolute_xyz) hipLaunchKernelGGL(pillar_sparse_kernel<true>, dim3(blocks), dim3(kWaves * 64), 0, stream, a);
    else hipLaunchKernelGGL(pillar_sparse_kernel<false>, dim3(blocks), dim3(kWaves * 64), 0, stream, a);
    return check_launch();
}
}  // namespace

extern "C" int coalign_pillar_encode_sparse(const float *voxel_features, const int32_t *voxel_num_points, const int32_t *voxel_coords, int M_capacity,
                                            const int32_t *M_dev, int P, const float *folded, int C, int use_absolute_xyz, const double *voxel_size,
                                            const double *range_min, int n_agents, int ny, int nx, float *pillar_features, void *stamps, int32_t *state,
                                            void *stream_) {
    return encode_sparse(nullptr, voxel_features, voxel_num_points, voxel_coords, M_capacity, M_dev, P, folded, C, use_absolute_xyz, voxel_size, range_min, n_agents, ny, nx,
                         pillar_features, stamps, state, stream_);
}

extern "C" int coalign_pillar_encode_sparse_frame(const coalign_pillar_frame *frame, int M_capacity, int P, const float *folded, int C, int use_absolute_xyz,
                                                  const double *voxel_size, const double *range_min, int n_agents, int ny, int nx, float *pillar_features,
                                                  void *stamps, int32_t *state, void *stream_) {
    if (!frame) return COALIGN_ERR_NULL_POINTER;
    return encode_sparse(frame, nullptr, nullptr, nullptr, M_capacity, nullptr, P, folded, C, use_absolute_xyz, voxel_size, range_min, n_agents, ny, nx, pillar_features,
                         stamps, state, stream_);
}

#ifdef COALIGN_LAB
extern "C" int coalign_lab_sparse_trace(long long *host_dst) {      // laboratory build only: copy the per-workgroup trace out (synchronises the device)
    return coalign::hip_call(hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_sparse_trace), sizeof(long long) * 2048 * 4));
}
#endif
