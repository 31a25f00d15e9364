// 3x3 / pad 1 / stride 1 convolution with fused bias (+ residual) (+ ReLU) as Winograd F(2x2, 3x3) on the BF16 matrix cores, fp32 operands
// by the same error-free 3-way bf16 split as conv3x3_emu.hip, fp32 accumulation, channels-last in and out, gfx950.
//
// Layers: the stride-1 3x3 convolutions of the ResNet stages and of the shrink header
// (opencood/models/sub_modules/resblock.py:53-69, base_bev_backbone_resnet.py:59-138, downsample_conv.py:7-50).
//
// Why: every convolution layer of the detector runs at the board's power limit and the frame rate follows the number of executed matrix
// instructions.  F(2x2, 3x3) evaluates a 2x2 output tile from a 4x4 input patch with 16 products per (cin, cout)
// instead of 36:
//     Y = A^T [ (G g G^T) .* (B^T d B) ] A,      B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1],   A^T = [1 1 1 0; 0 1 -1 -1],
//     G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]
// i.e. 16 independent GEMMs over cin, one per position p = (i, j) of the 4x4 transform domain:  M_p[cout, tile] = sum_cin U_p[cout, cin] V_p[cin, tile].
//   * U = G g G^T is computed on the host in float64 from the BatchNorm-folded fp32 weights and split into three bf16 terms (ops.pack_conv3x3_wino_weight);
//   * V = B^T d B is computed in fp32 (additions only) from the fp32 activations while the patch is staged, and split AFTER the transform
//     (v = v_h + v_m + v_l exactly);
//   * each product U V is the six cross terms down to 2^-16 |u v| on v_mfma_f32_32x32x16_bf16 (what is dropped is <= 2^-24 |u v|), fp32 accumulators;
//   * the output transform, bias, residual and ReLU run in the epilogue.
//
// Workgroup = 8 wavefronts = one unit of 64 output channels x 64 Winograd tiles (a block of TBH x TBW tiles = 2 TBH x 2 TBW output pixels of
// the batch stacked as one tall image: `pitch` rows per image, the rows H .. pitch - 1 are the zero padding between two images).
// The K loop runs over 16-channel steps, each cut in two HALF STAGES h = 0, 1 that cover the transform columns j in {2h, 2h + 1}:
//   consumer role of wavefront w: transform row i = w & 3, output-channel half c = w >> 2: positions (i, 0..3) x 32 couts x 64 tiles = 8 accumulator
//     tiles of 32 x 32 (128 registers).  Its A operands (U) are private to it: six 1 KB loads per half stage straight from L2 into registers, one half
//     stage ahead (the image is stored in exactly that order).  Its B operands (V) come from LDS, one ds_read_b128 each.
//   producer role of wavefront w: transform row i = w & 3, channel group g = w >> 2 (8 of the 16 channels), lane = tile: reads the two raw rows and
//     three raw columns its two positions need from the fp32 patch in LDS, transforms (5 additions per 2 values), splits, and writes 6 ready B operands.
//   The two wavefronts that share a SIMD (w, w + 4) run the two roles in opposite order, so one feeds the matrix pipe while the other transforms.
// LDS: two V buffers of 48 KB ([jj][i][term][channel group][tile] x 16 B), two raw fp32 patches ([4-channel chunk][column parity][row][column / 2] x 16 B:
// tiles that are neighbours in x read neighbouring 16-byte slots), 150 KB in all; the epilogue reuses it for the cross-wavefront half of the output transform.
#include "common.h"
#include <cstdlib>
#include <type_traits>

#ifndef WINO_SWAP_COND
#define WINO_SWAP_COND (wc == 0)
#endif

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void *lptr_w_t;

constexpr int kCoutUnit = 64;        // output channels per workgroup
constexpr int kTiles = 64;           // Winograd tiles per workgroup
constexpr int kVBuf = 2 * 4 * 3 * 2 * kTiles;      // uint4 per V buffer: [jj][i][term][g][tile]

struct WinoArgs {
    const float *__restrict__ x;          // [N][H][W][Cin]
    const uint4 *__restrict__ u;          // [Cout / 64][Cin / 16][2 h][8 waves][2 jj][3 terms][64 lanes] x 16 B
    const float *__restrict__ bias;       // [Cout]
    const float *__restrict__ residual;   // [N][H][W][Cout] or null
    float *__restrict__ y;                // [N][H][W][Cout]
    int N, Cin, Cout, H, W, relu;
    int pitch;                            // rows per image in the stacked image (H + 1 for odd H, H + 2 for even H)
    int blocks_x, blocks, units, xcd;
    const float *zero16;                  // 16 zero bytes in global memory (the tail of the weight image): the DMA source of padding
};

template <int TBW>
struct WGeo {
    static_assert(TBW == 8 || TBW == 16, "tile blocks of 8 x 8 or 4 x 16 tiles");
    static constexpr int TBH = kTiles / TBW;
    static constexpr int PR = 2 * TBH + 2, PC = 2 * TBW + 2;        // raw patch rows / columns
    static constexpr int PCH = TBW == 16 ? 24 : 12;                 // column pairs per row, padded so that the 16 lanes one ds_read_b128 cycle serves ({0-3, 12-15, 20-27}, ...) hit 64 distinct banks: tile rows 256 B (TBW 16) / 128 B (TBW 8) apart mod 256
    static constexpr int RAWQ = 2 * PR * PCH;                       // uint4 per 4-channel chunk plane: [parity][row][column / 2]
    static constexpr int RAWSZ = 4 * RAWQ;                          // uint4 per raw buffer
    static constexpr int NU = PR * PC * 4;                          // 16-byte units of one patch, pixel-major, chunk fastest
    static constexpr int NJ = (NU + 511) / 512;
    static constexpr int ND = (RAWSZ + 511) / 512;                  // LDS-DMA instructions per wavefront and K step (64 slots each, 8 wavefronts)
    static constexpr int V_OFF = 0, R_OFF = 2 * kVBuf;
    static constexpr int ZSTRIDE = 68;                              // floats per (row, b, tile) in the epilogue exchange: 64 couts + 4 of bank shift
    static constexpr size_t LDS_MAIN = (size_t)(2 * kVBuf + 2 * RAWSZ) * 16, LDS_EPI = (size_t)4 * 2 * kTiles * ZSTRIDE * 4;
    static constexpr size_t LDS_BYTES = LDS_MAIN > LDS_EPI ? LDS_MAIN : LDS_EPI;
};

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float floatx2 __attribute__((ext_vector_type(2)));

// error-free 3-way split of 8 fp32 values: out[t] = term t of each, 8 bf16 = one matrix operand.  Pairs go through v_cvt_pk_bf16_f32; the
// fp32 value of a term is the packed word shifted / masked (no second conversion).
__device__ __forceinline__ void split8(const float (&v)[8], uint4 (&out)[3]) {
    unsigned o[3][4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float a = v[2 * e], b = v[2 * e + 1];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const unsigned w = __builtin_bit_cast(unsigned, __builtin_convertvector(floatx2{a, b}, bf16x2));
            o[t][e] = w;
            if (t < 2) {
                a -= __builtin_bit_cast(float, w << 16);              // exact
                b -= __builtin_bit_cast(float, w & 0xffff0000u);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 3; ++t) out[t] = uint4{o[t][0], o[t][1], o[t][2], o[t][3]};
}

// workgroup barrier of a half stage: this wavefront's LDS writes AND its LDS-DMA transfers (vmcnt) are complete.  (The operand prefetch issued in the same half
// stage is needed right behind the barrier anyway.)
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// ABL (lab builds only, -DCOALIGN_LAB): bit 0 = no matrix steps, bit 1 = no transform / split, bit 2 = operands of half stage 0 only (wrong results; what each part costs)
template <int TBW, int ABL = 0, int MODE = 0>
__global__ __launch_bounds__(512) void conv3x3_wino_kernel(const WinoArgs a) {
    using G = WGeo<TBW>;
    extern __shared__ __attribute__((aligned(1024))) uint4 lds[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int wi = wave & 3, wc = wave >> 2;              // consumer: transform row, cout half; producer: transform row, channel group
    const int K = a.Cin >> 4, groups = a.Cout / kCoutUnit;
    // producer constants: the raw rows of R[i] = d[ra] + sg d[rb]  (B^T rows: d0 - d2, d1 + d2, d2 - d1, d1 - d3)
    const int ra = wi == 0 ? 0 : wi == 2 ? 2 : 1, rb = wi == 0 ? 2 : wi == 1 ? 2 : wi == 2 ? 1 : 3;
    const float sg = wi == 1 ? 1.f : -1.f;
    const int pty = lane / TBW, ptx = lane % TBW;         // producer: this lane's tile inside the block
    // raw slot of patch pixel (2 pty + r, 2 ptx + c) for chunk q: ((q * 2 + (c & 1)) * PR + 2 pty + r) * PCH + ptx + (c >> 1)
    const int praw = (2 * pty) * G::PCH + ptx;
    const int vwr = (wi * 3) * 2 * kTiles + wc * kTiles + lane;                        // + jj * 4 * 3 * 2 * 64 + term * 128: this lane's B-operand slot as a producer
    const int vrd = (wi * 3) * 2 * kTiles + (lane >> 5) * kTiles + (lane & 31);        // + jj * ... + term * 128 + 32 tb: as a consumer

    // Which of the two wavefronts of a SIMD transforms first and which runs its matrix steps first: decided per SIMD (hardware SIMD id + one LDS
    // counter each), not from the wavefront index -- the dispatcher's wavefront -> SIMD assignment is not w % 4.
    __shared__ int simd_count[4];
    if (tid < 4) simd_count[tid] = 0;
    __syncthreads();
    int first = 0;
    if (lane == 0) first = atomicAdd(&simd_count[(__builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4)) & 3], 1);      // HW_REG_HW_ID bits 5:4 = SIMD_ID
    first = __builtin_amdgcn_readfirstlane(first) & 1;
    int g = blockIdx.x;
    if (a.xcd) g = coalign::xcd_remap(g, gridDim.x);       // neighbouring units (same tiles, other couts; same couts, next tiles) share an L2
    for (int unit = g; unit < a.units; unit += gridDim.x) {
        const int cg = unit % groups, blk = unit / groups;
        const int by = blk / a.blocks_x, bx = blk - by * a.blocks_x;
        const int ty0 = by * G::TBH, tx0 = bx * TBW;      // first tile of the block (stacked tile rows)
        // ---- raw patch by LDS-DMA (global_load_lds_dwordx4: 64 lanes x 16 B land at 64 consecutive LDS slots, no registers, no ds_write, no bank conflicts).
        // Slot s of a raw buffer = [q][parity][row][column pair] in LDS order; DMA instruction d of this wavefront covers slots 64 (wave + 8 d) ...; every lane
        // computes the global address of ITS slot's pixel chunk once per unit (padding slots and pixels outside the image read a zero word).
        int goff[G::ND];
#pragma unroll
        for (int d = 0; d < G::ND; ++d) {
            const int sl = 64 * (wave + 8 * d) + lane;
            const int ch = sl % G::PCH, t1 = sl / G::PCH, pr = t1 % G::PR, t2 = t1 / G::PR, par = t2 & 1, q = t2 >> 1;
            const int pc = 2 * ch + par;
            const int st = 2 * ty0 - 1 + pr, xg = 2 * tx0 - 1 + pc;
            const int n = st >= 0 ? st / a.pitch : 0, r = st - n * a.pitch;
            const bool ok = sl < G::RAWSZ && pc < G::PC && st >= 0 && n < a.N && r < a.H && xg >= 0 && xg < a.W;
            goff[d] = ok ? ((n * a.H + r) * a.W + xg) * a.Cin + 4 * q : -1;
        }
        auto dma_raw = [&](int k, int buf) {
#pragma unroll
            for (int d = 0; d < G::ND; ++d) {
                if (64 * (wave + 8 * d) < G::RAWSZ) {      // (wave-uniform)
                    const float *src = goff[d] < 0 ? a.zero16 : a.x + goff[d] + 16 * k;
                    const unsigned dst = (unsigned)(size_t)(lptr_w_t)(lds + G::R_OFF + buf * G::RAWSZ + 64 * (wave + 8 * d));
                    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(__builtin_amdgcn_readfirstlane(dst)), "v"(src) : "memory", "m0");
                }
            }
        };
        const uint4 *ubase = a.u + ((size_t)cg * K * 2 * 8 + wave) * (2 * 3 * 64) + lane;        // + (k * 2 + h) * 8 * 384 + (jj * 3 + term) * 64
        auto load_a = [&](int k, int h, uint4 (&av)[2][3]) {
            if (ABL & 4) k = 0;                            // (lab) every half stage re-reads the first one: L1-resident operands
            const uint4 *p = ubase + (size_t)(k * 2 + h) * (8 * 2 * 3 * 64);
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int t = 0; t < 3; ++t) av[jj][t] = p[(jj * 3 + t) * 64];
        };
        // producer: half stage (k, h) -> V buffer vb, from raw buffer rbi
        auto produce = [&](auto hc, int rbi, int vb) {
            constexpr int h = decltype(hc)::value;
            if (ABL & 2) return;
            const uint4 *rbuf = lds + G::R_OFF + rbi * G::RAWSZ + praw;
            float v0[8], v1[8];
#pragma unroll
            for (int q2 = 0; q2 < 2; ++q2) {
                const int q = 2 * wc + q2;
                float4 R[3];
#pragma unroll
                for (int ci = 0; ci < 3; ++ci) {
                    const int c = h + ci;                  // raw columns h .. h + 2
                    const int base = (q * 2 + (c & 1)) * G::PR * G::PCH + (c >> 1);
                    const float4 da = __builtin_bit_cast(float4, rbuf[base + ra * G::PCH]);
                    const float4 db = __builtin_bit_cast(float4, rbuf[base + rb * G::PCH]);
                    R[ci].x = fmaf(sg, db.x, da.x); R[ci].y = fmaf(sg, db.y, da.y); R[ci].z = fmaf(sg, db.z, da.z); R[ci].w = fmaf(sg, db.w, da.w);
                }
                // h = 0: V[i][0] = R0 - R2, V[i][1] = R1 + R2 (columns 0 1 2);  h = 1: V[i][2] = R2 - R1, V[i][3] = R1 - R3 (columns 1 2 3 -> R[0..2])
                float4 p0, p1;
                if (h == 0) {
                    p0 = float4{R[0].x - R[2].x, R[0].y - R[2].y, R[0].z - R[2].z, R[0].w - R[2].w};
                    p1 = float4{R[1].x + R[2].x, R[1].y + R[2].y, R[1].z + R[2].z, R[1].w + R[2].w};
                } else {
                    p0 = float4{R[1].x - R[0].x, R[1].y - R[0].y, R[1].z - R[0].z, R[1].w - R[0].w};
                    p1 = float4{R[0].x - R[2].x, R[0].y - R[2].y, R[0].z - R[2].z, R[0].w - R[2].w};
                }
                v0[4 * q2] = p0.x; v0[4 * q2 + 1] = p0.y; v0[4 * q2 + 2] = p0.z; v0[4 * q2 + 3] = p0.w;
                v1[4 * q2] = p1.x; v1[4 * q2 + 1] = p1.y; v1[4 * q2 + 2] = p1.z; v1[4 * q2 + 3] = p1.w;
            }
            uint4 s0[3], s1[3];
            split8(v0, s0);
            split8(v1, s1);
            uint4 *vbuf = lds + G::V_OFF + vb * kVBuf + vwr;
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                vbuf[t * 2 * kTiles] = s0[t];
                vbuf[4 * 3 * 2 * kTiles + t * 2 * kTiles] = s1[t];
            }
        };

        floatx16 acc[4][2];
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int tb = 0; tb < 2; ++tb) acc[p][tb] = floatx16{0};
        auto consume = [&](auto hc, int vb, const uint4 (&av)[2][3]) {
            constexpr int h = decltype(hc)::value;
            if (ABL & 1) return;
            const uint4 *vbuf = lds + G::V_OFF + vb * kVBuf + vrd;
            constexpr int wt[6] = {0, 1, 2, 0, 1, 0}, bt[6] = {2, 1, 0, 1, 0, 0};      // smallest products first
            // (tiles below the stacked image are computed like the others and never stored: a liveness branch around the matrix steps makes hipcc
            //  copy the accumulators into temporaries and back -- 64 v_mov_b64 per half stage)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                uint4 b[2][3];
#pragma unroll
                for (int t = 0; t < 3; ++t)
#pragma unroll
                    for (int tb = 0; tb < 2; ++tb) b[tb][t] = vbuf[jj * 4 * 3 * 2 * kTiles + t * 2 * kTiles + 32 * tb];
#pragma unroll
                for (int i = 0; i < 6; ++i)
#pragma unroll
                    for (int tb = 0; tb < 2; ++tb)         // two independent accumulator chains
                        acc[2 * h + jj][tb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av[jj][wt[i]]), __builtin_bit_cast(bf16x8, b[tb][bt[i]]),
                                                                                      acc[2 * h + jj][tb], 0, 0, 0);
            }
        };

        // MODE 2: the interleave spelled out for the scheduler -- per matrix instruction one LDS read and five VALU instructions, an LDS write every fourth
        auto interleave = [&]() {
            if constexpr (MODE == 2) {
#pragma unroll
                for (int i = 0; i < 24; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // MFMA
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // DS read
                    __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);      // VALU
                    if (i % 4 == 3) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);      // DS write
                }
            }
        };
        constexpr std::integral_constant<int, 0> H0{};
        constexpr std::integral_constant<int, 1> H1{};
        // ---- prologue
        uint4 a0[2][3], a1[2][3];
        dma_raw(0, 0);
        load_a(0, 0, a0);
        lds_barrier();                                     // (waits for the DMA: vmcnt(0))
        produce(H0, 0, 0);
        // ---- K loop: two half stages per 16 channels; everything one half stage ahead, one barrier per half stage
        for (int k = 0; k < K; ++k) {
            lds_barrier();                                 // V(k, 0) complete; V buffer 1 and raw buffer (k + 1) & 1 free
            long long tq[4] = {0, 0, 0, 0};
            if constexpr ((ABL & 8) != 0) tq[0] = (long long)__builtin_amdgcn_s_memtime();
            if (k + 1 < K) dma_raw(k + 1, (k + 1) & 1);      // raw(k + 1): needed by the transform of half stage (k, 1), i.e. behind the next barrier; its buffer was last read in (k - 1, 0)
            load_a(k, 1, a1);
            if constexpr (MODE == 0) {
#pragma nounroll
                for (int sub = 0; sub < 2; ++sub) {        // the two wavefronts of a SIMD: one transforms while the other runs its matrix steps
                    if (sub == first) consume(H0, 0, a0);
                    else produce(H1, k & 1, 1);
                    if constexpr ((ABL & 8) != 0) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); tq[1 + sub] = (long long)__builtin_amdgcn_s_memtime(); }
                }
                if constexpr ((ABL & 8) != 0) {
                    if (k == 3 && unit == g && (blockIdx.x == 0 || blockIdx.x == 77) && lane == 0)
                        printf("wg %d wave %d simd %d first %d: barrier exit %lld, role 1 done +%lld, role 2 done +%lld (role 1 = %s)\n", (int)blockIdx.x, wave,
                               (int)(__builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4) & 3), first, tq[0] % 1000000, tq[1] - tq[0], tq[2] - tq[0], first == 0 ? "matrix steps" : "transform");
                }
            } else {                                       // one basic block: the matrix steps and the transform interleaved inside every wavefront
                consume(H0, 0, a0);
                produce(H1, k & 1, 1);
                interleave();
            }
            lds_barrier();                                 // V(k, 1) and raw(k + 1) complete; V buffer 0 free
            if (k + 1 < K) load_a(k + 1, 0, a0);
            if constexpr (MODE == 0) {
#pragma nounroll
                for (int sub = 0; sub < 2; ++sub) {
                    if (sub == first) consume(H1, 1, a1);
                    else if (k + 1 < K) produce(H0, (k + 1) & 1, 0);
                }
            } else {
                consume(H1, 1, a1);
                produce(H0, (k + 1) & 1, 0);               // (after the last step: transforms a stale patch into a buffer nobody reads)
                interleave();
            }
        }
        // ---- epilogue.  Output transform, column half inside the wavefront: Z[i][0] = M[i][0] + M[i][1] + M[i][2], Z[i][1] = M[i][1] - M[i][2] - M[i][3]
        __syncthreads();                                   // every wavefront is done with the V / raw buffers
        float *zf = reinterpret_cast<float *>(lds);
#pragma unroll
        for (int tb = 0; tb < 2; ++tb) {
            const int tile = 32 * tb + (lane & 31);
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                float4 z0, z1;
                float *o0 = &z0.x, *o1 = &z1.x;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * rg + e;
                    o0[e] = (acc[0][tb][r] + acc[1][tb][r]) + acc[2][tb][r];
                    o1[e] = (acc[1][tb][r] - acc[2][tb][r]) - acc[3][tb][r];
                }
                const int cout = 32 * wc + 8 * rg + 4 * (lane >> 5);          // accumulator register r = 4 rg + e holds cout 8 rg + 4 (lane >> 5) + e
                *reinterpret_cast<float4 *>(zf + ((wi * 2 + 0) * kTiles + tile) * G::ZSTRIDE + cout) = z0;
                *reinterpret_cast<float4 *>(zf + ((wi * 2 + 1) * kTiles + tile) * G::ZSTRIDE + cout) = z1;
            }
        }
        __syncthreads();
        // row half across the wavefronts: Y[0][b] = Z[0][b] + Z[1][b] + Z[2][b], Y[1][b] = Z[1][b] - Z[2][b] - Z[3][b]; item = (a, b, tile, 4 couts),
        // 16 consecutive lanes = the 64 couts of one pixel (256 contiguous bytes)
        {
            const int cq = tid & 15;
            const float4 bias = *reinterpret_cast<const float4 *>(a.bias + cg * kCoutUnit + 4 * cq);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int ab = j >> 1, oa = ab >> 1, ob = ab & 1;
                const int tile = (tid >> 4) + 32 * (j & 1);
                const int ty = tile / TBW, tx = tile - ty * TBW;
                const int s = 2 * (ty0 + ty) + oa, ox = 2 * (tx0 + tx) + ob;
                const int n = s / a.pitch, oy = s - n * a.pitch;
                if (n < a.N && oy < a.H && ox < a.W) {
                    const float *zp = zf + (ob * kTiles + tile) * G::ZSTRIDE + 4 * cq;
                    const float4 z0 = *reinterpret_cast<const float4 *>(zp + (oa ? 1 : 0) * 2 * kTiles * G::ZSTRIDE);
                    const float4 z1 = *reinterpret_cast<const float4 *>(zp + (oa ? 2 : 1) * 2 * kTiles * G::ZSTRIDE);
                    const float4 z2 = *reinterpret_cast<const float4 *>(zp + (oa ? 3 : 2) * 2 * kTiles * G::ZSTRIDE);
                    float4 o;
                    if (oa == 0) o = float4{(z0.x + z1.x) + z2.x, (z0.y + z1.y) + z2.y, (z0.z + z1.z) + z2.z, (z0.w + z1.w) + z2.w};
                    else o = float4{(z0.x - z1.x) - z2.x, (z0.y - z1.y) - z2.y, (z0.z - z1.z) - z2.z, (z0.w - z1.w) - z2.w};
                    const size_t off = ((size_t)(n * a.H + oy) * a.W + ox) * a.Cout + cg * kCoutUnit + 4 * cq;
                    o.x += bias.x; o.y += bias.y; o.z += bias.z; o.w += bias.w;
                    if (a.residual) {
                        const float4 rsd = *reinterpret_cast<const float4 *>(a.residual + off);
                        o.x += rsd.x; o.y += rsd.y; o.z += rsd.z; o.w += rsd.w;
                    }
                    if (a.relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
                    *reinterpret_cast<float4 *>(a.y + off) = o;
                }
            }
        }
        __syncthreads();                                   // the next unit's prologue overwrites the exchange area
    }
}

template <int TBW, int ABL = 0, int MODE = 0>
int launch_wino(const WinoArgs &a0, hipStream_t s) {
    using G = WGeo<TBW>;
    static_assert(G::LDS_BYTES <= 160 * 1024, "geometry does not fit the 160 KB LDS");
    static int cus = 0;
    auto kern = conv3x3_wino_kernel<TBW, ABL, MODE>;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) prop.multiProcessorCount = 256;
        const int rc = coalign::hip_call(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES));
        if (rc != COALIGN_OK) {
            (void)hipGetLastError();
            return rc;
        }
        cus = prop.multiProcessorCount;
    }
    WinoArgs a = a0;
    a.pitch = a.H + ((a.H & 1) ? 1 : 2);
    const int tile_rows = a.N * a.pitch / 2, tile_cols = (a.W + 1) / 2;
    a.blocks_x = (tile_cols + TBW - 1) / TBW;
    a.blocks = a.blocks_x * ((tile_rows + G::TBH - 1) / G::TBH);
    a.units = a.blocks * (a.Cout / kCoutUnit);
    const int grid = a.units < cus ? a.units : cus;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), G::LDS_BYTES, s, a);
    return COALIGN_OK;
}

}  // namespace

extern "C" size_t coalign_conv3x3_wino_weight_bytes(int Cin, int Cout) {
    if (Cin < 16 || Cout < 64 || Cin % 16 || Cout % kCoutUnit) return 0;
    return (size_t)(Cout / kCoutUnit) * (Cin / 16) * 2 * 8 * 2 * 3 * 64 * 16 + 16;       // + 16 zero bytes: the LDS-DMA source of the zero padding
}

extern "C" int coalign_conv3x3_wino(const float *x, const void *u_split, const float *bias, const float *residual, float *y, int N, int Cin, int Cout,
                                    int H, int W, int relu, int tile_block_w, void *stream) {
    using namespace coalign;
    if (!x || !u_split || !bias || !y) return COALIGN_ERR_NULL_POINTER;
    if (N < 0 || H < 1 || W < 1 || Cin < 1 || Cout < 1) return COALIGN_ERR_BAD_SHAPE;
    if (Cin % 16 || Cout % kCoutUnit) return COALIGN_ERR_UNSUPPORTED;
    if ((int64_t)N * (H + 2) * W * (Cin > Cout ? Cin : Cout) >= (int64_t)1 << 31) return COALIGN_ERR_UNSUPPORTED;      // 32-bit pixel offsets
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(u_split) | reinterpret_cast<uintptr_t>(bias) |
         reinterpret_cast<uintptr_t>(residual)) & 15)
        return COALIGN_ERR_UNSUPPORTED;
    if (N == 0) return COALIGN_OK;
    WinoArgs a{x, static_cast<const uint4 *>(u_split), bias, residual, y, N, Cin, Cout, H, W, relu, 0, 0, 0, 0, 1,
               reinterpret_cast<const float *>(static_cast<const char *>(u_split) + coalign_conv3x3_wino_weight_bytes(Cin, Cout) - 16)};
    hipStream_t s = static_cast<hipStream_t>(stream);
    // block shape: 4 x 16 tiles (8 x 32 pixels) unless the map is narrow or 8 x 8 tiles waste fewer columns
    int tbw = tile_block_w;
    if (tbw == 0) {
        const int tc = (W + 1) / 2;
        const int waste16 = (tc + 15) / 16 * 16 - tc, waste8 = (tc + 7) / 8 * 8 - tc;
        tbw = waste8 < waste16 ? 8 : 16;
    }
    int rc;
#ifdef COALIGN_LAB
    static const int mode = getenv("COALIGN_WINO_MODE") ? atoi(getenv("COALIGN_WINO_MODE")) : 0;
    if (mode == 1 && tbw == 16) { rc = launch_wino<16, 0, 1>(a, s); return rc != COALIGN_OK ? rc : check_launch(); }
    if (mode == 2 && tbw == 16) { rc = launch_wino<16, 0, 2>(a, s); return rc != COALIGN_OK ? rc : check_launch(); }
    static const int abl = getenv("COALIGN_WINO_ABL") ? atoi(getenv("COALIGN_WINO_ABL")) : 0;
    if (abl) {
        switch (abl * 100 + tbw) {
            case 116: rc = launch_wino<16, 1>(a, s); break;
            case 216: rc = launch_wino<16, 2>(a, s); break;
            case 316: rc = launch_wino<16, 3>(a, s); break;
            case 416: rc = launch_wino<16, 4>(a, s); break;
            case 716: rc = launch_wino<16, 7>(a, s); break;
            case 816: rc = launch_wino<16, 8>(a, s); break;
            default: return COALIGN_ERR_UNSUPPORTED;
        }
        return rc != COALIGN_OK ? rc : check_launch();
    }
#endif
    if (tbw == 16) rc = launch_wino<16>(a, s);
    else if (tbw == 8) rc = launch_wino<8>(a, s);
    else return COALIGN_ERR_UNSUPPORTED;
    return rc != COALIGN_OK ? rc : check_launch();
}
