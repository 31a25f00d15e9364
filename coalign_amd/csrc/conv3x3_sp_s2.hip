// 3x3 / STRIDE 2 / pad 1 convolution whose input arrives ALREADY SPLIT (round 6), gfx950: the strided first convolution of a ResNet stage
// (opencood/models/sub_modules/resblock.py:53-69 with stride 2, 150-174; base_bev_backbone_resnet.py:59-119) in the form of conv3x3_sp.hip -- the K loop is
// LDS-DMA + ds_read_b128 + v_mfma_f32_32x32x16_f16, no VALU, one barrier per 16 input channels, both operands double buffered.  Rounds 4-5 ran these three
// layers on conv3x3_emu.hip's consumer-split kernel (float32 pixels loaded into registers, split on the VALU, written to LDS: ~7500 clocks per 8 input
// channels of which ~2200 are matrix steps, matrix pipe busy 0.11-0.15; profiles/round5/experiments/conv_strided_timeline.txt).
//
// Input, one of
//   * DENSE: an SP map [N][Cin / 16][4 planes][H][W][8 x fp16] (conv3x3_sp.hip; the previous stage's output), or
//   * SPARSE CANVAS (the first stage): sp16 feature ROWS [M][Cin / 16][4 planes][8 x fp16] (coalign_sp_pack_rows of pillar_sparse.hip's float32 rows) + the
//     8-byte cell stamps of pillar_sparse.hip: pixel (n, y, x) = row (stamp & 0xffffffff) if stamp >> 32 == *tag, else zero.  A lane of an LDS-DMA instruction
//     takes its 16 bytes from its own address, so the gather costs no instruction more than the dense form; the stamps of a tile (two per lane) are loaded one
//     interval before the tile's first DMA is issued.
// Output: an SP map [N][Cout / 16][4][Ho][Wo][8], Ho = ceil(H / 2), Wo = ceil(W / 2); bias, ReLU, range word as conv3x3_sp.hip.
// SKIP (round 6): the block's 1 x 1 / stride-2 down-sampling convolution (resblock.py:165-174, `downsample`) rides along as a TENTH TAP -- its input pixel (2 y, 2 x) is the
// centre tap's operand, already in registers: three more matrix instructions per interval into a second accumulator pair (weights: two 16-byte loads per lane and interval
// straight from L2, one interval ahead), a second epilogue that writes the skip map channels-last float32 (no bias: its BatchNorm shift sits in the block's second
// convolution, no ReLU).  One launch instead of two for the first block of a stage; the pointwise launch it replaces took 20-30 us for 0.5 us of matrix work.
// Weight image: the tap-major terms-16 image (coalign_conv3x3_emu_weight_bytes_ex(Cin, Cout, 16, 1)), as conv3x3_sp.hip.
//
// Arithmetic: per output value the SAME operations in the SAME order as conv3x3_sp.hip (intervals ascending, taps 0..8, w_h x_h -> acc, w_h x_l' then w_l' x_h
// -> accl, tile = (acc + 2^-10 accl) * 2^-k_c + bias): output (y, x) equals, bit for bit, output (2 y, 2 x) of coalign_conv3x3_sp on the same (dense) map
// (tests/test_s2_gpu.py).
//
// Tiles: 4 x 32 output pixels x 64 output channels per workgroup of 8 wavefronts; a wavefront owns a 4 x 8 pixel block x 32 channels (one 32 x 32 accumulator
// tile x two accumulators).  The halo patch is 9 x 65 input pixels; in LDS its columns are DE-INTERLEAVED (even columns | odd columns of a row, 34 groups each):
// the 8 pixels of a block row then read 8 consecutive groups for every tap, and rows two apart lie 136 = 8 (mod 16) groups apart -- the four lane groups of a
// ds_read_b128 touch disjoint banks, as with conv3x3_sp.hip's 4 x 8 blocks.  LDS: 2 x 36 KB weights + 2 x 40 KB patch + 8 KB bias / scale = 160 KB, one workgroup per CU.
#include "common.h"

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 halfx8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void *lptr_t;

constexpr int kCoutTile = 64;
constexpr int TH = 4, TW = 32;                     // output tile
constexpr int PR = 2 * TH + 1;                     // patch rows
constexpr int CPP = 34, ROWP = 2 * CPP;            // groups per column phase, per patch row
constexpr int PIX = PR * ROWP, PIXP = (PIX + 63) / 64 * 64, PINS = PIXP / 64;      // 612 -> 640 groups per plane, 10 DMA instructions
constexpr int WQ = 9 * 2 * 2 * kCoutTile, WINS = WQ / 64;                          // 2304 groups = 36 DMA instructions per interval
constexpr int W_BYTES = WQ * 16, B_BYTES = 4 * PIXP * 16;
constexpr int kMaxCout = 1024, kMaxCoutSkip = 512; // the layer's bias and 2^-k_c words (+ the skip's 2^-k_c words) live in LDS behind the operand buffers (8 KB)
constexpr size_t OPERAND_BYTES = 2 * (size_t)W_BYTES + 2 * (size_t)B_BYTES, LDS_BYTES = OPERAND_BYTES + 2 * kMaxCout * sizeof(float);
constexpr int WAVES = 8, THREADS = 64 * WAVES;
constexpr int WJ = 5, PJ = 2, OPS = WJ + 4 * PJ;   // DMA operations of a wavefront and interval (some empty): wavefronts 0-1 carry 3 weight pieces + 2 patch instructions x 4 planes, 2-7 5 + 1 x 4
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
static_assert(PINS == 10 && WINS == 36, "the piece schedule below is written for 10 patch and 36 weight instructions");

struct S2Args {
    const uint4 *__restrict__ x;        // dense: SP map; sparse: sp16 rows
    const unsigned long long *__restrict__ stamps;      // sparse only
    const int *__restrict__ tag_ptr;
    unsigned sparse_rows;
    const uint4 *__restrict__ wt, *__restrict__ zero;
    const float *__restrict__ bias, *__restrict__ wscale;
    uint4 *__restrict__ y;
    const uint4 *__restrict__ ws;        // SKIP: the 1 x 1 weight image [Cout / 64][Cin / 16][2 terms][2 channel halves][64 cout][8 cin] fp16
    const float *__restrict__ ws_scale;  // SKIP: [Cout] 2^-k_c of the skip weights
    float *__restrict__ yskip;           // SKIP: [N, Ho, Wo, Cout] channels-last float32
    int *range_flag;
    int N, Cin, Cout, H, W, Ho, Wo, relu, tiles_x, tiles_y, total_tasks;
    int ablate;                         // laboratory switch COALIGN_S2_ABLATE (tools/trace_conv_s2.py, tools/README.md)
#ifdef COALIGN_LAB
    long long *trace;                   // laboratory: [2 workgroups][8 waves][64 intervals][4 stamps] (tools/trace_conv_s2.py)
#endif
};

#ifdef COALIGN_LAB
#define S2_STAMP(k) if (a.trace && (g == 0 || g == 100) && lane == 0 && L < 64) a.trace[((((g ? 1 : 0) * 8 + wave) * 64) + L) * 4 + (k)] = (long long)__builtin_amdgcn_s_memtime()
#else
#define S2_STAMP(k)
#endif

__device__ __forceinline__ void swap32(unsigned &a, unsigned &b) {       // lanes 32-63 of a <-> lanes 0-31 of b
    const auto q = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    const unsigned x = q[0], y = q[1];
    a = x;
    b = y;
}

__device__ __forceinline__ void dma16(const uint4 *src, unsigned lds_byte) {      // 64 lanes x 16 bytes -> LDS [lds_byte, + 1024): lane l lands at + 16 l
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(__builtin_amdgcn_readfirstlane(lds_byte)), "v"(src) : "memory", "m0");
}

struct Task {
    int cg, n, oy0, ox0;
};

struct Plan {
    int off[PJ];             // the lane's group inside (interval 0, plane 0) of the input, -1 = zero
    const uint4 *wsrc;
};

template <bool SPARSE, bool SKIP>
__global__ __launch_bounds__(THREADS, 2) void conv3x3_sp_s2_kernel(const S2Args a) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, half = lane >> 5, p = lane & 31;
    const int HW = a.H * a.W, HWo = a.Ho * a.Wo, CI16 = a.Cin / 16, CO16 = a.Cout / 16, groups = a.Cout / kCoutTile, chunks = CI16;
    const unsigned lds0 = (unsigned)(size_t)(lptr_t)lds;
    const int bx = wave & 3, cq = wave >> 2;                                 // pixel block (column), channel half of the 64
    const int oyl = p >> 3, oxl = bx * 8 + (p & 7);                          // this lane's output pixel inside the tile
    int boff[9];                                                             // group of the lane's pixel under tap s, plane (2 * half + term 0)
#pragma unroll
    for (int s = 0; s < 9; ++s) {
        const int dy = s / 3, dx = s % 3;
        boff[s] = 2 * half * PIXP + (2 * oyl + dy) * ROWP + (dx & 1) * CPP + oxl + (dx >> 1);
    }
    const int wlane = half * kCoutTile + cq * 32 + p;                        // this lane's group inside one (tap, term) weight block
    const unsigned tag = SPARSE ? (unsigned)*a.tag_ptr : 0u;
    // bias | 2^-k_c of every output channel -> LDS, once per workgroup: the epilogue reads them with ds_read_b128 (its own counter) instead of global loads whose
    // compiler-placed vmcnt wait would also wait for the next interval's LDS-DMA (a task's epilogue 4000 -> 3200 clocks)
    float *lds_par = reinterpret_cast<float *>(lds + OPERAND_BYTES);
    for (int i = tid; i < a.Cout / 4; i += THREADS) {
        reinterpret_cast<float4 *>(lds_par)[i] = reinterpret_cast<const float4 *>(a.bias)[i];
        reinterpret_cast<float4 *>(lds_par + a.Cout)[i] = reinterpret_cast<const float4 *>(a.wscale)[i];
        if constexpr (SKIP) reinterpret_cast<float4 *>(lds_par + 2 * a.Cout)[i] = reinterpret_cast<const float4 *>(a.ws_scale)[i];
    }
    // SKIP: this lane's two weight groups (term 0 / 1) of interval c of channel group cg: the A operand of the tenth tap
    auto skip_w = [&](int cg, int c, uint4 (&w)[2]) {
        const uint4 *base = a.ws + ((size_t)(cg * chunks + c) * 4 + half) * kCoutTile + cq * 32 + p;
        w[0] = base[0];
        w[1] = base[2 * kCoutTile];
    };

    auto decode = [&](int t) {
        Task c;
        c.cg = t % groups;
        const int sp = t / groups, ty = sp / a.tiles_x;
        c.ox0 = (sp - ty * a.tiles_x) * TW;
        c.n = ty / a.tiles_y;
        c.oy0 = (ty - c.n * a.tiles_y) * TH;
        return c;
    };
    // patch position j of this lane (DMA instruction `wave`, and 8 + wave on the first two wavefronts): input pixel, or nothing
    auto position = [&](const Task &t, int j, int &iy, int &ix) {
        const int ins = j == 0 ? wave : 8 + wave;
        const int i = ins * 64 + lane, r = i / ROWP, rem = i - r * ROWP, cp = rem >= CPP ? 1 : 0, k = rem - cp * CPP;
        iy = 2 * t.oy0 - 1 + r;
        ix = 2 * t.ox0 - 1 + 2 * k + cp;
        return (j == 0 || wave < 2) && i < PIX && k < (cp ? TW : TW + 1) && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
    };
    auto make_plan_dense = [&](const Task &t) {
        Plan pl;
        pl.wsrc = a.wt + (size_t)t.cg * chunks * WQ + lane;
#pragma unroll
        for (int j = 0; j < PJ; ++j) {
            int iy, ix;
            const bool ok = position(t, j, iy, ix);
            pl.off[j] = ok ? t.n * CI16 * 4 * HW + iy * a.W + ix : -1;
        }
        return pl;
    };
    struct Stamps {
        unsigned long long st[PJ];
    };
    auto load_stamps = [&](const Task &t) {          // (cells outside the image read stamp 0: tag 0 is never current)
        Stamps s;
#pragma unroll
        for (int j = 0; j < PJ; ++j) {
            int iy, ix;
            const bool ok = position(t, j, iy, ix);
            s.st[j] = 0ull;
            if (ok) s.st[j] = a.stamps[((size_t)t.n * a.H + iy) * a.W + ix];
        }
        return s;
    };
    auto make_plan_sparse = [&](const Task &t, const Stamps &s) {
        Plan pl;
        pl.wsrc = a.wt + (size_t)t.cg * chunks * WQ + lane;
#pragma unroll
        for (int j = 0; j < PJ; ++j) {
            const unsigned row = (unsigned)s.st[j];
            const bool ok = (unsigned)(s.st[j] >> 32) == tag && tag != 0u && row < a.sparse_rows;
            pl.off[j] = ok ? (int)(row * (unsigned)(CI16 * 4)) : -1;
        }
        return pl;
    };
    // DMA operation k of this wavefront for interval c into buffer `slot`
    auto issue_op = [&](const Plan &pl, int c, int slot, int k) {
        if (k < WJ) {
            const bool ok = wave < 2 ? k < 3 : true;
            const int piece = wave < 2 ? 3 * wave + k : 6 + 5 * (wave - 2) + k;
            if (ok && !(a.ablate & 1)) dma16(pl.wsrc + (size_t)c * WQ + piece * 64, lds0 + slot * W_BYTES + piece * 1024);
        } else {
            const int j = (k - WJ) / 4, q = (k - WJ) % 4, ins = j == 0 ? wave : 8 + wave;
            if ((j == 0 || wave < 2) && !(a.ablate & 2)) {
                const uint4 *src = pl.off[j] < 0 ? a.zero : SPARSE ? a.x + (size_t)pl.off[j] + (c * 4 + q) : a.x + (size_t)pl.off[j] + ((size_t)c * 4 + q) * HW;
                dma16(src, lds0 + 2 * W_BYTES + slot * B_BYTES + (q * PIXP + ins * 64) * 16);
            }
        }
    };
    auto issue_all = [&](const Plan &pl, int c, int slot) {
#pragma unroll
        for (int k = 0; k < OPS; ++k) issue_op(pl, c, slot, k);
    };

    // persistent workgroups over whole tasks g, g + n, ...; XCD k takes the k-th eighth of the logical ids (neighbouring tasks share the patch and the weights in its L2)
    const int n_wg = gridDim.x;
    int g = blockIdx.x;
    {
        const int q = n_wg >> 3, r = n_wg & 7, k = g & 7, j = g >> 3;
        g = k * q + (k < r ? k : r) + j;
    }
    const int n_local = ((a.total_tasks - g + n_wg - 1) / n_wg) * chunks;
    if (n_local <= 0) return;
    int task = g;
    Task cur = decode(task);
    Plan plan;
    if constexpr (SPARSE) plan = make_plan_sparse(cur, load_stamps(cur));
    else plan = make_plan_dense(cur);
    issue_all(plan, 0, 0);
    if (wave >= WAVES / 2) __builtin_amdgcn_s_setprio(1);          // (as conv3x3_sp.hip: the later-dispatched half of the wavefronts loses every arbitration otherwise)
    // (Round 6, measured and dropped: the epilogue DEFERRED into the first interval of the workgroup's next task, its four steps behind taps 1, 3, 5, 7 -- the
    //  interval's body grew by the epilogue's ~2000 clocks (3330 -> 5180: a wavefront's instruction stream, not the matrix pipe, bounds an interval) and the three
    //  layers got 1.5-2 us slower: profiles/round6/experiments/conv_sp_s2.md.)
    uint4 sw_cur[2] = {}, sw_nxt[2] = {};                          // SKIP: the tenth tap's weight operands of this / the next interval
    if constexpr (SKIP) skip_w(cur.cg, 0, sw_nxt);
    int L = 0;
    while (L < n_local) {
        const int oy = cur.oy0 + oyl, ox = cur.ox0 + oxl;
        const bool live = oy < a.Ho && ox < a.Wo;
        const bool wave_live = __builtin_amdgcn_readfirstlane((int)(cur.ox0 + bx * 8 < a.Wo)) != 0;
        const size_t pix = live ? (size_t)oy * a.Wo + ox : 0;
        floatx16 acc = {0}, accl = {0}, sacc = {0}, saccl = {0};
        Task next = cur;
        Plan nplan = plan;
        int ntask = task;
        Stamps nst;
#pragma unroll
        for (int j = 0; j < PJ; ++j) nst.st[j] = 0ull;
        for (int chunk = 0; chunk < chunks; ++chunk, ++L) {
            S2_STAMP(0);
            __builtin_amdgcn_s_waitcnt(0);
            S2_STAMP(1);
            __syncthreads();
            S2_STAMP(2);
            if constexpr (SKIP) {                                  // first touch of the words loaded an interval ago: nothing of this wavefront is in flight here
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    asm volatile("" : "+v"(sw_nxt[t].x), "+v"(sw_nxt[t].y), "+v"(sw_nxt[t].z), "+v"(sw_nxt[t].w));
                    sw_cur[t] = sw_nxt[t];
                }
            }
            const bool more = L + 1 < n_local;
            int nc = chunk + 1;
            if constexpr (SPARSE) {                                 // the next task's stamps: one interval ahead of its first DMA
                if (chunk == chunks - 2 && L + 2 < n_local) {
                    ntask = task + n_wg;
                    next = decode(ntask);
                    nst = load_stamps(next);
                }
            }
            if (more && nc == chunks) {                             // the next interval opens this workgroup's next task
                nc = 0;
                if constexpr (SPARSE) {
                    nplan = make_plan_sparse(next, nst);
                } else {
                    ntask = task + n_wg;
                    next = decode(ntask);
                    nplan = make_plan_dense(next);
                }
            }
            if constexpr (SKIP) {                                  // ... and the next interval's, ahead of this interval's DMA
                if (more) skip_w(nc == 0 ? next.cg : cur.cg, nc, sw_nxt);
            }
            const int slot_cur = L & 1, slot_next = (L + 1) & 1;
            if (wave_live && !(a.ablate & 4)) {
                const uint4 *bq = reinterpret_cast<const uint4 *>(lds + 2 * W_BYTES + slot_cur * B_BYTES), *wq = reinterpret_cast<const uint4 *>(lds + slot_cur * W_BYTES) + wlane;
                auto load_b = [&](int s, halfx8 (&b)[2]) {
#pragma unroll
                    for (int t = 0; t < 2; ++t) b[t] = __builtin_bit_cast(halfx8, bq[t * PIXP + boff[s]]);
                };
                auto load_w = [&](int s, halfx8 (&w)[2]) {
#pragma unroll
                    for (int t = 0; t < 2; ++t) w[t] = __builtin_bit_cast(halfx8, wq[((s * 2 + t) * 2) * kCoutTile]);
                };
                halfx8 bc[2], wc[2];
                load_b(0, bc);
                load_w(0, wc);
#pragma unroll
                for (int s = 0; s < 9; ++s) {
                    if (!(a.ablate & 16)) {                        // progress-based issue priority (conv3x3_sp.hip): whoever is behind wins the SIMD's matrix pipe
                        if (s == 0) __builtin_amdgcn_s_setprio(2);
                        if (s == 3) __builtin_amdgcn_s_setprio(1);
                        if (s == 6) __builtin_amdgcn_s_setprio(0);
                    }
                    halfx8 bn[2], wn[2];
                    if (s + 1 < 9) {                               // operands of the next tap are in flight while this tap's matrix instructions issue (two taps ahead: no gain measured)
                        load_b(s + 1, bn);
                        load_w(s + 1, wn);
                    }
                    accl = __builtin_amdgcn_mfma_f32_32x32x16_f16(wc[0], bc[1], accl, 0, 0, 0);      // w_h x_l'
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wc[0], bc[0], acc, 0, 0, 0);        // w_h x_h
                    accl = __builtin_amdgcn_mfma_f32_32x32x16_f16(wc[1], bc[0], accl, 0, 0, 0);      // w_l' x_h
                    if constexpr (SKIP) {
                        if (s == 4) {                              // the centre tap's pixel is the 1 x 1 / stride-2 convolution's input pixel
                            const halfx8 sh = __builtin_bit_cast(halfx8, sw_cur[0]), sl = __builtin_bit_cast(halfx8, sw_cur[1]);
                            saccl = __builtin_amdgcn_mfma_f32_32x32x16_f16(sh, bc[1], saccl, 0, 0, 0);
                            sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(sh, bc[0], sacc, 0, 0, 0);
                            saccl = __builtin_amdgcn_mfma_f32_32x32x16_f16(sl, bc[0], saccl, 0, 0, 0);
                        }
                    }
                    if (more) {
                        if (a.ablate & 32) {                       // laboratory: one operation behind every tap (the first version) instead of front-loaded
#pragma unroll
                            for (int k = s; k < OPS; k += 9) issue_op(nplan, nc, slot_next, k);
                        } else {                                   // front-loaded: two operations behind each of the first taps (as conv3x3_sp.hip's 8-wavefront geometries)
#pragma unroll
                            for (int k = 2 * s; k < 2 * s + 2; ++k)
                                if (k < OPS) issue_op(nplan, nc, slot_next, k);
                        }
                    }
                    if (s + 1 < 9) {
#pragma unroll
                        for (int t = 0; t < 2; ++t) {
                            bc[t] = bn[t];
                            wc[t] = wn[t];
                        }
                    }
                }
            } else if (more) {
                issue_all(nplan, nc, slot_next);
            }
            S2_STAMP(3);
        }
        // ---- epilogue: y = (acc + 2^-10 accl) * 2^-k_c + bias, ReLU, stored as an SP map (conv3x3_sp.hip's epilogue for this wavefront's 32 channels)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = fmaf(accl[e], coalign::kSp16LowInv, acc[e]);
        const int ch0 = cur.cg * kCoutTile + cq * 32 + 4 * half;
        const float4 *bias4 = reinterpret_cast<const float4 *>(lds_par + ch0), *winv4 = reinterpret_cast<const float4 *>(lds_par + a.Cout + ch0);
        const float floor_v = a.relu ? 0.f : -__builtin_inff();
        float vmax = 0.f;
        uint4 *ysp = a.y + ((size_t)(cur.n * CO16 + cur.cg * 4 + cq * 2) * 4 + half) * HWo + pix;
        const size_t sp_step = 2 * (size_t)HWo;
#pragma unroll
        for (int g8 = 0; g8 < 4; ++g8) {                           // 4 groups of 4 consecutive channels per lane: channel = ch0 + 8 g8 + j
            const float4 b4 = bias4[2 * g8], i4 = winv4[2 * g8];
            const float bb[4] = {b4.x, b4.y, b4.z, b4.w}, ii[4] = {i4.x, i4.y, i4.z, i4.w};
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = fmaxf(acc[4 * g8 + j] * ii[j] + (0.f + bb[j]), floor_v);
            vmax = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fmaxf(fabsf(v[2]), fabsf(v[3])), vmax));
            unsigned h01, l01, h23, l23;
            coalign::sp16_split2(v[0], v[1], h01, l01);
            coalign::sp16_split2(v[2], v[3], h23, l23);
            swap32(h01, l01);          // lanes 0-31: h of channels 0,1 | 4,5 of the 8-channel group; lanes 32-63: l of the same channels
            swap32(h23, l23);
            if (live && wave_live && !(a.ablate & 8)) *ysp = uint4{h01, h23, l01, l23};
            ysp += sp_step;
        }
        if (a.range_flag && live && wave_live && vmax > 65504.f) atomicOr(a.range_flag, 1);
        if constexpr (SKIP) {                                       // the skip map: sums * 2^-k_c, channels-last float32 (16 bytes per pixel and store)
            const float4 *sinv4 = reinterpret_cast<const float4 *>(lds_par + 2 * a.Cout + ch0);
            float4 *ys = reinterpret_cast<float4 *>(a.yskip + ((size_t)cur.n * HWo + pix) * a.Cout + ch0);
#pragma unroll
            for (int g8 = 0; g8 < 4; ++g8) {
                const float4 i4 = sinv4[2 * g8];
                float4 o;
                o.x = fmaf(saccl[4 * g8], coalign::kSp16LowInv, sacc[4 * g8]) * i4.x;
                o.y = fmaf(saccl[4 * g8 + 1], coalign::kSp16LowInv, sacc[4 * g8 + 1]) * i4.y;
                o.z = fmaf(saccl[4 * g8 + 2], coalign::kSp16LowInv, sacc[4 * g8 + 2]) * i4.z;
                o.w = fmaf(saccl[4 * g8 + 3], coalign::kSp16LowInv, sacc[4 * g8 + 3]) * i4.w;
                if (live && wave_live && !(a.ablate & 8)) ys[2 * g8] = o;
            }
        }
        cur = next;
        plan = nplan;
        task = ntask;
    }
}

// float32 feature rows [M][C] (pillar_sparse.hip) -> sp16 rows [M][C / 16][4 planes][8 x fp16]: thread = (row, 8-channel group)
__global__ void sp_pack_rows_kernel(const float *__restrict__ rows, uint4 *__restrict__ out, int M, const int *__restrict__ count_ptr, int C, int *range_flag) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int G8 = C / 8;
    if (count_ptr) M = min(max(*count_ptr, 0), M);
    if (i >= (size_t)M * G8) return;
    const int g8 = (int)(i % G8);
    const size_t row = i / G8;
    const float4 *src = reinterpret_cast<const float4 *>(rows + row * C + 8 * g8);
    const float4 q0 = src[0], q1 = src[1];
    const float v[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
    unsigned h[4], l[4];
    bool big = false;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        coalign::sp16_split2(v[2 * k], v[2 * k + 1], h[k], l[k]);
        big = big || fabsf(v[2 * k]) > 65504.f || fabsf(v[2 * k + 1]) > 65504.f;
    }
    const size_t base = row * (size_t)(C / 16 * 4) + (size_t)(g8 / 2) * 4 + (g8 % 2) * 2;
    out[base] = uint4{h[0], h[1], h[2], h[3]};
    out[base + 1] = uint4{l[0], l[1], l[2], l[3]};
    if (range_flag && big) atomicOr(range_flag, 1);
}

#ifdef COALIGN_LAB
long long *g_s2_trace = nullptr;
#endif

int s2_check(int N, int Cin, int Cout, int H, int W) {
    if (N < 0 || H < 1 || W < 1 || Cin < 1 || Cout < 1) return COALIGN_ERR_BAD_SHAPE;
    if (Cin % 16 || Cout % kCoutTile || Cout > kMaxCout) return COALIGN_ERR_UNSUPPORTED;
    if ((int64_t)N * (Cin > Cout ? Cin : Cout) * H * W > (int64_t)1 << 32) return COALIGN_ERR_UNSUPPORTED;      // group offsets are 32-bit
    return COALIGN_OK;
}

template <bool SPARSE, bool SKIP>
int launch_s2(S2Args a, hipStream_t s) {
    constexpr int kMaxDev = 16;
    static int cus[kMaxDev] = {0};                                    // per device: the function attribute belongs to the device's code object
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) dev = 0;
    auto kern = conv3x3_sp_s2_kernel<SPARSE, SKIP>;
    if (!cus[dev]) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess || prop.multiProcessorCount <= 0) prop.multiProcessorCount = 256;
        const int rc = coalign::hip_call(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES));
        if (rc != COALIGN_OK) {
            (void)hipGetLastError();
            return rc;
        }
        cus[dev] = prop.multiProcessorCount;
    }
    a.Ho = (a.H + 1) / 2;
    a.Wo = (a.W + 1) / 2;
    a.tiles_x = (a.Wo + TW - 1) / TW;
    a.tiles_y = (a.Ho + TH - 1) / TH;
    a.total_tasks = a.tiles_x * a.tiles_y * a.N * (a.Cout / kCoutTile);
    // as many workgroups as give every one the same number of tasks (+-1): 780 tasks on 256 CUs are four rounds either way, 195 workgroups leave 61 CUs to the
    // other frame's kernels
    const int rounds = (a.total_tasks + cus[dev] - 1) / cus[dev], grid = (a.total_tasks + rounds - 1) / rounds;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(THREADS), LDS_BYTES, s, a);
    return coalign::check_launch();
}

int fill_common(S2Args &a, const void *w_split, const float *bias, void *y_sp, int N, int Cin, int Cout, int H, int W, int relu, int32_t *range_flag) {
    const size_t wbytes = coalign_conv3x3_emu_weight_bytes_ex(Cin, Cout, 16, 1), tail = (size_t)Cout * 8;
    const char *wb = static_cast<const char *>(w_split);
    a.wt = static_cast<const uint4 *>(w_split);
    a.zero = reinterpret_cast<const uint4 *>(wb + wbytes - tail - 16);
    a.bias = bias;
    a.wscale = reinterpret_cast<const float *>(wb + wbytes - tail);
    a.y = static_cast<uint4 *>(y_sp);
    a.range_flag = range_flag;
    a.N = N; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W; a.relu = relu;
#ifdef COALIGN_LAB
    a.trace = g_s2_trace;
#endif
    a.ablate = coalign::lab_env("COALIGN_S2_ABLATE", 0);      // laboratory build: 1 no weight DMA, 2 no patch DMA, 4 no matrix steps, 8 no stores, 16 fixed issue priorities, 32 DMA one piece per tap
    return COALIGN_OK;
}

}  // namespace

#ifdef COALIGN_LAB
extern "C" void coalign_conv3x3_sp_s2_set_trace(long long *p) { g_s2_trace = p; }
#endif

extern "C" size_t coalign_sp_rows_bytes(int M, int C) {
    if (M < 0 || C < 16 || C % 16) return 0;
    return (size_t)M * C * 4;
}

extern "C" int coalign_sp_pack_rows(const float *rows, int M_capacity, const int32_t *M_dev, int C, void *rows_sp, int32_t *range_flag, void *stream) {
    if (!rows || !rows_sp) return COALIGN_ERR_NULL_POINTER;
    if (M_capacity < 0 || C < 16) return COALIGN_ERR_BAD_SHAPE;
    if (C % 16 || ((reinterpret_cast<uintptr_t>(rows) | reinterpret_cast<uintptr_t>(rows_sp)) & 15)) return COALIGN_ERR_UNSUPPORTED;
    const size_t n = (size_t)M_capacity * (C / 8);
    if (n == 0) return COALIGN_OK;
    hipLaunchKernelGGL(sp_pack_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), rows, static_cast<uint4 *>(rows_sp), M_capacity, M_dev, C,
                       range_flag);
    return coalign::check_launch();
}

extern "C" int coalign_conv3x3_sp_s2(const void *x_sp, const void *w_split, const float *bias, void *y_sp, int N, int Cin, int Cout, int H, int W, int relu,
                                     int32_t *range_flag, void *stream) {
    if (!x_sp || !w_split || !bias || !y_sp) return COALIGN_ERR_NULL_POINTER;
    int rc = s2_check(N, Cin, Cout, H, W);
    if (rc != COALIGN_OK) return rc;
    if ((reinterpret_cast<uintptr_t>(x_sp) | reinterpret_cast<uintptr_t>(w_split) | reinterpret_cast<uintptr_t>(y_sp) | reinterpret_cast<uintptr_t>(bias)) & 15) return COALIGN_ERR_UNSUPPORTED;
    if (N == 0) return COALIGN_OK;
    S2Args a{};
    a.x = static_cast<const uint4 *>(x_sp);
    fill_common(a, w_split, bias, y_sp, N, Cin, Cout, H, W, relu, range_flag);
    return launch_s2<false, false>(a, static_cast<hipStream_t>(stream));
}

extern "C" int coalign_conv3x3_sp_s2_sparse(const void *rows_sp, int M_rows, const void *stamps, const int32_t *state, const void *w_split, const float *bias, void *y_sp,
                                            int N, int Cin, int Cout, int H, int W, int relu, int32_t *range_flag, void *stream) {
    if (!rows_sp || !stamps || !state || !w_split || !bias || !y_sp) return COALIGN_ERR_NULL_POINTER;
    if (M_rows < 0) return COALIGN_ERR_BAD_SHAPE;
    int rc = s2_check(N, Cin, Cout, H, W);
    if (rc != COALIGN_OK) return rc;
    if (Cin < 32) return COALIGN_ERR_UNSUPPORTED;                     // (a task's stamps are fetched one interval ahead: at least two intervals per task)
    if ((reinterpret_cast<uintptr_t>(rows_sp) | reinterpret_cast<uintptr_t>(w_split) | reinterpret_cast<uintptr_t>(y_sp) | reinterpret_cast<uintptr_t>(bias)) & 15) return COALIGN_ERR_UNSUPPORTED;
    if (reinterpret_cast<uintptr_t>(stamps) & 7) return COALIGN_ERR_UNSUPPORTED;
    if ((int64_t)M_rows * Cin / 4 >= (int64_t)1 << 31) return COALIGN_ERR_UNSUPPORTED;
    if (N == 0) return COALIGN_OK;
    S2Args a{};
    a.x = static_cast<const uint4 *>(rows_sp);
    a.stamps = static_cast<const unsigned long long *>(stamps);
    a.tag_ptr = state;
    a.sparse_rows = (unsigned)M_rows;
    fill_common(a, w_split, bias, y_sp, N, Cin, Cout, H, W, relu, range_flag);
    return launch_s2<true, false>(a, static_cast<hipStream_t>(stream));
}

// ---- the same launches with the block's 1 x 1 / stride-2 skip convolution as a tenth tap (SKIP) ----------------------------------------------------------------
extern "C" size_t coalign_conv1x1_sp_weight_bytes(int Cin, int Cout) {
    if (Cin < 16 || Cin % 16 || Cout < kCoutTile || Cout % kCoutTile) return 0;
    return (size_t)Cout * Cin * 4 + 16 + (size_t)Cout * 8;      // two fp16 terms per weight, 16 zero bytes, [Cout] 2^-k_c, [Cout] 2^k_c
}

static int fill_skip(S2Args &a, const void *w_skip, float *y_skip, int Cin, int Cout) {
    if (!w_skip || !y_skip) return COALIGN_ERR_NULL_POINTER;
    if (Cout > kMaxCoutSkip || ((reinterpret_cast<uintptr_t>(w_skip) | reinterpret_cast<uintptr_t>(y_skip)) & 15)) return COALIGN_ERR_UNSUPPORTED;
    a.ws = static_cast<const uint4 *>(w_skip);
    a.ws_scale = reinterpret_cast<const float *>(static_cast<const char *>(w_skip) + (size_t)Cout * Cin * 4 + 16);
    a.yskip = y_skip;
    return COALIGN_OK;
}

extern "C" int coalign_conv3x3_sp_s2_skip(const void *x_sp, const void *w_split, const float *bias, const void *w_skip, void *y_sp, float *y_skip, int N, int Cin, int Cout, int H, int W,
                                          int relu, int32_t *range_flag, void *stream) {
    if (!x_sp || !w_split || !bias || !y_sp) return COALIGN_ERR_NULL_POINTER;
    int rc = s2_check(N, Cin, Cout, H, W);
    if (rc != COALIGN_OK) return rc;
    if ((reinterpret_cast<uintptr_t>(x_sp) | reinterpret_cast<uintptr_t>(w_split) | reinterpret_cast<uintptr_t>(y_sp) | reinterpret_cast<uintptr_t>(bias)) & 15) return COALIGN_ERR_UNSUPPORTED;
    S2Args a{};
    rc = fill_skip(a, w_skip, y_skip, Cin, Cout);
    if (rc != COALIGN_OK) return rc;
    if (N == 0) return COALIGN_OK;
    a.x = static_cast<const uint4 *>(x_sp);
    fill_common(a, w_split, bias, y_sp, N, Cin, Cout, H, W, relu, range_flag);
    return launch_s2<false, true>(a, static_cast<hipStream_t>(stream));
}

extern "C" int coalign_conv3x3_sp_s2_skip_sparse(const void *rows_sp, int M_rows, const void *stamps, const int32_t *state, const void *w_split, const float *bias, const void *w_skip,
                                                 void *y_sp, float *y_skip, int N, int Cin, int Cout, int H, int W, int relu, int32_t *range_flag, void *stream) {
    if (!rows_sp || !stamps || !state || !w_split || !bias || !y_sp) return COALIGN_ERR_NULL_POINTER;
    if (M_rows < 0) return COALIGN_ERR_BAD_SHAPE;
    int rc = s2_check(N, Cin, Cout, H, W);
    if (rc != COALIGN_OK) return rc;
    if (Cin < 32) return COALIGN_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(rows_sp) | reinterpret_cast<uintptr_t>(w_split) | reinterpret_cast<uintptr_t>(y_sp) | reinterpret_cast<uintptr_t>(bias)) & 15) return COALIGN_ERR_UNSUPPORTED;
    if (reinterpret_cast<uintptr_t>(stamps) & 7) return COALIGN_ERR_UNSUPPORTED;
    if ((int64_t)M_rows * Cin / 4 >= (int64_t)1 << 31) return COALIGN_ERR_UNSUPPORTED;
    S2Args a{};
    rc = fill_skip(a, w_skip, y_skip, Cin, Cout);
    if (rc != COALIGN_OK) return rc;
    if (N == 0) return COALIGN_OK;
    a.x = static_cast<const uint4 *>(rows_sp);
    a.stamps = static_cast<const unsigned long long *>(stamps);
    a.tag_ptr = state;
    a.sparse_rows = (unsigned)M_rows;
    fill_common(a, w_split, bias, y_sp, N, Cin, Cout, H, W, relu, range_flag);
    return launch_s2<true, true>(a, static_cast<hipStream_t>(stream));
}
