// 3x3 / stride 1 / pad 1 convolution whose INPUT MAP ARRIVES ALREADY SPLIT (round 5), gfx950.  Same layers and arithmetic as the fp16 mode of
// conv3x3_emu.hip (opencood/models/sub_modules/resblock.py:53-69, base_bev_backbone_resnet.py:59-138, downsample_conv.py:7-50): every fp32 product is
// evaluated on v_mfma_f32_32x32x16_f16 from sp16 pairs (common.h), fp32 accumulation, bias + residual + ReLU in the epilogue -- but the operand split
// has left the consumer.  Round 4's interval timelines (profiles/round4/experiments/conv_interval_timeline_fp16x2_bf16x3.txt) showed half of every
// convolution going into work a consumer repeats per layer: 16 dword loads per lane, the VALU split, ds_write of the split patch, a second barrier.
// Here a layer's epilogue writes its output map as an "SP map": the pair (h, l) of every value in the 32 bits the fp32 value would occupy, laid out
// in the matrix instruction's operand order,
//     SP map of a logical [N, C, H, W] tensor (C % 16 == 0):   [N][C / 16][4 planes][H][W][8 x fp16],   plane = 2 * (channel half of the 16) + term,
// i.e. one 16-byte group = one B operand (8 consecutive input channels of one pixel, one term).  The consumer's halo patch of a 16-channel interval is then
// four planes of PH x PW groups that travel global -> LDS by LDS-DMA (global_load_lds_dwordx4: 64 consecutive groups of a plane per instruction, rows of
// the patch contiguous in memory), next to the weight image's DMA: the K loop contains NO VALU work, no register staging, ONE barrier per interval, both
// operands double buffered.  Zero padding = lanes whose group lies outside the image fetch the weight image's 16 zero bytes instead.
//
// Weight image: the tap-major fp16 image of conv3x3_emu.hip (coalign_conv3x3_emu_weight_bytes_ex(Cin, Cout, 16, 1): [Cout / 64][Cin / 16][9 taps][2 terms]
// [2 channel halves][64 cout][8 cin] + 16 zero bytes + the per-channel scale tail).  Products: w_h x_h -> accumulator `acc`; w_h x_l', w_l' x_h (both
// carry 2^10) -> accumulator `accl`; tile = (acc + 2^-10 accl) * 2^-k_c, with (bias + residual) * 2^k_c as acc's start value: the SAME operations in
// the same order as the fp16 mode of conv3x3_emu.hip, so the two kernels agree bit for bit on the same (22-bit) inputs (tests/test_round5_gpu.py).
//
// Tiles: the batch is tiled as ONE image of N * H rows (as conv3x3_emu.hip's stacked variants: a tile may straddle two images, two zero rows in the
// patch stand in for the padding between them); a workgroup = NPB wavefronts, each owning a BH x BW block of 32 pixels x 64 output channels (two 32 x 32
// accumulator tiles x two accumulators); persistent workgroups over (tile, interval) steps, XCD-aware tile order.
#include "common.h"

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef unsigned uintx4 __attribute__((ext_vector_type(4)));
typedef _Float16 halfx8 __attribute__((ext_vector_type(8)));
typedef _Float16 halfx4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void *lptr_t;

constexpr int kCoutTile = 64;
constexpr int kMaxCoutLds = 1024;       // output channels whose bias and scale words fit the 8 KB of LDS behind the operand buffers

struct SpArgs {
    const uint4 *__restrict__ x;        // SP map of the input
    const uint4 *__restrict__ wt;       // tap-major fp16 weight image
    const uint4 *__restrict__ zero;     // 16 zero bytes (the weight image's zero group)
    const float *__restrict__ bias, *__restrict__ wscale;      // wscale: [Cout] 2^-k_c, [Cout] 2^k_c
    const void *__restrict__ residual;  // SP map or channels-last fp32, per res_kind
    void *__restrict__ y;               // SP map or channels-last fp32, per the OUT template argument
    void *__restrict__ y2;              // OUT = SP_OUT_BOTH (round 6): the SP map beside the channels-last fp32 map in y (a stage's last layer: the fusion kernel reads y, the next stage's strided convolution y2)
    int *range_flag;                    // may be NULL: bit 0 is set when an SP output value exceeds the pair's range (|y| > 65504)
    int N, Cin, Cout, H, W, relu, res_kind, stack, tiles_x, tiles_y, total_tiles, xcd;
    int prio_mode;                      // round 6: 1 = progress-based issue priority (see the K loop); laboratory switch COALIGN_SP_PRIO
    int stream_out;                     // laboratory switch (round 6): the SP output leaves with streaming (non-temporal) stores (common.h store_stream); measured, not adopted
    // stream-K (split != 0): the (tile, interval) steps are cut into gridDim.x equal contiguous ranges; a workgroup that starts in the middle of a tile
    // publishes the partial sums of its share (slot g of `partial`, then flags[g] = 1), the workgroup that OPENED the tile adds them and runs the epilogue.
    int split;
    float *partial;                     // [grid][waves][32][64] float
    int *flags;                         // [grid], zero before the first launch; the consumer of a flag resets it
#ifdef SP_TRACE
    long long *trace;                   // profiling aid (tools/trace_conv_sp.py): [2 workgroups][16 waves][64 intervals][8 stamps] + [grid][2] wall clocks
    int ablate;                         // 1: no weight DMA, 2: no patch DMA, 4: no matrix steps (no LDS reads either), 8: no residual / bias start, 16: no stores
#endif
};

#ifdef SP_TRACE
#define SP_STAMP(k)                                                                                              \
    if ((g == 0 || g == 100) && lane == 0 && L < 64)                                                             \
        a.trace[((((g ? 1 : 0) * 16 + wave) * 64) + L) * 8 + (k)] = (long long)__builtin_amdgcn_s_memtime()
#define SP_STAMP_AT(k, idx)                                                                                      \
    if ((g == 0 || g == 100) && lane == 0 && (idx) >= 0 && (idx) < 64)                                           \
        a.trace[((((g ? 1 : 0) * 16 + wave) * 64) + (idx)) * 8 + (k)] = (long long)__builtin_amdgcn_s_memtime()
#define SP_ABLATE(bit) (a.ablate & (bit))
#else
#define SP_STAMP(k)
#define SP_STAMP_AT(k, idx)
#define SP_ABLATE(bit) 0
#endif

// CT (round 6): output channels of a tile -- 64 (a wavefront owns two 32 x 32 accumulator tiles x two accumulators), or 32 (one tile: half the matrix work per
// wavefront and interval, twice the tasks -- the 25 x 88 maps fill 256 CUs with 12-wavefront workgroups instead of 192 with 8-wavefront ones)
template <int BH, int BW, int NPB, int NBX, int CT = 64>
struct Geo {
    static_assert(BH * BW == 32 && NPB % NBX == 0, "a wavefront owns 32 pixels; whole block rows");
    static_assert(CT == 64 || CT == 32, "output-channel tile");
    static constexpr int NQ = CT / 32, NG8 = CT / 8;                           // 32 x 32 accumulator tiles per accumulator; 8-channel groups
    static constexpr int WAVES = NPB, THREADS = 64 * NPB;
    static constexpr int TH = BH * NPB / NBX, TW = BW * NBX;                   // output tile
    static constexpr int PWU = TW + 2;                                         // patch columns in use
    static constexpr int PW = NBX == 1 ? PWU : (PWU + 7) / 16 * 16 + 8;        // (as conv3x3_emu.hip: rows of a 4 x 8 block in alternating bank halves)
    static constexpr int PH = TH + 4;                                          // halo + the two zero rows of an image boundary inside the tile
    static constexpr int PIX = PH * PW, PIXP = (PIX + 63) / 64 * 64;           // groups per plane, padded to whole DMA instructions
    static constexpr int PINS = PIXP / 64;                                     // DMA instructions per plane
    static constexpr int WQ = 9 * 2 * 2 * CT;                                  // 16-byte groups of one interval's weights in LDS
    static constexpr int WQ_SRC = 9 * 2 * 2 * kCoutTile;                       // ... in the weight image (64 output channels per block)
    static constexpr int WINS = WQ / 64;
    static constexpr int W_BYTES = WQ * 16, B_BYTES = 4 * PIXP * 16;
    static constexpr size_t LDS_BYTES = 2 * (size_t)W_BYTES + 2 * (size_t)B_BYTES;      // both operands double buffered (Work::LDS_BYTES: what a mode really takes)
    static_assert(W_BYTES + B_BYTES <= 160 * 1024, "geometry does not fit the 160 KB LDS");
};

struct Tile {
    int cg, n0, yl0, yb, x0;      // output-channel group; image of the tile's first row, that row inside the image, rows left in the image (H - yl0; huge if not stacked); first column
};

__device__ __forceinline__ void swap32(unsigned &a, unsigned &b) {       // lanes 32-63 of a <-> lanes 0-31 of b
    const auto q = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    const unsigned x = q[0], y = q[1];
    a = x;
    b = y;
}

__device__ __forceinline__ void dma16(const uint4 *src, unsigned lds_byte) {      // 64 lanes x 16 bytes -> LDS [lds_byte, + 1024): lane l lands at + 16 l
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(__builtin_amdgcn_readfirstlane(lds_byte)), "v"(src) : "memory", "m0");
}

enum { SP_OUT_SP = 1, SP_OUT_NHWC = 2, SP_OUT_BOTH = 3 };
enum { SP_RES_NONE = 0, SP_RES_SP = 1, SP_RES_NHWC = 2 };

// MODE: who issues the LDS-DMA of the next interval, and when (measured: tools/trace_conv_sp.py)
//   0  every wavefront its share, all at once behind the interval's barrier (the matrix pipe idles until the texture addresser has taken the ~70 instructions);
//   1  every wavefront its share, one or two instructions behind each tap's matrix instructions;
//   2  LOADER wavefronts: the first four wavefronts (one per SIMD) issue everything, the others start their matrix steps straight away -- a SIMD's
//      loader runs its steps when its partners have finished theirs, the pipe never waits for the addresser;
//   3  one EXTRA wavefront per workgroup that only issues the DMA (8-wavefront geometries: 9 wavefronts = three per SIMD within the register budget):
//      the computing wavefronts' instruction streams are ds_read + matrix instructions only;
//   4  as 1, the tap's DMA instructions IN FRONT of its matrix instructions;   5  as 1, front-loaded: two instructions behind each of the first taps.
//   7  (round 6) as 3 with FOUR loader wavefronts, one per SIMD (8-wavefront geometries: 12 wavefronts = three per SIMD within the register budget): 17 DMA
//      instructions per loader and interval instead of 68 on one -- the computing wavefronts' streams are ds_read + matrix instructions only;
//   6  (round 6) TWO WORKGROUPS PER CU, out of phase: ONE weight buffer + ONE patch buffer (65-70 KB of LDS on the 8-wavefront geometries), registers capped at
//      128 (four wavefronts per SIMD), every wavefront issues its share of the next interval's DMA at once behind the barrier that ends an interval and waits for it.
//      A workgroup alone would expose the DMA latency every interval; its PARTNER on the CU runs matrix steps meanwhile (and through the other one's tile prologue,
//      epilogue and barrier waits: profiles/round5/experiments/conv_sp_epilogue_phases.md named this remedy).  The workgroup in the CU's odd slot (HW_ID.TG_ID) issues
//      at a higher priority, so that two workgroups that start in step fall out of step instead of sharing the pipe and then loading together.
template <int BH, int BW, int NPB, int NBX, int MODE, int CT = 64>
struct Work {
    using G = Geo<BH, BW, NPB, NBX, CT>;
    static constexpr int EXTRA = MODE == 3 ? 1 : MODE == 7 ? 4 : 0;             // loader-only wavefronts on top of the computing ones (MODE 7, round 6: one per SIMD)
    static constexpr int LOADERS = MODE == 2 ? 4 : EXTRA ? EXTRA : G::WAVES;
    static constexpr bool INTERLEAVED = MODE == 1 || MODE == 4 || MODE == 5;
    static constexpr bool PAIRED = MODE == 6;                                   // two workgroups per CU, single buffers
    static constexpr int NBUF = PAIRED ? 1 : 2;
    static constexpr size_t OPERAND_BYTES = (size_t)NBUF * ((size_t)G::W_BYTES + (size_t)G::B_BYTES);
    static constexpr size_t LDS_BYTES = OPERAND_BYTES + 2 * kMaxCoutLds * sizeof(float);      // + the layer's bias | 2^-k_c words (round 6: the epilogue reads them from LDS)
    static constexpr int WAVES_PER_EU = PAIRED ? 4 : (NPB + 3) / 4;              // (the defaults the work-group size implies, except for the paired mode's cap)
    static_assert(!PAIRED || NPB == 8, "two workgroups per CU: 8-wavefront geometries (16 wavefronts per CU = 4 per SIMD at 128 registers)");
    static_assert(!PAIRED || 2 * LDS_BYTES <= 160 * 1024, "two workgroups of this geometry do not fit the 160 KB LDS");
    static_assert(LDS_BYTES <= 160 * 1024, "geometry does not fit the 160 KB LDS in this mode");
    static constexpr int THREADS = G::THREADS + 64 * EXTRA;
    static constexpr int WJ = (G::WINS + LOADERS - 1) / LOADERS, PJ = (G::PINS + LOADERS - 1) / LOADERS, OPS = WJ + 4 * PJ;
};

template <int BH, int BW, int NPB, int NBX, int OUT, int MODE, bool SPLIT, int CT = 64>
__global__ __launch_bounds__(64 * NPB + (MODE == 3 ? 64 : MODE == 7 ? 256 : 0), (MODE == 6 ? 4 : MODE == 7 ? (NPB + 7) / 4 : (NPB + 3) / 4)) void conv3x3_sp_kernel(const SpArgs a) {
    using G = Geo<BH, BW, NPB, NBX, CT>;
    using K = Work<BH, BW, NPB, NBX, MODE, CT>;
    static_assert(!(K::PAIRED && SPLIT), "the paired mode runs whole tiles");
    static_assert(CT == 64 || !SPLIT, "stream-K hand-overs carry 64-channel tiles");
    constexpr int NQ = G::NQ, NG8 = G::NG8;
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    __shared__ int s_share_ready;                                   // stream-K: "the next gang's share is there already" (decided by one lane, read by all behind a barrier)
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, half = lane >> 5, p = lane & 31;
    const int HW = a.H * a.W, CI16 = a.Cin / 16, CO16 = a.Cout / 16, groups = a.Cout / CT, chunks = CI16;
    const unsigned lds0 = (unsigned)(size_t)(lptr_t)lds;
    // Round 6: bias | 2^-k_c of every output channel -> LDS, once per workgroup.  The epilogue used to load them from memory: ~800 clocks of exposed latency per
    // tile, and the vmcnt wait the compiler puts in front of their first use also waits for the next tile's LDS-DMA (invisible to its counter model).  The first
    // interval's s_waitcnt + barrier orders these stores before any epilogue.
    float *lds_par = reinterpret_cast<float *>(lds + K::OPERAND_BYTES);
    for (int i = tid; i < a.Cout / 4; i += K::THREADS) {
        reinterpret_cast<float4 *>(lds_par)[i] = reinterpret_cast<const float4 *>(a.bias)[i];
        reinterpret_cast<float4 *>(lds_par + a.Cout)[i] = reinterpret_cast<const float4 *>(a.wscale)[i];
    }
    // LDS map: weight buffers 0 | 1, patch buffers 0 | 1 (paired mode: one of each)
    int member_cg = 0;                                                        // (set below, before decode() is first called)
    const bool loader = K::EXTRA ? wave >= G::WAVES : wave < K::LOADERS, compute = wave < G::WAVES;
    const int lw = K::EXTRA ? (wave >= G::WAVES ? wave - G::WAVES : 0) : wave;      // index among the loaders

    const int blk_y = wave / NBX, blk_x = wave - blk_y * NBX;                 // this wavefront's pixel block inside the tile
    // Lane -> pixel inside the block.  2 x 16 blocks (PW = 18: a patch row is 288 bytes = 256 + 32): row 1's columns are rotated by two, so that the four lane
    // groups a ds_read_b128 is served in ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ...) touch disjoint banks -- unrotated, row 1 sits 32 bytes into row 0's
    // banks: SQ_LDS_BANK_CONFLICT was 25 % of the LDS cycles of the 50 x 176 layers (profiles/round5/pmc_summary.json, first pass).  Which lane owns which pixel of
    // its block is free: addresses in and out follow (py, px).
    const int py = blk_y * BH + p / BW, px = blk_x * BW + ((BW == 16 && NBX == 1) ? ((p % 16) + 16 - 2 * (p / 16)) % 16 : p % BW);
    int boff[9];                                                              // group of this lane's pixel under tap s, inside plane (2 * half + term 0)
#pragma unroll
    for (int s = 0; s < 9; ++s) boff[s] = 2 * half * G::PIXP + (py + s / 3) * G::PW + px + s % 3;
    const int wlane = half * CT + p;                                   // this lane's group inside one (tap, term) weight block

    // tile id t: whole-tile schedule = spatial tile * groups + output-channel group (the groups of one spatial tile are neighbours: they read the same patch);
    // stream-K = spatial tile only, the group is the workgroup's MEMBER index inside its gang (below)
    auto decode = [&](int t) {
        Tile c;
        c.cg = SPLIT ? member_cg : t % groups;
        const int sp = SPLIT ? t : t / groups, ty = sp / a.tiles_x;
        c.x0 = (sp - ty * a.tiles_x) * G::TW;
        if (a.stack) {
            const int y0 = ty * G::TH;
            c.n0 = y0 / a.H;
            c.yl0 = y0 - c.n0 * a.H;
            c.yb = a.H - c.yl0;
        } else {
            c.n0 = ty / a.tiles_y;
            c.yl0 = (ty - c.n0 * a.tiles_y) * G::TH;
            c.yb = 1 << 24;
        }
        return c;
    };
    // Patch plan of a tile: the lane's group of DMA instruction (wave + LOADERS * j) of a plane = patch pixel 64 * (wave + LOADERS * j) + lane; its offset (in
    // 16-byte groups) inside plane 0 of interval 0, or -1 for zero padding.  Patch rows (see conv3x3_emu.hip): 0 .. yb rows yl0 - 1 .. H - 1 of image n0,
    // yb + 1 and yb + 2 zero, yb + 3 .. rows 0 .. of image n0 + 1; without a boundary inside the tile (yb >= TH) rows yl0 - 1 .. yl0 + TH.
    struct Plan {
        int off[K::PJ];
        const uint4 *wsrc;
    };
    auto make_plan = [&](const Tile &t) {
        Plan pl;
        // (CT = 32: a 1 KB piece = the two channel halves of one (tap, term): lanes 0-31 / 32-63 fetch 32 of the image's 64 output channels each)
        pl.wsrc = CT == 64 ? a.wt + (size_t)t.cg * chunks * G::WQ_SRC + lane
                           : a.wt + (size_t)(t.cg >> 1) * chunks * G::WQ_SRC + (lane >> 5) * kCoutTile + (t.cg & 1) * 32 + (lane & 31);
#pragma unroll
        for (int j = 0; j < K::PJ; ++j) {
            const int i = (lw + K::LOADERS * j) * 64 + lane;
            const int y = i / G::PW, xq = i - y * G::PW, gx = t.x0 - 1 + xq;
            bool ok = i < G::PIX && xq < G::PWU && gx >= 0 && gx < a.W;
            int gy, img = 0;
            if (t.yb >= G::TH) { gy = t.yl0 - 1 + y; ok = ok && y < G::TH + 2; }
            else if (y <= t.yb) gy = t.yl0 - 1 + y;
            else if (y <= t.yb + 2) { gy = 0; ok = false; }
            else { gy = y - (t.yb + 3); img = 1; ok = ok && t.n0 + 1 < a.N; }
            ok = ok && gy >= 0 && gy < a.H;
            pl.off[j] = ok ? (t.n0 + img) * CI16 * 4 * HW + gy * a.W + gx : -1;
        }
        return pl;
    };
    // DMA operation k of this wavefront for interval c into buffer `slot`: k < WJ a 1 KB piece of the weights, then the four planes of patch piece j
    auto issue_op = [&](const Plan &pl, int c, int slot, int k) {
        if (k < K::WJ) {
            const int ins = lw + K::LOADERS * k;
            if (ins < G::WINS && !SP_ABLATE(1)) dma16(pl.wsrc + (size_t)c * G::WQ_SRC + ins * (64 * kCoutTile / CT), lds0 + slot * G::W_BYTES + ins * 1024);
        } else {
            const int j = (k - K::WJ) / 4, q = (k - K::WJ) % 4, ins = lw + K::LOADERS * j;
            if (ins < G::PINS && !SP_ABLATE(2))
                dma16(pl.off[j] < 0 ? a.zero : a.x + (size_t)pl.off[j] + ((size_t)c * 4 + q) * HW, lds0 + K::NBUF * G::W_BYTES + slot * G::B_BYTES + (q * G::PIXP + ins * 64) * 16);
        }
    };
    auto issue_all = [&](const Plan &pl, int c, int slot) {
#pragma unroll
        for (int k = 0; k < K::OPS; ++k) issue_op(pl, c, slot, k);
    };

    // Persistent workgroups.  Whole tiles g, g + n, ... -- or, stream-K (a.split), the (tile, interval) steps cut into n equal contiguous ranges, so that
    // every workgroup runs the same number of matrix instructions (+-1 interval) whatever the tile count: 192 tiles of 16 intervals on 256 CUs are 12
    // intervals each instead of 16 on three quarters of the chip.  A range may start in the middle of a tile (that share is computed from zero accumulators
    // and PUBLISHED) and may end in the middle of one (the workgroup that opened a tile OWNS it: it adds the shares the following workgroups published --
    // they compute them first thing in their ranges -- and runs the epilogue).  The cut is a pure function of the shape: the summation order, hence the
    // result, is deterministic.  XCD k takes the k-th eighth of the logical ids (neighbouring tiles share input rows and weights in its L2; hand-overs stay
    // between logical neighbours).
    const int n_wg = gridDim.x;
    int g = blockIdx.x;
    if (a.xcd) {
        const int q = n_wg >> 3, r = n_wg & 7, k = g & 7, j = g >> 3;
        g = k * q + (k < r ? k : r) + j;
    }
    // Stream-K runs in GANGS (round 5): the `groups` workgroups g = gang * groups + member that work on the output-channel groups of the SAME spatial tiles take
    // the SAME range of (spatial tile, interval) steps, so they read the same input patch at the same time and the XCD's L2 serves three of the four reads.
    // (Cut per workgroup over the flat tile list -- the first version, and conv3x3_emu.hip's -- neighbouring workgroups sat in different intervals of different
    // tiles: PMC HBM traffic of the shrink header's second convolution 425 MB for 74 MB algorithmic, L2 hit 44 %.)  A share is handed to the workgroup of the
    // same member one gang earlier.
    const int gangs = SPLIT ? n_wg / groups : 1, gang = SPLIT ? g / groups : 0;
    member_cg = SPLIT ? g - gang * groups : 0;
    const long long S_total = (long long)(a.total_tiles / groups) * chunks;
    auto range_start = [&](int j) { return (int)(S_total * j / gangs); };
    const int s0 = SPLIT ? range_start(gang) : 0;
    const int n_local = SPLIT ? range_start(gang + 1) - s0 : ((a.total_tiles - g + n_wg - 1) / n_wg) * chunks;
    if (n_local <= 0) return;
#ifdef SP_TRACE
    if (tid == 0) {
        a.trace[2 * 16 * 64 * 8 + 2 * g] = wall_clock64();
        if (g < 4096) a.trace[2 * 16 * 64 * 8 + 2 * 4096 + g] = (long long)__builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);      // HW_ID: where this workgroup runs (CU, SE, TG slot)
    }
#endif
    int tile = SPLIT ? s0 / chunks : g, chunk0 = SPLIT ? s0 - tile * chunks : 0;
    const int tile_step = SPLIT ? 1 : n_wg;
    Tile cur = decode(tile);
    if constexpr (K::EXTRA != 0) {
        static_assert(!SPLIT, "the loader wavefronts mirror the whole-tile schedule only (their barriers must match the computing wavefronts')");
        if (wave >= G::WAVES) {                                    // a loader wavefront: one barrier per interval like everybody else, nothing but DMA issue
            __builtin_amdgcn_s_setprio(3);
            Plan pl = make_plan(cur);
            issue_all(pl, 0, 0);
            int chunk = 0;
            for (int l = 0; l < n_local; ++l) {
                __builtin_amdgcn_s_waitcnt(0);
                __syncthreads();
                if (l + 1 < n_local) {
                    if (++chunk == chunks) {
                        chunk = 0;
                        tile += tile_step;
                        pl = make_plan(decode(tile));
                    }
                    issue_all(pl, chunk, (l + 1) & 1);
                }
            }
            return;
        }
    }
    const bool loader_here = K::EXTRA ? false : loader;            // (MODE 3 / 7: the computing wavefronts carry no plan and issue nothing)
    Plan plan{};
    if (loader_here) {
        plan = make_plan(cur);
        issue_all(plan, chunk0, 0);
    }
    if constexpr (K::PAIRED) {
        // HW_REG_HW_ID (register 4) bits 16-19 = TG_ID: the workgroup's slot on its CU.  The odd slot outranks the even one at every issue arbitration.
        const unsigned tg = __builtin_amdgcn_s_getreg((3 << 11) | (16 << 6) | 4);
        if (tg & 1) __builtin_amdgcn_s_setprio(2);
    } else if (MODE != 2 && wave >= G::WAVES / 2) __builtin_amdgcn_s_setprio(1);      // (as conv3x3_emu.hip: the later-dispatched half of the wavefronts loses every arbitration otherwise)
    int L = 0;
    bool pending_pub = false;                                      // stream-K (round 6): this workgroup's share has left, its flag goes up behind the next interval's barrier
    while (L < n_local) {
        const int c_begin = SPLIT ? chunk0 : 0, c_end = SPLIT && (n_local - L) < (chunks - c_begin) ? c_begin + (n_local - L) : chunks;
        const bool head = !SPLIT || c_begin == 0, complete = !SPLIT || c_end == chunks;
        // this lane's output pixel: image out_n, row gy, column gx; rows at / past an image boundary read the patch two rows lower
        const bool lower = py >= cur.yb;
        const int out_n = cur.n0 + (lower ? 1 : 0), gy = lower ? py - cur.yb : cur.yl0 + py, gx = cur.x0 + px;
        const int bshift = lower ? 2 * G::PW : 0;
        const bool live = compute && out_n < a.N && gy < a.H && gx < a.W;
        const bool wave_live = compute && __builtin_amdgcn_readfirstlane((int)(cur.n0 * a.H + cur.yl0 + blk_y * BH < (a.stack ? a.N * a.H : cur.n0 * a.H + a.H) && cur.x0 + blk_x * BW < a.W)) != 0;
        const size_t pix = live ? (size_t)gy * a.W + gx : 0;
        const int on = live ? out_n : 0;
        floatx16 acc[NQ], accl[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            acc[q] = floatx16{0};
            accl[q] = floatx16{0};
        }
        uintx4 rraw[8];                                            // the residual of this lane's 32 outputs, fetched behind the barrier of the owner's LAST interval
#pragma unroll                                                     // (stream-K owners: the same registers first carry the next gang's share, fetched beside that interval)
        for (int g8 = 0; g8 < 8; ++g8) rraw[g8] = uintx4{0, 0, 0, 0};
        Tile next = cur;
        Plan nplan = plan;
        int ntile = tile;
        bool share_early = false;
        // (The residual fetch steps a base pointer: two instructions per load instead of ten.  Moving it, or the next tile's plan, behind different taps per
        //  wavefront -- so that a SIMD's other wavefronts keep the matrix pipe fed -- was tried: behind a tap the 32 residual registers are live together with the
        //  prefetched operands, the 12-wavefront geometries spill, and the plan needs a third plan record; no gain measured, dropped.)
        auto fetch_residual = [&]() {
            if (a.res_kind == SP_RES_SP) {                         // h groups | l groups: 8 bytes each per (lane, 8-channel group); consecutive groups lie two planes apart
                const uint2 *rb = reinterpret_cast<const uint2 *>(a.residual) + (((size_t)(on * CO16 + cur.cg * (CT / 16)) * 4) * HW + pix) * 2 + half;
                const size_t step = 4 * (size_t)HW, lo = 2 * (size_t)HW;
#pragma unroll
                for (int g8 = 0; g8 < NG8; ++g8) {
                    const uint2 h = rb[g8 * step], l = rb[g8 * step + lo];
                    rraw[g8] = uintx4{h.x, h.y, l.x, l.y};
                }
            } else if (a.res_kind == SP_RES_NHWC) {
                const uint4 *rp = reinterpret_cast<const uint4 *>(static_cast<const float *>(a.residual) + ((size_t)on * HW + pix) * a.Cout + cur.cg * CT + 4 * half);
#pragma unroll
                for (int g8 = 0; g8 < NG8; ++g8) {
                    const uint4 t = rp[2 * g8];
                    rraw[g8] = uintx4{t.x, t.y, t.z, t.w};
                }
            }
        };
        for (int chunk = c_begin; chunk < c_end; ++chunk, ++L) {
            SP_STAMP(0);
            __builtin_amdgcn_s_waitcnt(0);
            SP_STAMP(1);
            __syncthreads();
            SP_STAMP(2);
            if (SPLIT && pending_pub) {                            // every wavefront's share stores are acknowledged (its s_waitcnt above) and all have passed the barrier
                if (tid == 0) __hip_atomic_store(a.flags + g, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                pending_pub = false;
            }
            const bool more = L + 1 < n_local;
            int nc = chunk + 1;
            if (more && nc == chunks) {                            // the next interval opens this workgroup's next tile
                nc = 0;
                ntile = tile + tile_step;
                next = decode(ntile);
                if (loader_here) nplan = make_plan(next);
            }
            if constexpr (SPLIT) {
                // Round 6: the owner of a tile its range does not finish looks for the next gang's share BEFORE its own last interval: if the flag is up (the normal
                // case: that share was computed first thing in the neighbour's range), the eight 16-byte loads fly beside this interval's matrix steps.
                if (!complete && head && chunk == c_end - 1) {
                    const int j = (gang + 1) * groups + member_cg;
                    if (tid == 0) {
                        const int up = __hip_atomic_load(a.flags + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (up) __hip_atomic_store(a.flags + j, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        s_share_ready = up;
                    }
                    __syncthreads();
                    share_early = s_share_ready != 0;
                    if (share_early && wave_live) {
                        const floatx4 *slot4 = reinterpret_cast<const floatx4 *>(a.partial + ((size_t)j * G::WAVES + wave) * (32 * 64)) + lane;
#pragma unroll
                        for (int q4 = 0; q4 < 8; ++q4) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(rraw[q4]) : "v"(slot4 + q4 * 64) : "memory");
                    }
                }
            }
            constexpr bool kPaired = K::PAIRED;
            const int slot_cur = kPaired ? 0 : (L & 1), slot_next = kPaired ? 0 : ((L + 1) & 1);
            if (!K::INTERLEAVED && !kPaired && more && loader_here) issue_all(nplan, nc, slot_next);
            if (!kPaired && chunk == c_end - 1 && head && complete && wave_live && !SP_ABLATE(8)) fetch_residual();      // (an owner that still has to take shares in fetches it behind them)
            SP_STAMP(3);
            if (wave_live && !SP_ABLATE(4)) {
                const uint4 *bq = reinterpret_cast<const uint4 *>(lds + K::NBUF * G::W_BYTES + slot_cur * G::B_BYTES) + bshift, *wq = reinterpret_cast<const uint4 *>(lds + slot_cur * G::W_BYTES) + wlane;
                auto load_b = [&](int s, halfx8 (&b)[2]) {
#pragma unroll
                    for (int t = 0; t < 2; ++t) b[t] = __builtin_bit_cast(halfx8, bq[t * G::PIXP + boff[s]]);
                };
                auto load_w = [&](int s, halfx8 (&w)[NQ][2]) {
#pragma unroll
                    for (int q = 0; q < NQ; ++q)
#pragma unroll
                        for (int t = 0; t < 2; ++t) w[q][t] = __builtin_bit_cast(halfx8, wq[((s * 2 + t) * 2) * CT + q * 32]);
                };
                halfx8 bc[2], wc[NQ][2];
                if constexpr (!kPaired) {
                    load_b(0, bc);
                    load_w(0, wc);
                }
#pragma unroll
                for (int s = 0; s < 9; ++s) {
                    // Round 6: PROGRESS-BASED issue priority.  A SIMD's two or three wavefronts share one matrix pipe; with fixed priorities the favoured one runs
                    // through its nine taps first and the last one finishes alone, at the pace of its own operand reads (12-wavefront geometries: the three
                    // wavefronts of a SIMD end their steps 3100 / 5300 / 6400 clocks after the barrier for 5184 clocks of matrix work; tools/trace_conv_sp.py
                    // ALLWAVES=1).  A wavefront now lowers its own priority as it advances (taps 0-2: 2, 3-5: 1, 6-8: 0): whoever is behind wins the arbitration.
                    if (a.prio_mode == 1) {
                        if (s == 0) __builtin_amdgcn_s_setprio(2);
                        if (s == 3) __builtin_amdgcn_s_setprio(1);
                        if (s == 6) __builtin_amdgcn_s_setprio(0);
                    } else if (a.prio_mode == 2) {                 // ... and the later-dispatched wavefronts (they lose every tie) step down one tap later
                        const int late = wave >= (2 * G::WAVES + 2) / 3 ? 2 : wave >= G::WAVES / 3 ? 1 : 0;      // thirds of the workgroup = the wavefronts of one SIMD, oldest first
                        if (s == 0) __builtin_amdgcn_s_setprio(2);
                        if (s == 3 && late == 0) __builtin_amdgcn_s_setprio(1);
                        if (s == 4 && late != 0) __builtin_amdgcn_s_setprio(1);
                        if (s == 6 && late == 0) __builtin_amdgcn_s_setprio(0);
                        if (s == 7 && late == 1) __builtin_amdgcn_s_setprio(0);
                        if (s == 8 && late == 2) __builtin_amdgcn_s_setprio(0);
                    }
                    halfx8 bn[2], wn[NQ][2];
                    if constexpr (kPaired) {                       // four wavefronts per SIMD hide the LDS latency; no second operand set in the 128 registers
                        load_b(s, bc);
                        load_w(s, wc);
                    } else if (s + 1 < 9) {                        // operands of the next tap are in flight while this tap's matrix instructions issue
                        load_b(s + 1, bn);
                        load_w(s + 1, wn);
                    }
                    if constexpr (MODE == 4) {
                        if (more) {
#pragma unroll
                            for (int k = s; k < K::OPS; k += 9) issue_op(nplan, nc, slot_next, k);
                        }
                    }
#pragma unroll
                    for (int q = 0; q < NQ; ++q) accl[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wc[q][0], bc[1], accl[q], 0, 0, 0);      // w_h x_l'
#pragma unroll
                    for (int q = 0; q < NQ; ++q) accl[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wc[q][1], bc[0], accl[q], 0, 0, 0);      // w_l' x_h
#pragma unroll
                    for (int q = 0; q < NQ; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wc[q][0], bc[0], acc[q], 0, 0, 0);        // w_h x_h
                    if constexpr (MODE == 1) {
                        if (more) {
#pragma unroll
                            for (int k = s; k < K::OPS; k += 9) issue_op(nplan, nc, slot_next, k);
                        }
                    }
                    if constexpr (MODE == 5) {
                        if (more) {
#pragma unroll
                            for (int k = 2 * s; k < 2 * s + 2; ++k)
                                if (k < K::OPS) issue_op(nplan, nc, slot_next, k);
                        }
                    }
                    if (!kPaired && s + 1 < 9) {
#pragma unroll
                        for (int t = 0; t < 2; ++t) {
                            bc[t] = bn[t];
#pragma unroll
                            for (int q = 0; q < NQ; ++q) wc[q][t] = wn[q][t];
                        }
                    }
                }
            } else if (K::INTERLEAVED && more) {
                issue_all(nplan, nc, slot_next);
            }
            if constexpr (kPaired) {                               // everyone has read the buffers: the next interval's operands take their place (the partner workgroup computes meanwhile)
                SP_STAMP(5);
                __syncthreads();
                SP_STAMP(6);
                if (more && loader_here) issue_all(nplan, nc, 0);
            }
            SP_STAMP(4);
        }
        if constexpr (K::PAIRED) {                                 // (no residual prefetch behind the last interval: its 32 registers do not fit beside the K loop's at 128)
            if (head && wave_live && !SP_ABLATE(8)) fetch_residual();
        }
        chunk0 = 0;                                                // every later segment of this range opens its tile
        SP_STAMP_AT(5, L - 1);                                     // (epilogue stamps land in the record of the tile's last interval: 5 start, 6 residual in float, 7 stored)
        // the segment's sums, both accumulators joined: t = acc + 2^-10 accl
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[q][e] = fmaf(accl[q][e], coalign::kSp16LowInv, acc[q][e]);
        // Stream-K hand-over, as conv3x3_emu.hip: agent-scope write-through stores / L2-bypassing loads (relaxed atomics), no fences.  Producer: partial
        // sums leave as write-through stores; s_waitcnt(0) = acknowledged; the barrier = true for every wavefront; only then the flag.  Consumer: one lane
        // spins on the flag (and clears it: exactly one consumer per flag and launch, launches on a stream are ordered), the barrier releases the
        // workgroup, the partial sums are read by loads issued after it.  Guarded by tests/test_round5_gpu.py::test_conv3x3_sp_stream_k_*.
        if constexpr (SPLIT) {
        if (!head) {
            if (wave_live) {
                // round 6: the share leaves as 8 agent-scope 16-byte stores per lane (global_store_dwordx4 sc1: the cache policy of the 4-byte relaxed atomic stores
                // they replace -- write-through past the XCD's L2 -- at a quarter of the instructions; 1 KB per wavefront instruction)
                floatx4 *slot4 = reinterpret_cast<floatx4 *>(a.partial + ((size_t)g * G::WAVES + wave) * (32 * 64)) + lane;
#pragma unroll
                for (int q4 = 0; q4 < 8; ++q4) {
                    const floatx4 v = {acc[q4 / 4][4 * (q4 % 4)], acc[q4 / 4][4 * (q4 % 4) + 1], acc[q4 / 4][4 * (q4 % 4) + 2], acc[q4 / 4][4 * (q4 % 4) + 3]};
                    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(slot4 + q4 * 64), "v"(v) : "memory");
                }
            }
            // (round 6: the flag is raised behind the NEXT interval's wait + barrier -- the stores' write-through latency, ~2 us, runs beside that interval's matrix
            //  steps instead of in front of them; a range that ends with its share raises it after the loop)
            pending_pub = true;
            cur = next;
            plan = nplan;
            tile = ntile;
            continue;
        }
        if (!complete) {                                  // owner of a tile this range does not finish: the following workgroups' shares
            int rem = chunks - c_end, jg0 = gang + 1;
            if (share_early) {                                     // fetched beside the last interval
                if (wave_live) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                    for (int q4 = 0; q4 < 8; ++q4)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const unsigned u = rraw[q4][e];        // (through a local: see the epilogue)
                            acc[q4 / 4][4 * (q4 % 4) + e] += __builtin_bit_cast(float, u);
                        }
#pragma unroll
                    for (int g8 = 0; g8 < 8; ++g8) rraw[g8] = uintx4{0, 0, 0, 0};
                }
                rem -= range_start(jg0 + 1) - range_start(jg0);
                ++jg0;
            }
            for (int jg = jg0; rem > 0; ++jg) {
                const int j = jg * groups + member_cg;             // the same member of the next gang
                if (tid == 0) {
                    while (__hip_atomic_load(a.flags + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) __builtin_amdgcn_s_sleep(2);
                    __hip_atomic_store(a.flags + j, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                __syncthreads();
                if (wave_live) {
                    const floatx4 *slot4 = reinterpret_cast<const floatx4 *>(a.partial + ((size_t)j * G::WAVES + wave) * (32 * 64)) + lane;
                    floatx4 sh[8];
#pragma unroll
                    for (int q4 = 0; q4 < 8; ++q4) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(sh[q4]) : "v"(slot4 + q4 * 64) : "memory");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                    for (int q4 = 0; q4 < 8; ++q4)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[q4 / 4][4 * (q4 % 4) + e] += sh[q4][e];
                }
                rem -= range_start(jg + 1) - range_start(jg);
            }
            if (wave_live && !SP_ABLATE(8)) fetch_residual();      // (not prefetched behind the last interval: its 32 registers are the early share's there)
        }
        }
        // ---- epilogue: y = (acc + 2^-10 accl) * 2^-k_c + (residual + bias), ReLU, stored as an SP map or as channels-last fp32.  It is VALU bound (two or
        // three wavefronts per SIMD all arrive here together, the matrix pipe idles): ~700 vector instructions per wavefront in the first version = 8400 cycles
        // per tile on the 12-wavefront geometries (profiles/round5/experiments/conv_sp_interval_timeline_stage1_stage2.txt, the "gap" of a tile's last interval).
        // Hence: the residual is converted to float ONCE, outside the channel-group loop (no per-group branches on its kind), ReLU is one v_max against a floor,
        // the range check one running maximum, the store address a running pointer, each group's bias / scale loads are issued one group ahead.
        const float4 *bias4 = reinterpret_cast<const float4 *>(lds_par + cur.cg * CT + 4 * half), *winv4 = reinterpret_cast<const float4 *>(lds_par + a.Cout + cur.cg * CT + 4 * half);
        float rr[8][4];
        if constexpr (K::PAIRED) {
            // (converted per channel group below: a float copy of all 32 residual values beside the 64 accumulators does not fit the paired mode's 128 registers)
        } else if (a.res_kind == SP_RES_SP) {
#pragma unroll
            for (int g8 = 0; g8 < NG8; ++g8) {
                const halfx4 h = __builtin_bit_cast(halfx4, uint2{rraw[g8].x, rraw[g8].y}), l = __builtin_bit_cast(halfx4, uint2{rraw[g8].z, rraw[g8].w});
#pragma unroll
                for (int j = 0; j < 4; ++j) rr[g8][j] = coalign::sp16_join(h[j], l[j]);
            }
        } else {                                                   // channels-last float32, or none (rraw is zero)
#pragma unroll
            for (int g8 = 0; g8 < NG8; ++g8) {
                const unsigned u0 = rraw[g8].x, u1 = rraw[g8].y, u2 = rraw[g8].z, u3 = rraw[g8].w;      // (bit-casting a vector ELEMENT directly is miscompiled by hipcc 7.2 -- pillar_sparse.hip swap32: go through locals)
                rr[g8][0] = __builtin_bit_cast(float, u0); rr[g8][1] = __builtin_bit_cast(float, u1);
                rr[g8][2] = __builtin_bit_cast(float, u2); rr[g8][3] = __builtin_bit_cast(float, u3);
            }
        }
        SP_STAMP_AT(6, L - 1);
        const float floor_v = a.relu ? 0.f : -__builtin_inff();   // (a NaN leaves as the floor: v_max returns the other operand, as fmaxf(v, 0) always did under ReLU)
        float vmax = 0.f;
        // first store position of this lane: group 0 of the tile's 64 channels; every further 8-channel group lies 2 planes (SP) / 8 floats (channels-last) on
        uint4 *ysp = static_cast<uint4 *>(OUT == SP_OUT_BOTH ? a.y2 : a.y) + ((size_t)(on * CO16 + cur.cg * (CT / 16)) * 4 + half) * HW + pix;
        float4 *ycl = reinterpret_cast<float4 *>(static_cast<float *>(a.y) + ((size_t)on * HW + pix) * a.Cout + cur.cg * CT + 4 * half);
        const size_t sp_step = 2 * (size_t)HW;
        float4 b4 = bias4[0], i4 = winv4[0];
#pragma unroll
        for (int g8 = 0; g8 < NG8; ++g8) {                         // 8 (CT = 32: 4) groups of 4 consecutive channels per lane: channel = 8 g8 + 4 half + j
            const float bb[4] = {b4.x, b4.y, b4.z, b4.w}, ii[4] = {i4.x, i4.y, i4.z, i4.w};
            if (g8 + 1 < NG8) {
                b4 = bias4[2 * (g8 + 1)];
                i4 = winv4[2 * (g8 + 1)];
            }
            float v[4];
            if constexpr (K::PAIRED) {
                if (a.res_kind == SP_RES_SP) {
                    const halfx4 h = __builtin_bit_cast(halfx4, uint2{rraw[g8].x, rraw[g8].y}), l = __builtin_bit_cast(halfx4, uint2{rraw[g8].z, rraw[g8].w});
#pragma unroll
                    for (int j = 0; j < 4; ++j) rr[g8][j] = coalign::sp16_join(h[j], l[j]);
                } else {
                    const unsigned u0 = rraw[g8].x, u1 = rraw[g8].y, u2 = rraw[g8].z, u3 = rraw[g8].w;      // (bit-casting a vector ELEMENT directly is miscompiled by hipcc 7.2 -- pillar_sparse.hip swap32: go through locals)
                rr[g8][0] = __builtin_bit_cast(float, u0); rr[g8][1] = __builtin_bit_cast(float, u1);
                    rr[g8][2] = __builtin_bit_cast(float, u2); rr[g8][3] = __builtin_bit_cast(float, u3);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int q = g8 / 4, e = 4 * (g8 % 4) + j;
                v[j] = fmaxf(acc[q][e] * ii[j] + (rr[g8][j] + bb[j]), floor_v);
            }
            if constexpr (OUT == SP_OUT_NHWC || OUT == SP_OUT_BOTH) {
                if (live && !SP_ABLATE(16)) ycl[2 * g8] = float4{v[0], v[1], v[2], v[3]};      // (16 bytes per pixel and instruction: streaming stores measured +16 % here -- partial lines; the SP map's 512-byte runs below gain 4 %)
            }
            if constexpr (OUT == SP_OUT_SP || OUT == SP_OUT_BOTH) {
                vmax = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fmaxf(fabsf(v[2]), fabsf(v[3])), vmax));
                unsigned h01, l01, h23, l23;
                coalign::sp16_split2(v[0], v[1], h01, l01);
                coalign::sp16_split2(v[2], v[3], h23, l23);
                swap32(h01, l01);          // lanes 0-31: h of channels 0,1 | 4,5 of the 8-channel group; lanes 32-63: l of the same channels
                swap32(h23, l23);
                if (live && !SP_ABLATE(16)) {                              // plane = 2 * channel half + term: lanes 32-63 hold term 1 (the `half` in ysp)
                    if (a.stream_out) coalign::store_stream(ysp, uint4{h01, h23, l01, l23});
                    else *ysp = uint4{h01, h23, l01, l23};
                }
                ysp += sp_step;
            }
        }
        if constexpr (OUT == SP_OUT_SP || OUT == SP_OUT_BOTH) {
            if (a.range_flag && live && vmax > 65504.f) atomicOr(a.range_flag, 1);
        }
        SP_STAMP_AT(7, L - 1);
        cur = next;
        plan = nplan;
        tile = ntile;
    }
    if (SPLIT && pending_pub) {                                    // the range ended with a share: nothing left to hide the stores behind
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        if (tid == 0) __hip_atomic_store(a.flags + g, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#ifdef SP_TRACE
    if (tid == 0) a.trace[2 * 16 * 64 * 8 + 2 * g + 1] = wall_clock64();
#endif
}

// fp32 (NCHW or channels-last) -> SP map and back: the entry / exit of a chain of SP layers where no kernel epilogue does it, and the tests' yardstick
__global__ void sp_pack_kernel(const float *__restrict__ x, uint4 *__restrict__ y, int N, int C, int HW, int in_nhwc, int *range_flag) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;      // (n, c16, pixel)
    const int C16 = C / 16;
    if (i >= (size_t)N * C16 * HW) return;
    const int pixel = (int)(i % HW), c16 = (int)((i / HW) % C16), n = (int)(i / ((size_t)HW * C16));
    float v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = in_nhwc ? x[((size_t)n * HW + pixel) * C + c16 * 16 + k] : x[((size_t)n * C + c16 * 16 + k) * HW + pixel];
    bool big = false;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        unsigned h[4], l[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            coalign::sp16_split2(v[8 * hh + 2 * k], v[8 * hh + 2 * k + 1], h[k], l[k]);
            big = big || fabsf(v[8 * hh + 2 * k]) > 65504.f || fabsf(v[8 * hh + 2 * k + 1]) > 65504.f;
        }
        const size_t base = ((size_t)(n * C16 + c16) * 4 + hh * 2) * HW + pixel;
        y[base] = uint4{h[0], h[1], h[2], h[3]};
        y[base + HW] = uint4{l[0], l[1], l[2], l[3]};
    }
    if (range_flag && big) atomicOr(range_flag, 1);
}

__global__ void sp_unpack_kernel(const uint4 *__restrict__ x, float *__restrict__ y, int N, int C, int HW, int out_nhwc) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int C16 = C / 16;
    if (i >= (size_t)N * C16 * HW) return;
    const int pixel = (int)(i % HW), c16 = (int)((i / HW) % C16), n = (int)(i / ((size_t)HW * C16));
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        const size_t base = ((size_t)(n * C16 + c16) * 4 + hh * 2) * HW + pixel;
        const halfx8 h = __builtin_bit_cast(halfx8, x[base]), l = __builtin_bit_cast(halfx8, x[base + HW]);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float v = coalign::sp16_join(h[k], l[k]);
            if (out_nhwc) y[((size_t)n * HW + pixel) * C + c16 * 16 + 8 * hh + k] = v;
            else y[((size_t)n * C + c16 * 16 + 8 * hh + k) * HW + pixel] = v;
        }
    }
}

struct SpLaunch {          // what the host needs to know about one (shape, geometry) pair
    int grid, split;
    size_t flag_bytes, ws_bytes;
};

// split_policy: 0 = by the rule below, 1 = never, 2 = whenever possible (laboratory)
template <int BH, int BW, int NPB, int NBX, int MODE, int CT = 64>
int launch_geo(SpArgs a, int out_kind, int split_policy, void *workspace, size_t workspace_bytes, hipStream_t s, SpLaunch *query) {
    using G = Geo<BH, BW, NPB, NBX, CT>;
    constexpr int kMaxDev = 16;
    static int cus[kMaxDev] = {0};                                    // per device: the function attribute belongs to the device's code object
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) dev = 0;
    using KW = Work<BH, BW, NPB, NBX, MODE, CT>;
    constexpr bool CAN_SPLIT = NPB == 8 && CT == 64 && MODE != 3 && MODE != 6 && MODE != 7;    // (the hand-over code needs the 8-wavefront geometries' register budget; the loader wavefront of MODE 3 mirrors whole tiles only; the paired mode runs whole tiles)
    auto k_sp = conv3x3_sp_kernel<BH, BW, NPB, NBX, SP_OUT_SP, MODE, false, CT>;
    auto k_cl = conv3x3_sp_kernel<BH, BW, NPB, NBX, SP_OUT_NHWC, MODE, false, CT>;
    auto k_sp_s = conv3x3_sp_kernel<BH, BW, NPB, NBX, SP_OUT_SP, MODE, CAN_SPLIT, CT>;
    auto k_cl_s = conv3x3_sp_kernel<BH, BW, NPB, NBX, SP_OUT_NHWC, MODE, CAN_SPLIT, CT>;
    auto k_both = conv3x3_sp_kernel<BH, BW, NPB, NBX, SP_OUT_BOTH, MODE, false, CT>;
    auto k_both_s = conv3x3_sp_kernel<BH, BW, NPB, NBX, SP_OUT_BOTH, MODE, CAN_SPLIT, CT>;
    if (!cus[dev]) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) prop.multiProcessorCount = 256;
        if (!query) {
            for (const void *fn : {reinterpret_cast<const void *>(k_sp), reinterpret_cast<const void *>(k_cl), reinterpret_cast<const void *>(k_sp_s), reinterpret_cast<const void *>(k_cl_s),
                                   reinterpret_cast<const void *>(k_both), reinterpret_cast<const void *>(k_both_s)}) {
                const int rc = coalign::hip_call(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)KW::LDS_BYTES));
                if (rc != COALIGN_OK) {
                    (void)hipGetLastError();
                    return rc;
                }
            }
            cus[dev] = prop.multiProcessorCount;
        }
        if (query && !cus[dev]) {                                     // (a size query before the first launch: same CU count, attributes set at the launch)
            SpArgs b = a;
            (void)b;
        }
    }
    hipDeviceProp_t prop2;
    const int n_cu = cus[dev] ? cus[dev] : (hipGetDeviceProperties(&prop2, dev) == hipSuccess && prop2.multiProcessorCount > 0 ? prop2.multiProcessorCount : 256);
    a.stack = (a.N > 1 && G::TH <= a.H) ? 1 : 0;
    a.tiles_x = (a.W + G::TW - 1) / G::TW;
    a.tiles_y = (a.H + G::TH - 1) / G::TH;                            // per image (not stacked)
    const int row_tiles = a.stack ? (a.N * a.H + G::TH - 1) / G::TH : a.N * a.tiles_y;
    a.total_tiles = a.tiles_x * row_tiles * (a.Cout / CT);
    int slots = n_cu * (KW::PAIRED ? 2 : 1);                           // 111-147 KB of LDS: one workgroup per CU (paired mode: 65-70 KB, two)
    {   // laboratory switch: fewer persistent workgroups than CUs (what a launch leaves free, the other frame's kernels take)
        const int cap = coalign::lab_env("COALIGN_SP_SLOTS", 0);
        if (cap > 0 && cap < slots) slots = cap;
    }
    const int chunks = a.Cin / 16;
    // Stream-K pays where whole tiles leave the last round badly filled AND a workgroup's range spans whole tiles, i.e. at most one hand-over per workgroup
    // (the shrink header: 572 tiles of 16 intervals = three rounds at 74 %: 132 -> 125 us).  Measured where a tile is cut into several shares
    // (profiles/round5/conv_sp_layers.json): 5 x 256 x 25 x 88 (192 tiles on 256 CUs, 12 of 16 intervals each) 42.6 -> 49.7 us, 2 x 256 x 25 x 63 (56 tiles)
    // 38.6 -> 67.9 us -- a hand-over (32 L2-bypassing 4-byte stores and loads per lane + the flag) costs more than the intervals it saves.
    const int rounds = (a.total_tiles + slots - 1) / slots;
    const long long steps = (long long)a.total_tiles * chunks;
    bool split = CAN_SPLIT && chunks >= 16 && a.total_tiles > slots && (long long)a.total_tiles * 100 < (long long)rounds * slots * 88;
    // Round 6: the hand-over became cheap -- a share is eight agent-scope 16-byte stores per lane (32 four-byte ones before), its flag goes up behind the publisher's
    // NEXT interval, the owner fetches it beside its own last interval -- and with it cutting an UNDER-FILLED single round pays alone on the GPU (5 x 256 x 25 x 88,
    // 192 tiles: 43.4-44.4 -> 41.8-42.8 us; 2 x 256 x 25 x 63, 56 tiles: 39.2 -> 27 us; round 5's hand-over: 42.6 -> 49.7 and 38.6 -> 67.9).  It is NOT done:
    // (1) inside the two-stream frame pipeline the cut stage-3 layers cost 4.5 % frames/s (594-596 against 621-624, same box, alternating: a launch that owns every
    // CU leaves the other frame's kernels nothing, and its owners spin); (2) the cut depends on the tile count, i.e. on the BATCH -- an agent's maps would differ in
    // the last bits between one rank encoding five agents and five ranks encoding one each, and the sharded runs are bit-identical to the single-rank run by
    // construction (tests/test_sharded_gpu.py caught it).  The shrink header (always one image) keeps its cut and gains the cheaper hand-over: 124.8 -> 117-119 us.
    // (profiles/round6/experiments/conv_sp_stream_k_handover.txt; split_policy 2 = the laboratory's forced cut)
    if (split_policy == 1) split = false;
    if (split_policy == 2) split = CAN_SPLIT && chunks >= 2 && steps >= slots;
    const int n_groups = a.Cout / CT;
    if (split && (slots < n_groups || steps / n_groups < slots / n_groups)) split = false;      // (every gang needs at least one step)
    SpLaunch l;
    l.split = split ? 1 : 0;
    l.grid = split ? slots / n_groups * n_groups : (a.total_tiles < slots ? a.total_tiles : slots);      // stream-K: whole gangs (one workgroup per output-channel group)
    if (!split && a.total_tiles > slots && coalign::lab_env("COALIGN_SP_BALANCE", 0)) {      // laboratory: as many workgroups as give every one the same number of tiles (462 tiles: 231 x 2 instead of 206 x 2 + 50 x 1)
        const int r = (a.total_tiles + slots - 1) / slots;
        l.grid = (a.total_tiles + r - 1) / r;
    }
    l.flag_bytes = split ? coalign::align_up((size_t)(l.grid + 4) * sizeof(int), 256) : 0;
    l.ws_bytes = split ? l.flag_bytes + (size_t)(l.grid + 4) * G::WAVES * 32 * 64 * sizeof(float) : 0;
    if (query) {
        *query = l;
        return COALIGN_OK;
    }
    if (split) {
        if (!workspace) return COALIGN_ERR_NULL_POINTER;
        if (workspace_bytes < l.ws_bytes || (reinterpret_cast<uintptr_t>(workspace) & 15)) return COALIGN_ERR_WORKSPACE;
        a.split = 1;
        a.flags = static_cast<int *>(workspace);
        a.partial = reinterpret_cast<float *>(static_cast<char *>(workspace) + l.flag_bytes);
    }
    constexpr int kThreads = Work<BH, BW, NPB, NBX, MODE, CT>::THREADS;
    if (split) hipLaunchKernelGGL(out_kind == SP_OUT_SP ? k_sp_s : out_kind == SP_OUT_BOTH ? k_both_s : k_cl_s, dim3(l.grid), dim3(kThreads), KW::LDS_BYTES, s, a);
    else hipLaunchKernelGGL(out_kind == SP_OUT_SP ? k_sp : out_kind == SP_OUT_BOTH ? k_both : k_cl, dim3(l.grid), dim3(kThreads), KW::LDS_BYTES, s, a);
    return COALIGN_OK;
}

template <int MODE>
int launch_mode(int geo, const SpArgs &a, int out_kind, int split_policy, void *ws, size_t ws_bytes, hipStream_t s, SpLaunch *query) {
    if constexpr (MODE == 5) {                                        // front-loaded DMA issue: the product carries it for the 8-wavefront geometries only (dispatch_sp)
        switch (geo) {
            case 81: return launch_geo<1, 32, 8, 1, MODE>(a, out_kind, split_policy, ws, ws_bytes, s, query);
            case 148: return launch_geo<4, 8, 8, 4, MODE>(a, out_kind, split_policy, ws, ws_bytes, s, query);
#if defined(COALIGN_LAB) || defined(SP_TRACE)
            case 121: return launch_geo<1, 32, 12, 1, MODE>(a, out_kind, split_policy, ws, ws_bytes, s, query);
            case 124: return launch_geo<2, 16, 12, 1, MODE>(a, out_kind, split_policy, ws, ws_bytes, s, query);
#endif
            default: return COALIGN_ERR_UNSUPPORTED;
        }
    } else if constexpr (MODE == 6 || MODE == 7) {                    // two workgroups per CU / four loader wavefronts: the 8-wavefront geometries
        switch (geo) {
            case 81: return launch_geo<1, 32, 8, 1, MODE>(a, out_kind, split_policy, ws, ws_bytes, s, query);   // 8 rows x 32 columns
            case 84: return launch_geo<2, 16, 8, 1, MODE>(a, out_kind, split_policy, ws, ws_bytes, s, query);   // 16 x 16 (2 x 16 blocks): maps 16 (mod 32) wide
            case 148: return launch_geo<4, 8, 8, 4, MODE>(a, out_kind, split_policy, ws, ws_bytes, s, query);   // 8 x 32 in 4 x 8 blocks
            default: return COALIGN_ERR_UNSUPPORTED;
        }
    } else {
        switch (geo) {
            case 81: return launch_geo<1, 32, 8, 1, MODE>(a, out_kind, split_policy, ws, ws_bytes, s, query);       // 8 rows x 32 columns
            case 121: return launch_geo<1, 32, 12, 1, MODE>(a, out_kind, split_policy, ws, ws_bytes, s, query);     // 12 x 32
            case 124: return launch_geo<2, 16, 12, 1, MODE>(a, out_kind, split_policy, ws, ws_bytes, s, query);     // 24 x 16 (2 x 16 blocks)
            case 148: return launch_geo<4, 8, 8, 4, MODE>(a, out_kind, split_policy, ws, ws_bytes, s, query);       // 8 x 32 in 4 x 8 blocks, four block columns
            case 326: return launch_geo<4, 8, 12, 6, MODE, 32>(a, out_kind, split_policy, ws, ws_bytes, s, query);  // (round 6) 8 x 48 in 4 x 8 blocks x 32 output channels, 12 wavefronts
            default: return COALIGN_ERR_UNSUPPORTED;
        }
    }
}

constexpr int kDefaultMode = 1;

// geometry code: 0 = chosen from the map size (the rules measured for conv3x3_emu.hip's fp16 mode), else 81 / 121 / 124 / 148 as there;
// + 1000 * (issue mode + 1) and + 100000 * split policy are laboratory switches (the product library carries kDefaultMode only)
int dispatch_sp(const SpArgs &a, int out_kind, int geometry, void *ws, size_t ws_bytes, hipStream_t s, SpLaunch *query) {
    const int split_policy = geometry / 100000;
    geometry %= 100000;
    int geo = geometry % 1000, mode = geometry >= 1000 ? geometry / 1000 - 1 : kDefaultMode;
    if (geo == 0) {
        if (a.W % 32 == 16 && a.H > 26 && a.H <= 52 && a.N * a.H >= 24) geo = 124;
        else if (a.H >= 64) geo = (a.N == 1 && a.Cin >= 128) ? 81 : 121;      // (one image, long tiles -- the shrink header: the 8-wavefront geometry, which can split)
        else if (a.H >= 8 && a.H <= 32 && a.W % 32 > 0 && a.W % 32 <= 24) geo = 148;
        else geo = 81;
        // Round 6: a 12-wavefront geometry that leaves more than a quarter of the CUs without a tile loses to the 8-wavefront one whose (smaller) tiles still fit one
        // round: 2 x 64 -> 64 @ 100 x 252 (DAIR stage 1: 136 tiles of 12 x 32 against 200 of 8 x 32 on 256 CUs) 21.5 -> 17.0 us.  Same sums in the same order:
        // the geometries are bit-equal (tests/test_round5_gpu.py).
        static int n_cu = 0;
        if (!n_cu) {
            int dev = 0;
            hipDeviceProp_t prop;
            n_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
        }
        const int rows = a.N * a.H, groups = a.Cout / kCoutTile;
        if (geo == 121 || geo == 124) {
            const int th12 = geo == 121 ? 12 : 24, tw12 = geo == 121 ? 32 : 16, alt = (a.W % 32 == 0 || a.W % 32 > 24) ? 81 : 148;
            const long long t12 = (long long)((rows + th12 - 1) / th12) * ((a.W + tw12 - 1) / tw12) * groups;
            const long long t8 = (long long)((rows + 7) / 8) * ((a.W + 31) / 32) * groups;
            if (t12 * 4 < 3LL * n_cu && t8 <= n_cu) geo = alt;
        }
        // Round 6, measured and NOT made the rule: an 8-wavefront geometry whose 64-channel tiles leave a fifth of the CUs idle, where 8 x 48 tiles of 32 output
        // channels still fit one round (geometry 326: twelve wavefronts of half the matrix work each; 5 x 256 x 25 x 88: 192 tiles of 8 wavefronts on 256 CUs ->
        // 256 tiles of 12).  Alone on the GPU the layer goes 43.1 -> 39.4 us (2 x 256 x 25 x 63: 39.3 -> 31.4) and one frame in flight 1.99 -> 1.90 ms, but a
        // wavefront that owns 32 x 32 outputs reads 1.33 LDS operands per matrix instruction instead of 1.0 -- an interval is ~4900 clocks for 2592 of matrix work
        // (64-channel tiles: 4600 for 3456) -- and the launch holds every CU: the two-stream frame pipeline LOSES 1.6 % (646.6 against 657.2 frames/s, same box,
        // alternating).  Same sums in the same order per output value: bit-equal to the other geometries (tests/test_s2_gpu.py).  A latency-bound caller selects
        // it per map size (coalign_amd/ops.py: COALIGN_SP_GEO="25x88:326"); the laboratory build's COALIGN_SP_CT32=1 makes it the rule.
        if ((geo == 81 || geo == 148) && a.H >= 8 && coalign::lab_env("COALIGN_SP_CT32", 0)) {
            const long long t64 = (long long)((rows + 7) / 8) * ((a.W + 31) / 32) * groups;
            const long long t32 = (long long)((rows + 7) / 8) * ((a.W + 47) / 48) * groups * 2;
            if (t64 * 5 <= 4LL * n_cu && t32 <= n_cu) geo = 326;
        }
    }
    // Round 6: the 8-wavefront geometries issue their LDS-DMA FRONT-LOADED (mode 5: two instructions behind each of the first taps instead of one behind every tap):
    // with the progress-based priority the wavefronts of a SIMD finish together, and the interval's closing s_waitcnt no longer waits ~300 clocks for a piece issued
    // behind the last tap -- stage 3 44.6-45.4 -> 43.4-43.8 us, shrink header 2nd 120.0-120.3 -> 116.7-117.0 (profiles/round6/experiments/conv_sp_geometry_modes.txt);
    // the 12-wavefront geometries gain nothing from it (47.2-47.9 -> 47.7-48.3, 41.1-41.3 -> 41.6-41.7).  Same arithmetic, same bits.
    if (geometry < 1000 && (geo == 81 || geo == 148) && coalign::lab_env("COALIGN_SP_FRONT", 1)) mode = 5;
#if defined(COALIGN_LAB) || defined(SP_TRACE)      // laboratory / trace builds carry every issue mode
    if (mode == 3 && (geo == 81 || geo == 148)) return launch_mode<3>(geo, a, out_kind, split_policy, ws, ws_bytes, s, query);
    if (mode == 6) return launch_mode<6>(geo, a, out_kind, split_policy, ws, ws_bytes, s, query);
    if (mode == 7) return launch_mode<7>(geo, a, out_kind, split_policy, ws, ws_bytes, s, query);
    if (mode == 4) return launch_mode<4>(geo, a, out_kind, split_policy, ws, ws_bytes, s, query);
    if (mode == 5) return launch_mode<5>(geo, a, out_kind, split_policy, ws, ws_bytes, s, query);
    return mode == 0 ? launch_mode<0>(geo, a, out_kind, split_policy, ws, ws_bytes, s, query) : mode == 2 ? launch_mode<2>(geo, a, out_kind, split_policy, ws, ws_bytes, s, query)
                                                                                              : launch_mode<1>(geo, a, out_kind, split_policy, ws, ws_bytes, s, query);
#else
    if (mode != kDefaultMode && geometry >= 1000) return COALIGN_ERR_UNSUPPORTED;      // (issue modes 0, 2-4, 6, 7 are measured-and-not-adopted variants: laboratory library only)
    if (mode == 5) return launch_mode<5>(geo, a, out_kind, split_policy, ws, ws_bytes, s, query);
    return launch_mode<kDefaultMode>(geo, a, out_kind, split_policy, ws, ws_bytes, s, query);
#endif
}

}  // namespace

#ifdef SP_TRACE
static long long *g_sp_trace = nullptr;
static int g_sp_ablate = 0;
extern "C" void coalign_conv3x3_sp_set_trace(long long *p) { g_sp_trace = p; }
extern "C" void coalign_conv3x3_sp_set_ablate(int v) { g_sp_ablate = v; }
#endif

extern "C" size_t coalign_sp_map_bytes(int N, int C, int H, int W) {
    if (N < 0 || C < 1 || H < 1 || W < 1 || C % 16) return 0;
    return (size_t)N * C * H * W * 4;
}

extern "C" int coalign_sp_pack(const float *x, int in_nhwc, void *y_sp, int N, int C, int H, int W, int32_t *range_flag, void *stream) {
    if (!x || !y_sp) return COALIGN_ERR_NULL_POINTER;
    if (N < 0 || C < 1 || H < 1 || W < 1) return COALIGN_ERR_BAD_SHAPE;
    if (C % 16 || (reinterpret_cast<uintptr_t>(y_sp) & 15)) return COALIGN_ERR_UNSUPPORTED;
    const size_t n = (size_t)N * (C / 16) * H * W;
    if (n == 0) return COALIGN_OK;
    hipLaunchKernelGGL(sp_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), x, static_cast<uint4 *>(y_sp), N, C, H * W, in_nhwc, range_flag);
    return coalign::check_launch();
}

extern "C" int coalign_sp_unpack(const void *x_sp, float *y, int out_nhwc, int N, int C, int H, int W, void *stream) {
    if (!x_sp || !y) return COALIGN_ERR_NULL_POINTER;
    if (N < 0 || C < 1 || H < 1 || W < 1) return COALIGN_ERR_BAD_SHAPE;
    if (C % 16 || (reinterpret_cast<uintptr_t>(x_sp) & 15)) return COALIGN_ERR_UNSUPPORTED;
    const size_t n = (size_t)N * (C / 16) * H * W;
    if (n == 0) return COALIGN_OK;
    hipLaunchKernelGGL(sp_unpack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), static_cast<const uint4 *>(x_sp), y, N, C, H * W, out_nhwc);
    return coalign::check_launch();
}

static int sp_check(int N, int Cin, int Cout, int H, int W) {
    if (N < 0 || H < 1 || W < 1 || Cin < 1 || Cout < 1) return COALIGN_ERR_BAD_SHAPE;
    if (Cin % 16 || Cout % kCoutTile || Cout > kMaxCoutLds) return COALIGN_ERR_UNSUPPORTED;
    if ((int64_t)N * (Cin > Cout ? Cin : Cout) * H * W > (int64_t)1 << 32) return COALIGN_ERR_UNSUPPORTED;      // group offsets are 32-bit
    return COALIGN_OK;
}

extern "C" size_t coalign_conv3x3_sp_workspace_bytes(int N, int Cin, int Cout, int H, int W, int geometry) {
    if (sp_check(N, Cin, Cout, H, W) != COALIGN_OK || N == 0) return 0;
    SpArgs a{};
    a.N = N; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W;
    SpLaunch l{};
    return dispatch_sp(a, SP_OUT_SP, geometry, nullptr, 0, nullptr, &l) == COALIGN_OK ? l.ws_bytes : 0;
}

static int conv3x3_sp_impl(const void *x_sp, const void *w_split, const float *bias, const void *residual, int residual_kind, void *y, void *y2, int out_kind,
                           int N, int Cin, int Cout, int H, int W, int relu, int geometry, int32_t *range_flag, void *workspace, size_t workspace_bytes, void *stream);

extern "C" int coalign_conv3x3_sp(const void *x_sp, const void *w_split, const float *bias, const void *residual, int residual_kind, void *y, int out_kind,
                                  int N, int Cin, int Cout, int H, int W, int relu, int geometry, int32_t *range_flag, void *workspace, size_t workspace_bytes,
                                  void *stream) {
    if (out_kind != SP_OUT_SP && out_kind != SP_OUT_NHWC) return y && x_sp && w_split && bias ? COALIGN_ERR_UNSUPPORTED : COALIGN_ERR_NULL_POINTER;
    return conv3x3_sp_impl(x_sp, w_split, bias, residual, residual_kind, y, nullptr, out_kind, N, Cin, Cout, H, W, relu, geometry, range_flag, workspace, workspace_bytes, stream);
}

// Round 6: a stage's LAST layer writes its map twice -- channels-last float32 (fusion kernel, exchange, 1 x 1 skip) and SP map (the next stage's strided
// convolution, coalign_conv3x3_sp_s2): the SP map holds coalign_sp_pack of the float32 map, bit for bit.
extern "C" int coalign_conv3x3_sp_both(const void *x_sp, const void *w_split, const float *bias, const void *residual, int residual_kind, float *y_nhwc, void *y_sp,
                                       int N, int Cin, int Cout, int H, int W, int relu, int geometry, int32_t *range_flag, void *workspace, size_t workspace_bytes,
                                       void *stream) {
    if (!y_sp) return COALIGN_ERR_NULL_POINTER;
    if (reinterpret_cast<uintptr_t>(y_sp) & 15) return COALIGN_ERR_UNSUPPORTED;
    return conv3x3_sp_impl(x_sp, w_split, bias, residual, residual_kind, y_nhwc, y_sp, SP_OUT_BOTH, N, Cin, Cout, H, W, relu, geometry, range_flag, workspace, workspace_bytes, stream);
}

static int conv3x3_sp_impl(const void *x_sp, const void *w_split, const float *bias, const void *residual, int residual_kind, void *y, void *y2, int out_kind,
                           int N, int Cin, int Cout, int H, int W, int relu, int geometry, int32_t *range_flag, void *workspace, size_t workspace_bytes, void *stream) {
    using namespace coalign;
    if (!x_sp || !w_split || !bias || !y || (residual_kind != SP_RES_NONE && !residual)) return COALIGN_ERR_NULL_POINTER;
    int rc = sp_check(N, Cin, Cout, H, W);
    if (rc != COALIGN_OK) return rc;
    if ((out_kind != SP_OUT_SP && out_kind != SP_OUT_NHWC && out_kind != SP_OUT_BOTH) || residual_kind < 0 || residual_kind > SP_RES_NHWC) return COALIGN_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(x_sp) | reinterpret_cast<uintptr_t>(w_split) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(residual)) & 15) return COALIGN_ERR_UNSUPPORTED;
    if (N == 0) return COALIGN_OK;
    const size_t wbytes = coalign_conv3x3_emu_weight_bytes_ex(Cin, Cout, 16, 1), tail = (size_t)Cout * 8;
    const char *wb = static_cast<const char *>(w_split);
    SpArgs a{};
    a.x = static_cast<const uint4 *>(x_sp);
    a.wt = static_cast<const uint4 *>(w_split);
    a.zero = reinterpret_cast<const uint4 *>(wb + wbytes - tail - 16);
    a.bias = bias;
    a.wscale = reinterpret_cast<const float *>(wb + wbytes - tail);
    a.residual = residual;
    a.y = y;
    a.y2 = y2;
    a.range_flag = range_flag;
    a.N = N; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W; a.relu = relu; a.res_kind = residual_kind;
    a.xcd = 1;
    a.prio_mode = coalign::lab_env("COALIGN_SP_PRIO", 1);      // (0: the fixed priorities of round 5; 2: + the later-dispatched wavefronts step down a tap later -- inside the noise)
    a.stream_out = coalign::lab_env("COALIGN_SP_STREAM", 0);      // (alone on the GPU the SP output gains 4 % from streaming stores; inside the two-stream frame pipeline it LOSES 1.7 %: laboratory switch)
#ifdef SP_TRACE
    a.trace = g_sp_trace;
    a.ablate = g_sp_ablate & 31;
    if (g_sp_ablate & 32) a.prio_mode = 1;      // (trace build: bit 5 of the ablation word switches the progress-based priority on)
#endif
    rc = dispatch_sp(a, out_kind, geometry, workspace, workspace_bytes, static_cast<hipStream_t>(stream), nullptr);
    return rc != COALIGN_OK ? rc : check_launch();
}
