// 3x3 / pad 1 convolution (stride 1 or 2) with fused bias (+ residual) (+ ReLU), fp32 in / fp32 (or SplitMap) out, computed on the 16-bit matrix cores by
// operand splitting in the CONSUMER ("fp32 emulation"), gfx950.  Since round 5 the detector's default route uses this kernel only where a chain of SplitMap
// layers (csrc/conv3x3_sp.hip) STARTS from float32 input -- the strided first convolution of a ResNet stage (channels-last or sparse-canvas input) and the shrink
// header's first convolution -- and for the arithmetics other than the default; rounds 2-4 ran every 3x3 layer of the model here.  NCHW in and out by default,
// optionally channels-last (NHWC) on the input or the output side, or a SplitMap on the output side (LAYOUT).
//
// Same layers and semantics as conv3x3.hip (opencood/models/sub_modules/resblock.py:53-69, base_bev_backbone_resnet.py:59-138,
// downsample_conv.py:7-50).  The fp32 MFMA runs at the VALU rate (157 TFLOP/s); the 16-bit matrix instructions are 16x faster and accumulate in fp32.
// Arithmetics (TERMS / VAR_F16; `terms` at the C ABI):
//   terms = 16 (the product default, COALIGN_CONV_EMU, backbone.py): sp16 pairs on v_mfma_f32_32x32x16_f16 -- every operand rounded to 22 significant bits,
//     x~ = x_h + 2^-10 x_l (common.h), weights scaled per output channel by a power of two; products w_h x_h in `acc`, w_h x_l + w_l x_h in `accl` (DUAL),
//     y = (acc + 2^-10 accl) 2^-k_c + (residual + bias); scale free;
//   terms = 3 (default of rounds 2-3): x = x_h + x_m + x_l, three bf16 terms, error-free; six products per fp32 product, dropped terms <= 2^-24 |w x|;
//   terms = 2: two bf16 terms, three products, dropped terms <= 2^-16 |w x| (opt-in).
// The weights are split on the host once; the input pixels are split in registers, once per pixel and chunk.
//
// GEMM view per image:  D[cout, pixel] = sum_{cin, tap} W[cout, cin, tap] * X[cin, pixel + tap].  Two weight images:
//   * tap-major (the detector's stride-1 layers, COALIGN_LAYOUT_W_TAPMAJOR, template bit VAR_TAPK): one MFMA has K = 16 = the 16 input
//     channels of ONE tap (lanes 0-31 channels 0-7, lanes 32-63 channels 8-15), nine per 16-channel interval, no zero tap; one
//     workgroup per CU (111 KB of double-buffered weights + one 16-channel split patch);
//   * tap pairs (the strided layers, the original kernel described below): K = 16 = two taps x the 8 input channels of a chunk:
//     lanes 0-31 (k 0..7) carry tap 2s, lanes 32-63 (k 8..15) tap 2s + 1, s = 0..4 (the tenth tap is zero weights).
// (A producer / consumer variant was measured and removed in round 4: per layer level with the tap-major kernel, whole frame 300 vs 311 frames/s.)  Measured history: DESIGN.md section 8.  A workgroup (8 or 12 wavefronts) owns 8 / 12 row segments of 32 pixels and 64 output channels; persistent
// workgroups, one barrier per 8-channel chunk, everything one chunk ahead: the split weights of chunk L + 1 arrive by LDS-DMA, the
// fp32 halo pixels of chunk L + 1 are loaded into registers (one lane = one pixel, coalesced along the patch rows), split ONCE per
// pixel after the MFMA steps of chunk L and written to LDS as [term][pixel][8 cin] bf16 -- a B operand is then one ds_read_b128
// and the MFMA stream carries no VALU work.  Long tiles are load-balanced stream-K style (see the kernel).
#include "common.h"
#include <cstdlib>

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 halfx8 __attribute__((ext_vector_type(8)));

// F16 (round 4, terms = 16 at the C ABI): the 2-way split with FP16 terms -- x = x_h + x_l, x_h = fp16(x) (11 significant bits), x_l = fp16(x - x_h):
// 22 bits of every operand instead of bf16's 16, at the 2-way split's three products.  The 16-byte operands travel through the kernel typed bf16x8;
// only the split and the matrix instruction differ.
template <bool F16>
__device__ __forceinline__ floatx16 mfma16(const bf16x8 &a, const bf16x8 &b, const floatx16 &c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(halfx8, a), __builtin_bit_cast(halfx8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

constexpr int kKC = 8;               // input channels per LDS chunk = k values per lane
constexpr int kCoutTile = 64;        // output channels per workgroup
constexpr int kSteps = 5;            // MFMA steps per chunk: taps (0,1) (2,3) (4,5) (6,7) (8,-)

typedef __attribute__((address_space(1))) const void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

struct EmuArgs {
    const float *__restrict__ x;
    const uint4 *__restrict__ wt;     // [Cout / 64][Cin / 8][5 steps][TERMS][2 k-groups][64 cout][8 bf16]
    const float *__restrict__ bias, *__restrict__ residual;
    float *__restrict__ y;
    int N, Cin, Cout, H, W, relu, tiles_x, tiles_per_img, total_tiles;      // H, W: OUTPUT size
    int Hin, Win;                     // input size (== H, W for stride 1; H = ceil(Hin / 2) for stride 2)
    float *__restrict__ partial;      // [grid][16 * NCO][threads]: accumulators of a tile whose chunks are split over two workgroups
    int *flags;                       // [grid], zeroed per launch: flags[g] = 1 once workgroup g has published its partial tile
    int prio_mode;                    // 1: the first-dispatched half of the grid outranks the second half (see the kernel)
    // round 4, sparse canvas (csrc/pillar_sparse.hip): x = pillar feature rows [M][Cin]; a pixel (n, y, x) of the logical channels-last input is row
    // (stamp & 0xffffffff) of x if stamps[(n * Hin + y) * Win + x] >> 32 equals *tag_ptr, else zero.  Channels-last input layouts only.
    const unsigned long long *__restrict__ stamps;
    const int *__restrict__ tag_ptr;
    int *range_flag;                  // round 6 (may be NULL): bit 0 is set when an SP output value exceeds the pair's range (|y| > 65504), as in csrc/conv3x3_sp.hip
    unsigned sparse_rows;             // round 6: rows behind x -- a stamp naming a row beyond them (a canvas object kept across a later encode through the same stamp map) reads as empty
    // round 5, fp16 split only: the per-output-channel power-of-two scale the weight image was multiplied with before it was split (its tail:
    // [Cout] 2^-k_c, then [Cout] 2^k_c).  bias + residual enter the accumulator times 2^k_c, the tile leaves it times 2^-k_c: both exact.
    const float *__restrict__ wscale;
#ifdef EMU_TRACE
    long long *trace;                 // profiling aid (tools/trace_conv_emu.py): [2 workgroups][waves][64 chunks][8 stamps]
    int ablate;                       // 1: no weight DMA, 2: no halo-pixel loads, 4: no matrix steps (wrong results; what each part costs)
#endif
};

#ifdef EMU_TRACE
#define EMU_STAMP(k)                                                                                              \
    if ((g == 0 || g == 100) && lane == 0 && L < 64)                                                              \
        a.trace[((((g ? 1 : 0) * G::WAVES + wave) * 64) + L) * 8 + (k)] = (long long)__builtin_amdgcn_s_memtime()
#else
#define EMU_STAMP(k)
#endif

enum { LAYOUT_NCHW = 0, LAYOUT_OUT_NHWC = 1, LAYOUT_IN_NHWC = 2, LAYOUT_NHWC = 3, LAYOUT_OUT_SP = 4 };      // bits: 1 = channels-last output, 2 = channels-last input, 4 (round 5, fp16 split) = the output is an SP map (csrc/conv3x3_sp.hip)

// TAPK ("K = 144"): a barrier interval is 16 input channels x 9 taps = nine MFMA steps whose K = 16 is the 16 channels of ONE tap (lanes
// 0-31 channels 0-7, lanes 32-63 channels 8-15) -- no zero tenth tap, 10 % fewer MFMAs.  Weight image [9 taps][term][2 channel halves]
// [64 cout][8 cin] (55 KB per interval, double buffered), split patch as for KCH = 2.
// XV bit 0 (STACK): the N images of the batch are tiled as ONE tall image of N * H rows (a tile may straddle two images: two zero rows
// in the patch stand in for the padding below the upper image and above the lower one); bit 1 (NCO1): a wavefront owns 32 output
// channels, the workgroup has 2 * NPB wavefronts.  Both exist for load balance, see dispatch_tapk().
template <int BH, int BW, int NPB, int TERMS, int KCH, int STRIDE = 1, int PBUF = 2, bool TAPK = false, int XV = 0>
struct Geo {
    static_assert(!TAPK || KCH == 2, "tap-major steps pair the two 8-channel halves of a 16-channel interval");
    static constexpr bool STACK = (XV & 1) != 0, NCO1 = (XV & 2) != 0;
    // XV bit 2 (NBX4): the workgroup's NPB pixel blocks form a grid of four block columns (a tile of BH * NPB / 4 rows x 4 * BW columns) instead of
    // one column of blocks: with 4 x 8-pixel blocks, an 88-column map ends in a tile whose fourth block column lies past the map -- those
    // wavefronts run no matrix steps (the 1 x 32 blocks compute 96 columns for 88).  PW is padded to 8 mod 16 so that the four rows a B-operand
    // read touches fall into alternating halves of the LDS banks.
    static constexpr int NBX = (XV & 4) ? 4 : 1;
    static_assert(NPB % NBX == 0, "whole block rows");
    static_assert(!STACK || STRIDE == 1, "stacked tiles: stride-1 layers only");
    static constexpr int NCO = (NPB >= 4 && !NCO1) ? 2 : 1;                   // accumulator tiles (32 output channels each) per wave
    static constexpr int WAVES = NCO == 2 ? NPB : 2 * NPB;
    static constexpr int THREADS = 64 * WAVES;
    static constexpr int TH = BH * NPB / NBX, TW = BW * NBX;                  // output tile
    static constexpr int PWU = STRIDE * TW + 3 - STRIDE;                      // patch columns in use
    static constexpr int PH = STRIDE * TH + 3 - STRIDE + (STACK ? 2 : 0), PW = NBX == 1 ? PWU : (PWU + 7) / 16 * 16 + 8;   // input halo patch: rows STRIDE * y0 - 1 .., columns STRIDE * x0 - 1 ..
    static constexpr int PIX = PH * PW;                                       // pixel slots of the patch
    static constexpr int SLOTS = (PIX + THREADS - 1) / THREADS;               // pixel slots one thread splits per chunk
    static constexpr int WQ = (TAPK ? 9 : kSteps) * TERMS * 2 * kCoutTile;    // 16-byte groups of one weight unit (8-channel chunk; TAPK: the whole interval)
    static constexpr int WUNITS = TAPK ? 1 : KCH;                             // weight units per barrier interval
    static constexpr int STEPS = TAPK ? 9 : kSteps * KCH;                     // MFMA steps per barrier interval
    static constexpr int WINSTR = WQ / 64;
    // KCH 8-channel chunks are processed per barrier.  LDS map (floats): two weight images of KCH chunks | two split patches
    // [chunk][term][y][x][8 cin] bf16
    static constexpr int W_OFF = 0, WSZ = WUNITS * WQ * 4, B_OFF = 2 * WSZ, BSZ1 = TERMS * PIX * 4, BSZ = KCH * BSZ1;
    // PBUF = 1: ONE split-patch buffer (one more barrier per chunk: the next chunk's pixels are written after every wave has read the
    // current ones).  With the 3-way split that takes a workgroup from 94 to 78 KB, i.e. TWO workgroups per CU: the barrier / operand
    // phases of one overlap the MFMA phase of the other.
    static constexpr size_t LDS_BYTES = ((size_t)B_OFF + PBUF * (size_t)BSZ) * 4;
};

struct Tile {
    int n, cg, y0, x0;
};

// error-free split of the 8 input channels of one pixel: out[t] = term t of each channel, 8 bf16 = one MFMA operand
template <int TERMS, bool F16 = false>
__device__ __forceinline__ void split_pixel(const float (&v)[8], bf16x8 (&out)[TERMS]) {
    if constexpr (F16) {
        // The "split pair" of common.h (sp16): the operand rounded to 22 significant bits, x~ = x_h + 2^-10 x_l', x_h = the leading 11 bits as an fp16 number
        // (exact truncation), x_l' = (x~ - x_h) * 2^10 -- the next 11 bits, SCALED into fp16's normal range so that they survive for every |x| >= 2^-14
        // (unscaled, x - x_h is an fp16 subnormal for |x| < 2^-3 and the split degrades to an absolute 2^-25: round 4's hole).  Products with an x_l' or
        // w_l' factor are summed in their own accumulator and enter with 2^-10 at the end of the tile (see the kernel).
        static_assert(TERMS == 2, "fp16 split: two terms");
        unsigned hi[4], lo[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) coalign::sp16_split2(v[2 * i], v[2 * i + 1], hi[i], lo[i]);
        out[0] = __builtin_bit_cast(bf16x8, uint4{hi[0], hi[1], hi[2], hi[3]});
        out[1] = __builtin_bit_cast(bf16x8, uint4{lo[0], lo[1], lo[2], lo[3]});
        return;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const __bf16 h = (__bf16)v[i];
        const float r = v[i] - (float)h;
        out[0][i] = h;
        const __bf16 m = (__bf16)r;
        out[1][i] = m;
        if (TERMS == 3) out[TERMS - 1][i] = (__bf16)(r - (float)m);
    }
}

// The split variant's occupancy is pinned (registers capped): its hand-over code, executed once per split tile, would otherwise
// cost a resident workgroup -- the spills it causes sit outside the chunk loop.  8 wavefronts x 2 workgroups = 4 per SIMD, 12
// wavefronts = 3 per SIMD.  (Pinning the plain variant as well changes hipcc's scheduling and was measured 5-20 % slower.)
// VAR: bit 0 = tap-major weight image (TAPK), bit 1 = the weight LDS-DMA is issued from inline assembly: hipcc cannot prove the DMA
// destination disjoint from the operand reads and otherwise waits for vmcnt(0) -- the full latency of everything just issued -- before
// the first matrix instruction of every interval (the top-of-loop s_waitcnt(0) is the real synchronisation point of the transfer).
// (Pinning the step loop's order "next step's operand reads, then this step's matrix instructions" with sched_barrier: 147 registers,
// 3-7 % slower per layer on the tap-major geometries; computing the bf16 split ahead of the second barrier: no change; issuing the next
// interval's weight DMA and halo loads in shares between the matrix steps instead of all at once after the barrier: no change -- the
// interval timelines (tools/trace_conv_emu.py) show the burst already overlapped by the other wavefronts' matrix instructions.  All removed.)
enum { VAR_TAPK = 1, VAR_ASM_DMA = 2, VAR_STACK = 4, VAR_NCO1 = 8, VAR_NBX4 = 16, VAR_F16 = 32, VAR_QUAD = 64 };
template <int BH, int BW, int NPB, int TERMS, int KCH, bool SPLIT, int STRIDE = 1, int LAYOUT = LAYOUT_NCHW, int PBUF = 2, int VAR = 0>
__global__ __launch_bounds__(64 * ((NPB >= 4 && !(VAR & VAR_NCO1)) ? NPB : 2 * NPB))
__attribute__((amdgpu_waves_per_eu(SPLIT ? ((VAR & 1) ? (NPB + 3) / 4 : NPB == 12 ? 3 : 4) : (VAR & 64) ? 4 : 1, SPLIT ? ((VAR & 1) ? (NPB + 3) / 4 : NPB == 12 ? 3 : 4) : 8)))
void conv3x3_emu_kernel(const EmuArgs a) {
    constexpr bool TAPK = (VAR & VAR_TAPK) != 0, STACK = (VAR & VAR_STACK) != 0, F16 = (VAR & VAR_F16) != 0;
    constexpr bool DUAL = F16;          // sp16 operands (common.h): a second accumulator for the products that carry 2^10, per-channel weight scale
    using G = Geo<BH, BW, NPB, TERMS, KCH, STRIDE, PBUF, TAPK, (VAR >> 2) & 7>;
    static_assert(!(SPLIT && (STRIDE != 1 || (LAYOUT != LAYOUT_NCHW && LAYOUT != LAYOUT_OUT_SP) || STACK)), "stream-K hand-over only for the plain stride-1 NCHW-input variants");
    extern __shared__ __attribute__((aligned(1024))) float lds[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, half = lane >> 5, p = lane & 31;      // wave: scalar
    const size_t plane = (size_t)a.H * a.W, plane_in = (size_t)a.Hin * a.Win;
    const unsigned sparse_tag = a.stamps ? (unsigned)*a.tag_ptr : 0u;
    const int groups = a.Cout / kCoutTile, chunks = a.Cin / (kKC * KCH);       // `chunks`: barrier intervals per tile, KCH x 8 channels each
    int member_cg = 0;
    auto decode = [&](int t) {
        Tile c;
        c.cg = SPLIT ? member_cg : t % groups;             // stream-K: t is the spatial tile, the output-channel group is the workgroup's member index in its gang (below)
        const int sp = SPLIT ? t : t / groups;
        c.n = sp / a.tiles_per_img;
        const int r = sp - c.n * a.tiles_per_img, ty = r / a.tiles_x;
        c.y0 = ty * G::TH;
        c.x0 = (r - ty * a.tiles_x) * G::TW;
        return c;
    };
    const int pb = wave % NPB, cb = (G::NCO == 2 ? 0 : wave / NPB) * 32;
    const int blk_y = pb / G::NBX, blk_x = pb - blk_y * G::NBX;             // this wavefront's pixel block inside the tile
    const int py = blk_y * BH + p / BW, px = blk_x * BW + p % BW;
    constexpr int NB = TAPK ? 9 : kSteps;
    int boff[NB];                                         // pixel slot of this lane's tap in step s (tap 9 -> tap 8, zeroed below)
#pragma unroll
    for (int s = 0; s < NB; ++s) {
        // TAPK: step s = tap s, this lane's channel half selects the 8-channel sub-patch
        const int t = TAPK ? s : (2 * s + half < 9 ? 2 * s + half : 8);
        boff[s] = (TAPK ? half * TERMS * G::PIX : 0) + (STRIDE * py + t / 3) * G::PW + STRIDE * px + t % 3;
    }
    const int wlane = half * kCoutTile + cb + p;          // 16-byte group of this lane inside one (step, term) weight block

    // Patch plan of a tile: thread t owns pixel slots t, t + THREADS, ... of the halo patch; per slot the offset of that pixel
    // inside an input plane, or -1 for the zero padding outside the image.  Computed once per tile.
    // QUAD (round 5, the strided fp16 layers on channels-last / sparse input): a lane stages FOUR channels of a pixel (one float4) instead of eight or sixteen,
    // slot = (pixel, channel quad), the quads of a pixel in neighbouring lanes.  The interval timelines of the strided layers (tools/trace_conv_emu.py STRIDE=2,
    // profiles/round5/experiments/conv_strided_timeline.txt) showed 2400-3500 clocks of "issue" per interval against 2200 of matrix steps: with one pixel per lane
    // every 16-byte load of a wavefront touches 64 different 256-byte pixel records -- the texture addresser takes them one cache line per cycle.  Two / four
    // lanes per pixel are 32 / 16 lines per instruction for the same bytes.
    // Together with ONE patch buffer (75 KB of LDS) and the registers capped at 128 (80 bytes of scratch outside the interval loop) TWO workgroups share a CU: the
    // staging phases of one run under the matrix steps of the other.  Dense channels-last input 64.0 -> 54.3 us (5 x 64 -> 128 @ 100 x 352, same box); the
    // SPARSE canvas loses (73 -> 81 us: most of its slots are empty cells, five slots per thread instead of three only add instructions) and keeps the
    // pixel-per-lane staging with two patch buffers.
    constexpr bool QUAD = (VAR & VAR_QUAD) != 0;
    static_assert(!QUAD || (F16 && STRIDE == 2 && (LAYOUT & LAYOUT_IN_NHWC) != 0 && !TAPK), "quad staging: the strided fp16 layers on channels-last input");
    constexpr int QN = 2 * KCH;                                                // channel quads per pixel and interval
    constexpr int PSLOTS = QUAD ? (G::PIX * QN + G::THREADS - 1) / G::THREADS : G::SLOTS, PVW = QUAD ? 4 : 8 * KCH;
    struct Plan {
        const float *base;     // first input plane of the tile's image
        const uint4 *wsrc;     // this lane inside the tile's first weight chunk
        int off[PSLOTS];
    };
    // STACK: t.y0 is a row of the stacked image (N * H rows).  n0 = its image, yl0 = its row inside that image, yb = H - yl0 = how many
    // of the tile's rows still belong to image n0 (yb >= TH: no image boundary inside the tile).  Patch rows with a boundary inside:
    //   0 .. yb        rows yl0 - 1 .. H - 1 of image n0        | yb + 1, yb + 2   ZERO (below image n0 / above image n0 + 1)
    //   yb + 3 ..      rows 0 .. of image n0 + 1                  so tile row j reads patch rows j + shift .. j + shift + 2, shift = 2 for j >= yb
    auto make_plan = [&](const Tile &t) {
        Plan pl;
        const int n0 = STACK ? t.y0 / a.H : t.n;
        pl.base = a.stamps ? a.x : a.x + (size_t)n0 * a.Cin * plane_in;
        pl.wsrc = a.wt + (size_t)t.cg * chunks * (G::WUNITS * G::WQ) + lane;
        const int yl0 = t.y0 - n0 * a.H, yb = a.H - yl0;
#pragma unroll
        for (int j = 0; j < PSLOTS; ++j) {
            const int i = QUAD ? (tid + j * G::THREADS) / QN : tid + j * G::THREADS;
            const int y = i / G::PW, xq = i - y * G::PW;
            const int gx = STRIDE * t.x0 - 1 + xq;
            if constexpr (STACK) {
                int gy, img = 0;
                bool ok = i < G::PIX && xq < G::PWU && gx >= 0 && gx < a.Win;
                if (yb >= G::TH || y <= yb) gy = yl0 - 1 + y;
                else if (y <= yb + 2) { gy = 0; ok = false; }
                else { gy = y - (yb + 3); img = 1; ok = ok && n0 + 1 < a.N; }
                ok = ok && gy >= 0 && gy < a.Hin;
                pl.off[j] = ok ? img * a.Cin * (int)plane_in + gy * a.Win + gx : -1;
            } else {
                const int gy = STRIDE * t.y0 - 1 + y;
                pl.off[j] = (i < G::PIX && xq < G::PWU && gy >= 0 && gy < a.Hin && gx >= 0 && gx < a.Win) ? gy * a.Win + gx : -1;
                if constexpr ((LAYOUT & LAYOUT_IN_NHWC) != 0) {
                    if (a.stamps) {                                 // sparse canvas: the pixel's feature row, or nothing
                        const unsigned long long st = a.stamps[(size_t)n0 * plane_in + (pl.off[j] < 0 ? 0 : pl.off[j])];
                        pl.off[j] = (pl.off[j] >= 0 && (unsigned)(st >> 32) == sparse_tag && (unsigned)st < a.sparse_rows) ? (int)(unsigned)st : -1;
                    }
                }
            }
        }
        return pl;
    };
    // chunk c of the tile: the 8 input channels of this thread's pixel slots -> registers (plain coalesced loads: consecutive
    // lanes = consecutive pixels of a patch row; clamped address + zero select, no divergent branch around the loads)
    auto load_patch = [&](const Plan &pl, int c, float (&v)[PSLOTS][PVW]) {
        if constexpr (QUAD) {
            const float *src = pl.base + (size_t)c * (kKC * KCH) + 4 * (tid & (QN - 1));      // (THREADS % QN == 0: a thread's quad is the same in every slot)
#pragma unroll
            for (int j = 0; j < PSLOTS; ++j) {
                const int o = pl.off[j] < 0 ? 0 : pl.off[j];
                const float4 t = *reinterpret_cast<const float4 *>(src + (size_t)o * a.Cin);
                v[j][0] = t.x; v[j][1] = t.y; v[j][2] = t.z; v[j][3] = t.w;
            }
        } else if constexpr ((LAYOUT & LAYOUT_IN_NHWC) != 0) {          // channels-last input: the 8 channels of a pixel are 32 contiguous bytes
            const float *src = pl.base + (size_t)c * (kKC * KCH);
#pragma unroll
            for (int j = 0; j < G::SLOTS; ++j) {
                const int o = pl.off[j] < 0 ? 0 : pl.off[j];
                const float4 *q = reinterpret_cast<const float4 *>(src + (size_t)o * a.Cin);
#pragma unroll
                for (int k4 = 0; k4 < 2 * KCH; ++k4) {
                    const float4 t = q[k4];
                    v[j][4 * k4] = t.x; v[j][4 * k4 + 1] = t.y; v[j][4 * k4 + 2] = t.z; v[j][4 * k4 + 3] = t.w;
                }
            }
        } else {
            const float *src = pl.base + (size_t)c * (kKC * KCH) * plane_in;
#pragma unroll
            for (int j = 0; j < G::SLOTS; ++j) {
                const int o = pl.off[j] < 0 ? 0 : pl.off[j];
#pragma unroll
                for (int k = 0; k < 8 * KCH; ++k) v[j][k] = src[(size_t)k * plane_in + o];
            }
        }
    };
    // ... split into bf16 terms and written as [term][pixel slot][8 cin] into split-patch buffer `slot`
    auto split_patch = [&](const Plan &pl, float (&v)[PSLOTS][PVW], uint4 (&sp)[G::SLOTS][KCH][TERMS]) {
#pragma unroll
        for (int j = 0; j < G::SLOTS; ++j) {
#pragma unroll
            for (int h = 0; h < KCH; ++h) {
                float u[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) u[k] = pl.off[j] < 0 ? 0.f : v[j][QUAD ? 0 : 8 * h + k];      // (QUAD never comes here)
                bf16x8 o[TERMS];
                split_pixel<TERMS, F16>(u, o);
#pragma unroll
                for (int t = 0; t < TERMS; ++t) sp[j][h][t] = __builtin_bit_cast(uint4, o[t]);
            }
        }
    };
    auto write_patch = [&](int slot, const uint4 (&sp)[G::SLOTS][KCH][TERMS]) {
        uint4 *bt = reinterpret_cast<uint4 *>(lds + G::B_OFF + (PBUF == 2 ? slot : 0) * G::BSZ);
#pragma unroll
        for (int j = 0; j < G::SLOTS; ++j) {
            const int i = tid + j * G::THREADS;
            if (i < G::PIX) {
#pragma unroll
                for (int h = 0; h < KCH; ++h)
#pragma unroll
                    for (int t = 0; t < TERMS; ++t) bt[(h * TERMS + t) * G::PIX + i] = sp[j][h][t];
            }
        }
    };
    auto store_patch = [&](const Plan &pl, int slot, float (&v)[PSLOTS][PVW]) {
        if constexpr (QUAD) {      // four channels -> 8 bytes of h and 8 bytes of l, each into its half of the 16-byte operand group [8-channel half][term][pixel]
            uint2 *bt = reinterpret_cast<uint2 *>(lds + G::B_OFF + (PBUF == 2 ? slot : 0) * G::BSZ);
            const int quad = tid & (QN - 1);
#pragma unroll
            for (int j = 0; j < PSLOTS; ++j) {
                const int i = (tid + j * G::THREADS) / QN;
                unsigned h01, l01, h23, l23;
                coalign::sp16_split2(pl.off[j] < 0 ? 0.f : v[j][0], pl.off[j] < 0 ? 0.f : v[j][1], h01, l01);
                coalign::sp16_split2(pl.off[j] < 0 ? 0.f : v[j][2], pl.off[j] < 0 ? 0.f : v[j][3], h23, l23);
                if (i < G::PIX) {
                    const int grp = ((quad >> 1) * TERMS) * G::PIX + i;
                    bt[(size_t)grp * 2 + (quad & 1)] = uint2{h01, h23};
                    bt[(size_t)(grp + G::PIX) * 2 + (quad & 1)] = uint2{l01, l23};
                }
            }
        } else {
            uint4 sp[G::SLOTS][KCH][TERMS];
            split_patch(pl, v, sp);
            write_patch(slot, sp);
        }
    };
    // LDS-DMA of weight chunk c of the tile into weight buffer `slot` (scalar LDS addresses, every lane active)
    constexpr int WJ = (G::WUNITS * G::WINSTR + G::WAVES - 1) / G::WAVES;
    auto issue_weights = [&](const Plan &pl, int c, int slot) {
        float *wdst = lds + G::W_OFF + slot * G::WSZ;
        const uint4 *wsrc = pl.wsrc + (size_t)c * (G::WUNITS * G::WQ);
#pragma unroll
        for (int j = 0; j < WJ; ++j) {
            const int ins = wave + G::WAVES * j;
            if (ins < G::WUNITS * G::WINSTR) {
                if constexpr (VAR & VAR_ASM_DMA) {
                    const unsigned dst = (unsigned)(size_t)(lptr_t)(wdst + ins * 256);      // wave-uniform LDS byte address -> M0
                    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(__builtin_amdgcn_readfirstlane(dst)), "v"(wsrc + ins * 64) : "memory", "m0");
                } else {
                    __builtin_amdgcn_global_load_lds((gptr_t)(wsrc + ins * 64), (lptr_t)(wdst + ins * 256), 16, 0, 0);
                }
            }
        }
    };

    // Work = total_tiles x chunks (tile, chunk) steps.  SPLIT (stream-K): cut into gridDim.x equal contiguous ranges, so every
    // persistent workgroup runs the same number of MFMAs (+-1 chunk) whatever the tile count -- with whole tiles per workgroup,
    // 715 tiles on 512 resident workgroups are two rounds at 70 % fill.  A range may start in the middle of a tile (then this
    // workgroup computes the tile's last chunks from zero accumulators and publishes them) and may end in the middle of a tile
    // (then it owns that tile: bias / residual start, its own first chunks, plus the partial the next workgroup published at the
    // very beginning of its range).  Ranges are at least one tile long, so a tile has at most two contributors and the split --
    // hence the summation order -- is a pure function of the shape: results stay deterministic.  !SPLIT: whole tiles g, g + n, ...
    // Logical workgroup id.  The hardware hands workgroup b to XCD b % 8, each with its own L2; tiles are numbered with the output-channel
    // group fastest, then along x: neighbours in that order read the same input patch (other weights) or share halo columns and cache lines.
    // XCD k therefore takes the k-th EIGHTH of the logical ids instead of every eighth one, so that those re-reads hit its L2
    // (prio_mode bit 3, COALIGN_EMU_XCD=0 switches it off; the stream-K hand-over is between logical neighbours g, g + 1 as before).  Measured per layer,
    // on / off: 100 x 352 77.0 / 81.4 us, 50 x 176 72.7 / 76.2 us, 25 x 88 86.0 / 87.1 us at 1.7 % less board power; frame rate 345-346 / 340-344.
    const int n_wg = gridDim.x;
    int g = blockIdx.x;
    if (a.prio_mode & 8) {
        const int q = n_wg >> 3, r = n_wg & 7, k = g & 7, j = g >> 3;
        g = k * q + (k < r ? k : r) + j;
    }
    // Round 5: stream-K in GANGS -- the `groups` workgroups gang * groups + member working on the output-channel groups of the same spatial tiles take the SAME
    // range of (spatial tile, chunk) steps: they read the same input patch at the same time, so the XCD's L2 serves all but one of those reads (cut per
    // workgroup over the flat tile list, neighbours sat in different chunks of different tiles: 1142 MB of HBM traffic for the shrink header's 54 MB input,
    // L2 hit 10 %, profiles/round5/pmc_summary.json).  Hand-overs go to the same member of the previous gang; the grid is a whole number of gangs.
    const int gangs = SPLIT ? n_wg / groups : 1, gang = SPLIT ? g / groups : 0, hand = SPLIT ? groups : 1;
    member_cg = SPLIT ? g - gang * groups : 0;
    const long long S = (long long)(SPLIT ? a.total_tiles / groups : a.total_tiles) * chunks;
    const int s0 = SPLIT ? (int)(S * gang / gangs) : 0;
    const int n_local = SPLIT ? (int)(S * (gang + 1) / gangs) - s0 : ((a.total_tiles - g + n_wg - 1) / n_wg) * chunks;
    auto global_step = [&](int l) { return SPLIT ? s0 + l : (g + (l / chunks) * n_wg) * chunks + l % chunks; };
    if (n_local <= 0) return;
#ifdef EMU_TRACE
    if (tid == 0) a.trace[2 * 16 * 64 * 8 + 2 * g] = wall_clock64();   // 100 MHz wall clock: start / end of every workgroup
#endif

    // Chunk-level software pipeline, one barrier per chunk, everything one chunk ahead.  In iteration L a wavefront
    //   * issues the LDS-DMA of the split weights of chunk L + 1 (weight buffer (L + 1) % 2) and loads the fp32 halo pixels of
    //     chunk L + 1 into registers,
    //   * runs the MFMA steps of chunk L out of split-patch buffer L % 2 and weight buffer L % 2,
    //   * then splits the loaded pixels ONCE (its 8 channels -> TERMS operands of 8 bf16) into split-patch buffer (L + 1) % 2.
    // The barrier at the top of iteration L + 1 (vmcnt / lgkmcnt 0) closes all three; what iteration L writes was last read in
    // iteration L - 1.  The MFMA stream carries no VALU work: a B operand is one ds_read_b128.
    int tile = global_step(0) / chunks;
    Tile cur = decode(tile);
    Plan plan = make_plan(cur);
    float pv[PSLOTS][PVW];
    load_patch(plan, global_step(0) - tile * chunks, pv);
    issue_weights(plan, global_step(0) - tile * chunks, 0);
    store_patch(plan, 0, pv);
    // Issue priorities.  Inside a workgroup the later-dispatched half of the wavefronts loses every arbitration otherwise.  Between the two
    // workgroups that share a CU (prio_mode 1): the interval timelines show them running IN PHASE -- both in their matrix steps, then both in
    // the barrier / load / split part with the matrix pipe idle (kernel time = matrix time + the rest, no overlap).  A strict rank makes the
    // higher-ranked workgroup take the matrix pipe alone and finish its steps in half the time, so the other one's steps fall into its
    // barrier / load / split part: anti-phase by arbitration.  Workgroups g and g + gridDim / 2 are the usual co-residents (dispatch order).
    {
        const int rank = (wave >= G::WAVES / 2 ? 1 : 0) + ((a.prio_mode & 7) == 1 && 2 * (int)blockIdx.x < (int)gridDim.x ? 2 : 0);
        if (rank == 1) __builtin_amdgcn_s_setprio(1);
        else if (rank == 2) __builtin_amdgcn_s_setprio(2);
        else if (rank == 3) __builtin_amdgcn_s_setprio(3);
    }
    int L = 0;
    while (L < n_local) {
        const int gs0 = global_step(L);
        const int c_begin = gs0 - tile * chunks;
        const int c_end = (n_local - L) < (chunks - c_begin) ? c_begin + (n_local - L) : chunks;
        const bool head = c_begin == 0, complete = c_end == chunks;
        // STACK: this lane's output pixel lives in image out_n at row gy; tile rows at / past the image boundary read the patch two rows lower
        int out_n = cur.n, gy = cur.y0 + py, bshift = 0;
        const int gx = cur.x0 + px;
        bool live;
        if constexpr (STACK) {
            const int n0 = cur.y0 / a.H, yl0 = cur.y0 - n0 * a.H, yb = a.H - yl0;
            const bool lower = py >= yb;                   // (yb >= TH: never)
            out_n = n0 + (lower ? 1 : 0);
            gy = lower ? py - yb : yl0 + py;
            bshift = lower ? 2 * G::PW : 0;
            live = out_n < a.N && gx < a.W;
        } else {
            live = gy < a.H && gx < a.W;
        }
        // (a lane without an output pixel reads the residual of pixel 0 of image 0: a block of several rows may end past the last image)
        const size_t obase = ((size_t)(live ? out_n : 0) * a.Cout + cur.cg * kCoutTile + cb + 4 * half) * plane + (live ? (size_t)gy * a.W + gx : 0);
        const float *bias = a.bias + cur.cg * kCoutTile + cb + 4 * half;
        const float *winv = DUAL ? a.wscale + cur.cg * kCoutTile + cb + 4 * half : nullptr;
        // a wavefront whose output rows lie below the map (the last row tile: 108 rows for 100, 56 for 50, 32 for 25) still stages pixels,
        // issues weight transfers and meets the barriers, but runs no matrix steps: its share of the padded work costs no energy
        const bool wave_live = __builtin_amdgcn_readfirstlane((int)(cur.y0 + blk_y * BH < (STACK ? a.N * a.H : a.H) && cur.x0 + blk_x * BW < a.W)) != 0;
        floatx16 acc[G::NCO], accl[DUAL ? G::NCO : 1];
#pragma unroll
        for (int q = 0; q < (DUAL ? G::NCO : 1); ++q) accl[q] = floatx16{0};
        if ((SPLIT && !head) || !wave_live || DUAL) {      // (fp16 split: bias + residual are added in the epilogue, y = tile * 2^-k_c + (residual + bias), as conv3x3_sp.hip)
#pragma unroll
            for (int q = 0; q < G::NCO; ++q) acc[q] = floatx16{0};
        } else if (a.residual) {
#pragma unroll
            for (int q = 0; q < 16 * G::NCO; ++q) {
                const int c = (q / 16) * 32 + 8 * ((q % 16) / 4) + (q % 4);
                acc[q / 16][q % 16] = a.residual[obase + (size_t)c * plane] + bias[c];
            }
        } else {
#pragma unroll
            for (int q = 0; q < 16 * G::NCO; ++q) {
                const int c = (q / 16) * 32 + 8 * ((q % 16) / 4) + (q % 4);
                acc[q / 16][q % 16] = bias[c];
            }
        }
        Tile next = cur;
        Plan nplan = plan;
        int ntile = tile;
        for (int chunk = c_begin; chunk < c_end; ++chunk, ++L) {
            EMU_STAMP(0);
            __builtin_amdgcn_s_waitcnt(0);
            EMU_STAMP(1);
            __syncthreads();
            EMU_STAMP(2);
            const bool more = L + 1 < n_local;
            if (more) {
                const int ns = global_step(L + 1);
                const int nt = ns / chunks;
                if (nt != tile) {                                      // the next chunk opens this workgroup's next tile
                    ntile = nt;
                    next = decode(nt);
                    nplan = make_plan(next);
                }
#ifdef EMU_TRACE
                if (!(a.ablate & 1)) issue_weights(nplan, ns - nt * chunks, (L + 1) & 1);
                if (!(a.ablate & 2)) load_patch(nplan, ns - nt * chunks, pv);
#else
                issue_weights(nplan, ns - nt * chunks, (L + 1) & 1);
                load_patch(nplan, ns - nt * chunks, pv);
#endif
            }
            EMU_STAMP(3);
            const uint4 *bq = reinterpret_cast<const uint4 *>(lds + G::B_OFF + (PBUF == 2 ? (L & 1) : 0) * G::BSZ) + bshift;
            const uint4 *wq = reinterpret_cast<const uint4 *>(lds + G::W_OFF + (L & 1) * G::WSZ) + wlane;
            constexpr int NS = G::STEPS;               // MFMA steps of this interval: step = (8-channel chunk h, tap pair s); TAPK: step = tap
            auto load_b = [&](int st, bf16x8 (&b)[TERMS]) {
                const int h = TAPK ? 0 : st / kSteps, s = TAPK ? st : st % kSteps;
#pragma unroll
                for (int t = 0; t < TERMS; ++t) {
                    uint4 v = bq[(h * TERMS + t) * G::PIX + boff[s]];
                    if (!TAPK && s == kSteps - 1 && half) v = uint4{0, 0, 0, 0};       // the tenth tap does not exist
                    b[t] = __builtin_bit_cast(bf16x8, v);
                }
            };
            auto load_w = [&](int st, bf16x8 (&w)[G::NCO][TERMS]) {
                const int h = TAPK ? 0 : st / kSteps, s = TAPK ? st : st % kSteps;
#pragma unroll
                for (int q = 0; q < G::NCO; ++q)
#pragma unroll
                    for (int t = 0; t < TERMS; ++t) w[q][t] = __builtin_bit_cast(bf16x8, wq[h * G::WQ + ((s * TERMS + t) * 2) * kCoutTile + q * 32]);
            };
            bf16x8 bc[TERMS], wc[G::NCO][TERMS];
#ifdef EMU_TRACE
            if (wave_live && !(a.ablate & 4)) {
#else
            if (wave_live) {
#endif
            load_b(0, bc);
            load_w(0, wc);
#pragma unroll
            for (int st = 0; st < NS; ++st) {
                bf16x8 wn[G::NCO][TERMS], bn[TERMS];
                constexpr int NT = TERMS == 3 ? 6 : 3;
                constexpr int wi[6] = {0, 1, TERMS == 3 ? 2 : 0, 0, 1, 0};                       // weight term of product i
                constexpr int bi[6] = {TERMS == 3 ? 2 : 1, TERMS == 3 ? 1 : 0, 0, 1, 0, 0};      // pixel term of product i
                if (st + 1 < NS) {                     // operands of the next step are in flight while this step's MFMAs issue
                    load_b(st + 1, bn);
                    load_w(st + 1, wn);
                }
#pragma unroll
                for (int i = 0; i < NT; ++i)
#pragma unroll
                    for (int q = 0; q < G::NCO; ++q) {
                        if (DUAL && i < 2) accl[q] = mfma16<F16>(wc[q][wi[i]], bc[bi[i]], accl[q]);       // w_h x_l', w_l' x_h: both carry 2^10
                        else acc[q] = mfma16<F16>(wc[q][wi[i]], bc[bi[i]], acc[q]);
                    }
                if (st + 1 < NS) {
#pragma unroll
                    for (int t = 0; t < TERMS; ++t) {
                        bc[t] = bn[t];
#pragma unroll
                        for (int q = 0; q < G::NCO; ++q) wc[q][t] = wn[q][t];
                    }
                }
            }
            }
            EMU_STAMP(4);
            if (PBUF == 1) __syncthreads();            // every wave is done reading the single patch buffer
            EMU_STAMP(5);
            if (more) store_patch(nplan, (L + 1) & 1, pv);
            EMU_STAMP(6);
        }
        // The hand-over of a split tile uses agent-scope *write-through* stores / L2-bypassing loads (relaxed atomics) and no
        // fences (an agent-scope fence writes back and invalidates the XCD's whole L2; see conv3x3.hip).
        // Why this is ordered although every access is memory_order_relaxed (the C++ model alone does not give it; the ISA does):
        //   producer  (1) the partial sums leave as agent-scope atomic stores = write-through (sc1), they never sit dirty in the
        //                 XCD-private L2;  (2) s_waitcnt vmcnt(0): each wave's stores have been ACKNOWLEDGED, i.e. are visible at agent
        //                 scope;  (3) the workgroup barrier: that holds for all waves;  (4) only then is the flag store issued -- stores
        //                 are not speculated, so "flag visible" implies "partials visible".
        //   consumer  (5) one thread spins on an agent-scope atomic load of the flag (bypasses L2, sees the coherent value);  (6) the
        //                 barrier releases the workgroup, and __syncthreads is a compiler barrier for memory operations;  (7) the
        //                 partials are read by agent-scope atomic loads ISSUED after the barrier: they bypass the cache, so no stale
        //                 line can satisfy them, and the hardware neither hoists nor speculates vector loads across s_barrier.
        //   flags are zeroed by a kernel earlier on the stream (fill_words) and each (launch, g) writes its flag once.
        // Guarded by tests/test_round3_gpu.py::test_stream_k_handover_stress (thousands of launches next to a second busy stream,
        // bit-equality every time); the same argument and test cover conv3x3.hip.
        // slot layout [wave][q][lane]: one base pointer per 16 values + immediate offsets (q x 256 B), so the 32 addresses cost
        // four registers, not sixty-four
        if constexpr (DUAL) {
#pragma unroll
            for (int q = 0; q < 16 * G::NCO; ++q) acc[q / 16][q % 16] = fmaf(accl[q / 16][q % 16], coalign::kSp16LowInv, acc[q / 16][q % 16]);
        }
        if (SPLIT && !head) {              // contributor: publish the partial sums of the tile's last chunks (slot g)
            float *slot = a.partial + (size_t)g * (16 * G::NCO * G::THREADS) + (size_t)wave * (16 * G::NCO * 64) + lane;
#pragma unroll
            for (int q = 0; q < 16 * G::NCO; ++q)
                __hip_atomic_store(slot + (q / 16) * 1024 + (q % 16) * 64, acc[q / 16][q % 16], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_s_waitcnt(0);     // my write-throughs are acknowledged ...
            __syncthreads();                   // ... and so are everyone's
            if (tid == 0) __hip_atomic_store(a.flags + g, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (SPLIT && !complete) {      // owner of a split tile: add what workgroup g + 1 published
                if (tid == 0)
                    while (__hip_atomic_load(a.flags + g + hand, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) __builtin_amdgcn_s_sleep(2);
                __syncthreads();
                const float *slot = a.partial + (size_t)(g + hand) * (16 * G::NCO * G::THREADS) + (size_t)wave * (16 * G::NCO * 64) + lane;
#pragma unroll
                for (int q = 0; q < 16 * G::NCO; ++q)
                    acc[q / 16][q % 16] += __hip_atomic_load(slot + (q / 16) * 1024 + (q % 16) * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if constexpr ((LAYOUT & LAYOUT_OUT_SP) != 0) {
                // SP map output (csrc/conv3x3_sp.hip): the sp16 pair of every value, [N][Cout / 16][2 * channel half + term][H][W][8 x fp16].  A lane holds 4 of a
                // group's 8 channels (its partner lane, 32 further, the other 4): one v_permlane32_swap per dword makes lanes 0-31 hold the h terms of all 8,
                // lanes 32-63 the l terms -- one 16-byte store per lane and group.
                static_assert(DUAL, "SP maps belong to the fp16 split");
                uint4 *ysp = reinterpret_cast<uint4 *>(a.y);
                const size_t pix = live ? (size_t)gy * a.W + gx : 0;
                const int on = live ? out_n : 0;
                float vmax = 0.f;
#pragma unroll
                for (int r = 0; r < 4 * G::NCO; ++r) {
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int c = (r / 4) * 32 + 8 * (r % 4) + j;
                        v[j] = acc[r / 4][4 * (r % 4) + j] * winv[c] + bias[c];       // (SP outputs take no residual)
                        if (a.relu) v[j] = fmaxf(v[j], 0.f);
                    }
                    vmax = fmaxf(vmax, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
                    unsigned h01, l01, h23, l23;
                    coalign::sp16_split2(v[0], v[1], h01, l01);
                    coalign::sp16_split2(v[2], v[3], h23, l23);
                    const auto s0 = __builtin_amdgcn_permlane32_swap(h01, l01, false, false);
                    const unsigned a0 = s0[0], b0 = s0[1];
                    const auto s1 = __builtin_amdgcn_permlane32_swap(h23, l23, false, false);
                    const unsigned a1 = s1[0], b1 = s1[1];
                    const int g8 = (cur.cg * kCoutTile + cb + (r / 4) * 32) / 8 + r % 4;          // 8-channel group of the output
                    const size_t idx = ((size_t)(on * (a.Cout / 16) + g8 / 2) * 4 + (g8 % 2) * 2 + half) * plane + pix;
                    if (live) ysp[idx] = uint4{a0, a1, b0, b1};
                }
                if (a.range_flag && live && vmax > 65504.f) atomicOr(a.range_flag, 1);      // (the pair clamps: finite, never NaN -- the host reads the word)
            } else if (live) {
                if constexpr ((LAYOUT & LAYOUT_OUT_NHWC) != 0) {      // channels-last output: accumulators 4 r .. 4 r + 3 are 4 consecutive channels
                    float *yp = a.y + (((size_t)out_n * a.H + gy) * a.W + gx) * a.Cout + cur.cg * kCoutTile + cb + 4 * half;
#pragma unroll
                    for (int r = 0; r < 4 * G::NCO; ++r) {
                        float4 o;
                        o.x = acc[r / 4][4 * (r % 4)]; o.y = acc[r / 4][4 * (r % 4) + 1]; o.z = acc[r / 4][4 * (r % 4) + 2]; o.w = acc[r / 4][4 * (r % 4) + 3];
                        if constexpr (DUAL) {
                            const int c0 = (r / 4) * 32 + 8 * (r % 4);
                            const float *wi4 = winv + c0, *b4 = bias + c0;
                            float r4[4] = {0.f, 0.f, 0.f, 0.f};
                            if (a.residual) {
#pragma unroll
                                for (int j = 0; j < 4; ++j) r4[j] = a.residual[obase + (size_t)(c0 + j) * plane];
                            }
                            o.x = o.x * wi4[0] + (r4[0] + b4[0]); o.y = o.y * wi4[1] + (r4[1] + b4[1]); o.z = o.z * wi4[2] + (r4[2] + b4[2]); o.w = o.w * wi4[3] + (r4[3] + b4[3]);
                        }
                        if (a.relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
                        *reinterpret_cast<float4 *>(yp + (r / 4) * 32 + 8 * (r % 4)) = o;
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < 16 * G::NCO; ++q) {
                        const int c = (q / 16) * 32 + 8 * ((q % 16) / 4) + (q % 4);
                        const float v = DUAL ? acc[q / 16][q % 16] * winv[c] + ((a.residual ? a.residual[obase + (size_t)c * plane] : 0.f) + bias[c]) : acc[q / 16][q % 16];
                        a.y[obase + (size_t)c * plane] = a.relu ? fmaxf(v, 0.f) : v;
                    }
                }
            }
        }
        cur = next;
        plan = nplan;
        tile = ntile;
    }
#ifdef EMU_TRACE
    if (tid == 0) a.trace[2 * 16 * 64 * 8 + 2 * g + 1] = wall_clock64();
#endif
}

struct Launch {                    // what the host needs to know about one (shape, geometry) pair
    int grid;
    size_t flag_bytes, ws_bytes;
    bool split;
};

// stream-K pays when the whole-tile schedule leaves the last round badly filled; a tile must have at least two chunks to split
inline bool want_split(int total_tiles, int slots, int chunks, int long_tile = 32) {
    static const int force = coalign::lab_env("COALIGN_EMU_SPLIT", -1);     // laboratory build only
    if (chunks < 2) return false;
    if (force >= 0) return force != 0;
    // measured (tools/bench_conv_emu_geo.py with COALIGN_EMU_SPLIT=0|1): as for the fp32 kernel, splitting pays on long tiles
    // (>= 32 chunks: the shrink header, 4-11 %) whose last round is under-filled; short tiles would nearly all be split and the
    // extra tile starts (residual fetch, epilogue) cost more than the imbalance
    if (chunks < long_tile || total_tiles <= slots) return false;       // long_tile: 32 chunks of 8 channels = 16 tap-major intervals
    const int rounds = (total_tiles + slots - 1) / slots;
    return total_tiles * 10 < rounds * slots * 9;
}

// Output rows per workgroup tile (= wavefronts per workgroup).  Measured (tools/ab_bench.sh, round 2): tiles fitted to the map height
// (13 rows for the 25-row maps, 10 for 50 / 100) cut the padded work by up to 19 % per layer and changed nothing in the pipeline
// (259 vs 260 frames/s): a workgroup owns its CU (94 KB of LDS with the 3-way split), so what a layer costs is CUs x time, and
// 13 wavefronts on 4 SIMDs quantise as badly as 26 rows on 8-row tiles.  The rule below is the round-1 one.
inline int rows_per_tile(int H, int terms) { return (terms == 3 && H >= 64) ? 12 : 8; }

// the strided / channels-last variants: whole tiles only (no stream-K), same persistent-workgroup schedule
template <int BH, int BW, int NPB, int TERMS, int KCH, int STRIDE, int LAYOUT, int PBUF = 2, int VAR = 0>
int launch_variant(const EmuArgs &a0, hipStream_t s) {
    using G = Geo<BH, BW, NPB, TERMS, KCH, STRIDE, PBUF, (VAR & VAR_TAPK) != 0, (VAR >> 2) & 7>;
    static_assert(G::LDS_BYTES <= 160 * 1024, "geometry does not fit the 160 KB LDS");
    static int resident = 0, cus = 0;
    auto kern = conv3x3_emu_kernel<BH, BW, NPB, TERMS, KCH, false, STRIDE, LAYOUT, PBUF, VAR>;
    if (!resident) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) prop.multiProcessorCount = 256;
        cus = prop.multiProcessorCount;
        const int rc = coalign::hip_call(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES));
        if (rc != COALIGN_OK) {
            (void)hipGetLastError();
            return rc;
        }
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kern, G::THREADS, G::LDS_BYTES) != hipSuccess || n < 1) n = 1;
        resident = n;
    }
    EmuArgs a = a0;
    a.tiles_x = (a.W + G::TW - 1) / G::TW;
    if constexpr (G::STACK) {                                  // the batch as one image of N * H rows: decode() then yields n = 0, y0 = stacked row
        if (G::TH > a.H) return COALIGN_ERR_UNSUPPORTED;       // at most one image boundary per tile
        a.tiles_per_img = a.tiles_x * ((a.N * a.H + G::TH - 1) / G::TH);
        a.total_tiles = a.tiles_per_img * (a.Cout / kCoutTile);
    } else {
        a.tiles_per_img = a.tiles_x * ((a.H + G::TH - 1) / G::TH);
        a.total_tiles = a.tiles_per_img * (a.Cout / kCoutTile) * a.N;
    }
    const int slots = cus * resident;
    const int grid = a.total_tiles < slots ? a.total_tiles : slots;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(G::THREADS), G::LDS_BYTES, s, a);
    return COALIGN_OK;
}

template <int TERMS, bool F16 = false>
int dispatch_variant(const EmuArgs &a, int stride, int layout, hipStream_t s) {
    if constexpr (F16) {                      // fp16 2-way split: the strided layers, 8 output rows per workgroup, double-buffered patch (as the bf16 2-way split)
        if (stride != 2) return COALIGN_ERR_UNSUPPORTED;
#ifdef COALIGN_LAB      // laboratory: 16 channels per barrier interval (64 contiguous bytes per channels-last pixel, half the barriers) with ONE patch buffer (151 KB of LDS)
        static const int kch2 = coalign::lab_env("COALIGN_EMU_S2_KCH2", 0);
        if (kch2 && a.Cin % 16 == 0 && layout == (LAYOUT_IN_NHWC | LAYOUT_OUT_SP)) return launch_variant<1, 32, 8, 2, 2, 2, LAYOUT_IN_NHWC | LAYOUT_OUT_SP, 1, VAR_F16>(a, s);
#endif
        if (layout == LAYOUT_NCHW) return launch_variant<1, 32, 8, 2, 1, 2, LAYOUT_NCHW, 2, VAR_F16>(a, s);
        if (layout == LAYOUT_IN_NHWC) return launch_variant<1, 32, 8, 2, 1, 2, LAYOUT_IN_NHWC, 2, VAR_F16>(a, s);
        if (layout == LAYOUT_NHWC) return launch_variant<1, 32, 8, 2, 1, 2, LAYOUT_NHWC, 2, VAR_F16>(a, s);
        if (layout == LAYOUT_OUT_SP) return launch_variant<1, 32, 8, 2, 1, 2, LAYOUT_OUT_SP, 2, VAR_F16>(a, s);
        if (layout == (LAYOUT_IN_NHWC | LAYOUT_OUT_SP)) {       // the product's strided layers: dense input -> quad staging, one patch buffer, two workgroups per CU (see the kernel)
            if (a.stamps) return launch_variant<1, 32, 8, 2, 1, 2, LAYOUT_IN_NHWC | LAYOUT_OUT_SP, 2, VAR_F16>(a, s);      // (one buffer + two workgroups per CU without the quad staging: 73.4 -> 76.1 us, 715 tiles fill 512 slots worse than 256)
            return launch_variant<1, 32, 8, 2, 1, 2, LAYOUT_IN_NHWC | LAYOUT_OUT_SP, 1, VAR_F16 | VAR_QUAD>(a, s);
        }
        return COALIGN_ERR_UNSUPPORTED;
    }
    if (stride == 2) {
        // the input halo patch of a strided tile is (2 TH + 1) x 65 pixels: 6 rows per workgroup with the 3-way split (142 KB of LDS),
        // 8 with the 2-way split
        constexpr int NPB2 = TERMS == 3 ? 6 : 8;
        // 3-way split: 8 output rows per workgroup with ONE patch buffer (114 KB) instead of 6 rows double buffered (142 KB): 8 wavefronts per
        // CU instead of 6; measured 287.4 -> 291.1 frames/s (tools/ab_bench.sh, same box)
        static const int s2pb = coalign::lab_env("COALIGN_EMU_S2_PBUF1", 1);
        if (TERMS == 3 && s2pb) {
            if (layout == LAYOUT_NCHW) return launch_variant<1, 32, 8, TERMS, 1, 2, LAYOUT_NCHW, 1>(a, s);
            if (layout == LAYOUT_IN_NHWC) return launch_variant<1, 32, 8, TERMS, 1, 2, LAYOUT_IN_NHWC, 1>(a, s);
            if (layout == LAYOUT_NHWC) return launch_variant<1, 32, 8, TERMS, 1, 2, LAYOUT_NHWC, 1>(a, s);      // round 4: channels-last in AND out (feeds the Winograd layers)
        }
        if (layout == LAYOUT_NCHW) return launch_variant<1, 32, NPB2, TERMS, 1, 2, LAYOUT_NCHW>(a, s);
        if (layout == LAYOUT_IN_NHWC) return launch_variant<1, 32, NPB2, TERMS, 1, 2, LAYOUT_IN_NHWC>(a, s);
        if (layout == LAYOUT_NHWC) return launch_variant<1, 32, NPB2, TERMS, 1, 2, LAYOUT_NHWC>(a, s);
        return COALIGN_ERR_UNSUPPORTED;
    }
    if (layout == LAYOUT_OUT_NHWC) {
        if (rows_per_tile(a.H, TERMS) == 12) return launch_variant<1, 32, 12, TERMS, 1, 1, LAYOUT_OUT_NHWC>(a, s);
        return launch_variant<1, 32, 8, TERMS, 1, 1, LAYOUT_OUT_NHWC>(a, s);
    }
    return COALIGN_ERR_UNSUPPORTED;
}

template <int BH, int BW, int NPB, int TERMS, int KCH, int PBUF = 2, int VAR = 0, int LAYOUT = LAYOUT_NCHW>
int launch(const EmuArgs &a0, void *workspace, size_t workspace_bytes, hipStream_t s, Launch *query) {
    constexpr bool TAPK = (VAR & VAR_TAPK) != 0;
    static_assert(!(VAR & (VAR_STACK | VAR_NCO1)), "stacked / 32-channel variants go through launch_variant (whole tiles)");
    using G = Geo<BH, BW, NPB, TERMS, KCH, 1, PBUF, TAPK>;
    static_assert(!TAPK || G::LDS_BYTES <= 160 * 1024, "geometry does not fit the 160 KB LDS");
    static int resident = 0, cus = 0;
    if (!resident) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) prop.multiProcessorCount = 256;
        cus = prop.multiProcessorCount;
        for (int sp = 0; sp < 2; ++sp) {
            const void *fn = sp ? reinterpret_cast<const void *>(conv3x3_emu_kernel<BH, BW, NPB, TERMS, KCH, true, 1, LAYOUT, PBUF, VAR>)
                                : reinterpret_cast<const void *>(conv3x3_emu_kernel<BH, BW, NPB, TERMS, KCH, false, 1, LAYOUT, PBUF, VAR>);
            const int rc = coalign::hip_call(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES));
            if (rc != COALIGN_OK) {                // geometry does not fit this device's LDS: report, leave no sticky error behind
                (void)hipGetLastError();
                return rc;
            }
        }
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv3x3_emu_kernel<BH, BW, NPB, TERMS, KCH, true, 1, LAYOUT, PBUF, VAR>, G::THREADS, G::LDS_BYTES) != hipSuccess || n < 1) n = 1;
        resident = n;
    }
    EmuArgs a = a0;
    a.tiles_x = (a.W + G::TW - 1) / G::TW;
    a.tiles_per_img = a.tiles_x * ((a.H + G::TH - 1) / G::TH);
    a.total_tiles = a.tiles_per_img * (a.Cout / kCoutTile) * a.N;
    const int slots = cus * resident, chunks = a.Cin / (kKC * KCH);
    Launch l;
    l.split = want_split(a.total_tiles, slots, chunks, TAPK ? 16 : 32);
    l.grid = a.total_tiles < slots ? a.total_tiles : slots;
    if (l.split && a.total_tiles < slots) {
        // fewer tiles than slots: an exact two-way split of every tile (ranges of chunks / 2 steps: range 2t opens tile t,
        // range 2t + 1 finishes it) when that fits; otherwise whole tiles
        if (2 * a.total_tiles <= slots && chunks % 2 == 0) l.grid = 2 * a.total_tiles;
        else l.split = false;
    }
    const int n_groups = a.Cout / kCoutTile;
    if (l.split) {                                                // stream-K runs in gangs of one workgroup per output-channel group (see the kernel): whole gangs only
        l.grid = l.grid / n_groups * n_groups;
        if (l.grid < n_groups) l.split = false, l.grid = a.total_tiles < slots ? a.total_tiles : slots;
    }
    l.flag_bytes = coalign::align_up((size_t)(l.grid + n_groups) * sizeof(int), 256);
    l.ws_bytes = l.flag_bytes + (size_t)(l.grid + n_groups) * 16 * G::NCO * G::THREADS * sizeof(float);
    if (query) {
        *query = l;
        return COALIGN_OK;
    }
    if (l.split) {
        if (!workspace) return COALIGN_ERR_NULL_POINTER;
        if (workspace_bytes < l.ws_bytes) return COALIGN_ERR_WORKSPACE;
        a.flags = static_cast<int *>(workspace);
        a.partial = reinterpret_cast<float *>(static_cast<char *>(workspace) + l.flag_bytes);
        const int rc = coalign::fill_words(workspace, l.flag_bytes / 4, 0u, s);
        if (rc != COALIGN_OK) return rc;
        hipLaunchKernelGGL((conv3x3_emu_kernel<BH, BW, NPB, TERMS, KCH, true, 1, LAYOUT, PBUF, VAR>), dim3(l.grid), dim3(G::THREADS), G::LDS_BYTES, s, a);
    } else {
        hipLaunchKernelGGL((conv3x3_emu_kernel<BH, BW, NPB, TERMS, KCH, false, 1, LAYOUT, PBUF, VAR>), dim3(l.grid), dim3(G::THREADS), G::LDS_BYTES, s, a);
    }
    return COALIGN_OK;
}

template <int TERMS>
int dispatch(const EmuArgs &a, void *ws, size_t ws_bytes, hipStream_t s, Launch *query) {
    static const int force = coalign::lab_env("COALIGN_EMU_GEO", -1);      // laboratory build only
    // geometry code = 10 * (wavefronts per workgroup = output rows per tile) + (8-channel chunks per barrier)
    const bool even = (a.Cin / kKC) % 2 == 0;
    int geo = 10 * rows_per_tile(a.H, TERMS) + 1;
    if (force >= 0) geo = force;
    // 3-way split: 8-wavefront workgroups with ONE split-patch buffer (78 KB of LDS: two workgroups per CU) on every map size.
    // Measured in the pipeline (round 2, same box): 258-262 frames/s against 249-250 with the double-buffered 8 / 12-wavefront geometries.
    static const int pbuf1 = coalign::lab_env("COALIGN_EMU_PBUF1", 2);     // laboratory build: 0 = off, 1 = small maps only
    if (force < 0 && TERMS == 3 && ((pbuf1 == 1 && geo == 81) || pbuf1 == 2)) geo = 83;
    // (One weight image + one patch buffer -- 47 KB, three workgroups per CU at 80 registers -- measured 157 vs 286 frames/s: 160 B of
    //  scratch per lane in the chunk loop and the weight DMA exposed between the two barriers.)
    // (4-row tiles with one patch buffer on the 25- and 50-row maps -- one wavefront per SIMD, 28 / 52 padded rows -- measured 281-286
    //  against 285-287 frames/s: no gain.)
    // (9-row tiles -- 79.4 KB, nominally still two per CU, 27 / 54 / 108 padded rows -- measured 256 vs 283 frames/s: 18 wavefronts do not
    //  spread over four SIMDs at this kernel's 95 registers, the second workgroup does not become resident.)
    if (!even && geo % 10 == 2) geo -= 1;
    switch (geo) {
        case 81: return launch<1, 32, 8, TERMS, 1>(a, ws, ws_bytes, s, query);             // (the 2-way split)
        case 83: return launch<1, 32, 8, TERMS, 1, 1>(a, ws, ws_bytes, s, query);          // one patch buffer: two workgroups per CU with the 3-way split
#ifdef COALIGN_LAB      // geometries measured and not adopted: laboratory build only
        case 82: return launch<1, 32, 8, TERMS, 2>(a, ws, ws_bytes, s, query);
        case 84: return launch<1, 32, 8, TERMS, 1, 1, VAR_ASM_DMA>(a, ws, ws_bytes, s, query);      // ... with the weight DMA hidden from hipcc's waitcnt pass
        case 121: return launch<1, 32, 12, TERMS, 1>(a, ws, ws_bytes, s, query);
        case 122: return launch<1, 32, 12, TERMS, 2>(a, ws, ws_bytes, s, query);
        case 41: return launch<1, 32, 4, TERMS, 1>(a, ws, ws_bytes, s, query);
#endif
        default: return COALIGN_ERR_UNSUPPORTED;
    }
}

// Tap-major weight image (TAPK, "K = 144"): 16-channel intervals of nine steps, one workgroup per CU (111 KB of weights + one split
// patch).  Output rows per workgroup = wavefronts: 8 (143 KB), 10 (150 KB) or 12 (156 KB) with the 3-way split.
template <int TERMS, int BH, int BW, int NPB, int VAR>
int tapk_launch(const EmuArgs &a, int layout, void *ws, size_t ws_bytes, hipStream_t s, Launch *query) {
    if (layout == LAYOUT_OUT_NHWC) {
        if (query) {
            *query = Launch{0, 0, 0, false};
            return COALIGN_OK;
        }
        return launch_variant<BH, BW, NPB, TERMS, 2, 1, LAYOUT_OUT_NHWC, 1, VAR>(a, s);
    }
    if constexpr ((VAR & VAR_F16) != 0) {          // SplitMap output: the same whole-tile / stream-K choice as the NCHW output (the shrink header's first convolution)
        if (layout == LAYOUT_OUT_SP) return launch<BH, BW, NPB, TERMS, 2, 1, VAR, LAYOUT_OUT_SP>(a, ws, ws_bytes, s, query);
    }
    if (layout != LAYOUT_NCHW) return COALIGN_ERR_UNSUPPORTED;
    return launch<BH, BW, NPB, TERMS, 2, 1, VAR>(a, ws, ws_bytes, s, query);
}

// Stacked tiles (round 3): the batch is tiled as one image of N * H rows, whole tiles only.  Same output layouts as tapk_launch.
template <int TERMS, int BH, int BW, int NPB, int VAR>
int tapk_stacked(const EmuArgs &a, int layout, hipStream_t s, Launch *query) {
    if (query) {
        *query = Launch{0, 0, 0, false};
        return COALIGN_OK;
    }
    if (layout == LAYOUT_OUT_NHWC) return launch_variant<BH, BW, NPB, TERMS, 2, 1, LAYOUT_OUT_NHWC, 1, VAR>(a, s);
    if (layout == LAYOUT_NCHW) return launch_variant<BH, BW, NPB, TERMS, 2, 1, LAYOUT_NCHW, 1, VAR>(a, s);
    if constexpr ((VAR & VAR_F16) != 0) {
        if (layout == LAYOUT_OUT_SP) return launch_variant<BH, BW, NPB, TERMS, 2, 1, LAYOUT_OUT_SP, 1, VAR>(a, s);
    }
    return COALIGN_ERR_UNSUPPORTED;
}

template <int TERMS, int VAR>
int tapk_rows(int rows, const EmuArgs &a, int layout, void *ws, size_t ws_bytes, hipStream_t s, Launch *query) {
    switch (rows) {
        case 8: return tapk_launch<TERMS, 1, 32, 8, VAR>(a, layout, ws, ws_bytes, s, query);
        case 12: return tapk_launch<TERMS, 1, 32, 12, VAR>(a, layout, ws, ws_bytes, s, query);
        case 26: return tapk_launch<TERMS, 2, 16, 13, VAR>(a, layout, ws, ws_bytes, s, query);       // 13 wavefronts x (2 rows x 16 pixels): 26 x 16 tiles
        // stacked: 12 wavefronts x (2 rows x 16 pixels) = 24 x 16 tiles over N * H rows (the 50 x 176 maps: 11 x 11 x 2 = 242 tiles of 12 units)
        case 124: return tapk_stacked<TERMS, 2, 16, 12, VAR | VAR_STACK>(a, layout, s, query);
        // stacked, 32 output channels per wavefront: 6 rows x 32 pixels x 64 channels per workgroup = 12 wavefronts of half a unit (the 25 x 88
        // maps: 21 x 3 x 4 = 252 tiles, three half units per SIMD instead of two whole ones)
#ifdef COALIGN_LAB      // (opt-in since round 3, never the default: laboratory build only)
        case 106: return tapk_stacked<TERMS, 1, 32, 6, VAR | VAR_STACK | VAR_NCO1>(a, layout, s, query);
#endif
        // stacked, 4 x 8-pixel blocks in four block columns: 8 x 32 tiles whose fourth block column is idle where the map ends after 88 columns
        case 148: return tapk_stacked<TERMS, 4, 8, 8, VAR | VAR_STACK | VAR_NBX4>(a, layout, s, query);
        default: return COALIGN_ERR_UNSUPPORTED;
    }
}

template <int TERMS, bool F16 = false>
int dispatch_tapk(const EmuArgs &a, int layout, void *ws, size_t ws_bytes, hipStream_t s, Launch *query) {
    if constexpr (F16) {                      // fp16 2-way split on the tap-major image: 12 / 8 rows per workgroup or the 26 x 16 tiles, as the bf16 2-way split
        if (a.Cin % (2 * kKC)) return COALIGN_ERR_UNSUPPORTED;
        int rows = (a.W % 32 == 16 && a.H > 26 && a.H <= 52) ? 26 : (a.H >= 64 ? 12 : 8);
        // stacked tiles as for the 3-way split (see below): the 24 x 16 tiles on the 50 x 176 maps (bit 0), the 4 x 8-pixel blocks on the 25 x 88 maps (bit 2).
        // Measured with the fp16 split (round 4, same box, alternating; bit-equal outputs): 50 x 176 layer 62.1 -> 57.0 us alone; whole frame 486 / 494 (off)
        // -> 496 / 501 frames/s (both bits), one frame in flight 2.57 -> 2.55 ms; bit 0 alone 480 / 491.
        static const int stack16 = coalign::lab_env("COALIGN_EMU_STACK16", 5);
        if (rows == 26 && a.H >= 24 && (stack16 & 1)) rows = 124;
        else if (rows == 8 && a.H >= 8 && a.H <= 32 && a.W % 32 > 0 && a.W % 32 <= 24 && (stack16 & 4)) rows = 148;
        return tapk_rows<2, VAR_TAPK | VAR_ASM_DMA | VAR_F16>(rows, a, layout, ws, ws_bytes, s, query);
    }
    static const int force = coalign::lab_env("COALIGN_EMU_TAPK_ROWS", 0);      // laboratory build only
    if (a.Cin % (2 * kKC)) return COALIGN_ERR_UNSUPPORTED;
    // measured per layer (tools/bench_conv_tapk.py): 12 rows on the 100-row maps (495 tiles = two full rounds of 256 workgroups), 8 rows
    // on the 25-row maps; 10 rows lose everywhere but on the shrink header (-1 %)
    // 50 x 176 maps (W a multiple of 16, not of 32): 13 wavefronts of 2 rows x 16 pixels = 26 x 16 tiles cover the map exactly (220 tiles =
    // ONE round; 8 x 32 tiles: 420 = 1.6 rounds, 9 % dead columns): 95 vs 108 us per layer
    static const int t26 = coalign::lab_env("COALIGN_EMU_TAPK_26", 1);
    // Round 3, load balance.  A workgroup owns its CU (143-159 KB of LDS), so a layer costs (busiest SIMD's work) x time: with per-image tiles
    // the 25 x 88 maps of a 5-agent frame are 240 tiles of 8 wavefront units (two per SIMD) of which 60 hold a single live row (the 25th) and
    // still keep their CU for the whole K loop; the 50 x 176 maps 220 tiles of 13 units (four on one SIMD).  Tiling the batch as ONE image
    // of N * H rows removes the per-image remainder: 252 tiles of 6 rows x 32 pixels with 32-channel wavefronts = three HALF units per SIMD
    // (1.5 instead of 2); 242 tiles of 24 x 16 pixels = three units per SIMD (instead of four).  COALIGN_EMU_STACK=0: the per-image tiles.
    // Bit 2: the 25 x 88 maps in 4 x 8-pixel blocks, four block columns per tile (8 x 32-pixel tiles over the stacked batch): the map ends inside the
    // third tile column after three of its four block columns, so 11 block columns are computed for 88 pixels instead of 12 (the 1 x 32 blocks
    // compute 96): -6 % matrix instructions on these layers, same time per layer alone (87.0 vs 86.4 us) at 6.6 % less board power (1202 vs
    // 1287 W) -- and, the board being at its power limit whenever three frames are in flight, 352-353 vs 336-338 frames/s (same box, alternating).
    // DEFAULT = 5: bits 0 and 2.  The 6 x 32 / 32-channel variant (bit 1; takes the 25-row maps when bit 2 does not) is 9 % faster on its layer in
    // isolation (94 vs 103 us) and bit-equal to the per-image tiles; in the 3-lane frame pipeline -1 ... +2 % depending on the box (it executes the same
    // matrix instructions: balance inside a launch is not what a power-limited pipeline is short of), with one frame in flight 3.55 vs 3.66 ms per
    // frame.  It stays opt-in: it is the kernel that exposed the packed-fp32 hazard written up in profiles/round3/README.md -- a wavefront of
    // ANOTHER kernel that shares a SIMD with this variant's three matrix wavefronts got wrong results in lanes 48-63 of its v_pk_mul_f32 /
    // v_pk_add_f32 instructions (warp_fuse_nhwc: ~1/3 of fused maps differed).  Every kernel of this library is now built without packed fp32
    // instructions (build.py), after which tools/diag_fuse_corun.py and a 3000-frame soak (tools/soak_pipeline.py) show no difference with the
    // variant on; kernels that are not ours and share the GPU may still contain them.
    static const int stack = coalign::lab_env("COALIGN_EMU_STACK", 5);
    int rows = force ? force : (t26 && a.W % 32 == 16 && a.H > 26 && a.H <= 52) ? 26 : (a.H >= 64 ? 12 : 8);
    if (!force && stack && TERMS == 3) {          // (bit 0: the 24 x 16 tiles, bit 1: the 6 x 32 tiles, bit 2: the 4 x 8-pixel blocks -- separately switchable)
        if (rows == 26 && a.H >= 24 && (stack & 1)) rows = 124;
        else if (rows == 8 && a.H >= 8 && a.H <= 32 && a.W % 32 > 0 && a.W % 32 <= 24 && (stack & 4)) rows = 148;
        else if (rows == 8 && a.H >= 6 && a.H <= 32 && (stack & 2)) rows = 106;
    }
    // (the weight DMA is always issued from inline assembly: with the builtin, hipcc waits vmcnt(0) in front of the first matrix step of every interval)
    return tapk_rows<TERMS, VAR_TAPK | VAR_ASM_DMA>(rows, a, layout, ws, ws_bytes, s, query);
}

constexpr int kLayoutTapMajor = 4;      // COALIGN_LAYOUT_W_TAPMAJOR: flag bit of `layout`
constexpr int kLayoutOutSp = 8;         // COALIGN_LAYOUT_OUT_SP: flag bit of `layout` (the kernels' template bit is LAYOUT_OUT_SP = 4)

}  // namespace

#ifdef EMU_TRACE
static long long *g_emu_trace = nullptr;
static int g_emu_ablate = 0;
extern "C" void coalign_conv3x3_emu_set_trace(long long *p) { g_emu_trace = p; }
extern "C" void coalign_conv3x3_emu_set_ablate(int v) { g_emu_ablate = v; }
#endif

extern "C" size_t coalign_conv3x3_emu_weight_bytes(int Cin, int Cout, int terms) {
    if (Cin < 1 || Cout < 1 || Cin % kKC || Cout % kCoutTile || (terms != 2 && terms != 3 && terms != 16)) return 0;
    const size_t tail = terms == 16 ? (size_t)Cout * 8 : 0;      // fp16 split: + [Cout] 2^-k_c + [Cout] 2^k_c (float), the per-channel weight scale
    if (terms == 16) terms = 2;           // two 16-bit terms
    return (size_t)(Cout / kCoutTile) * (Cin / kKC) * kSteps * terms * 2 * kCoutTile * 16 + 16 + tail;      // + one zero group
}

extern "C" size_t coalign_conv3x3_emu_weight_bytes_ex(int Cin, int Cout, int terms, int tap_major) {
    if (!tap_major) return coalign_conv3x3_emu_weight_bytes(Cin, Cout, terms);
    if (Cin < 1 || Cout < 1 || Cin % (2 * kKC) || Cout % kCoutTile || (terms != 2 && terms != 3 && terms != 16)) return 0;
    const size_t tail = terms == 16 ? (size_t)Cout * 8 : 0;
    if (terms == 16) terms = 2;
    return (size_t)(Cout / kCoutTile) * (Cin / (2 * kKC)) * 9 * terms * 2 * kCoutTile * 16 + 16 + tail;
}

// where the fp16 image's scale tail starts (nullptr for the bf16 splits, which are scale-free)
static const float *emu_wscale(const void *w_split, int Cin, int Cout, int terms, int tap_major) {
    if (terms != 16) return nullptr;
    return reinterpret_cast<const float *>(static_cast<const char *>(w_split) + coalign_conv3x3_emu_weight_bytes_ex(Cin, Cout, 16, tap_major) - (size_t)Cout * 8);
}

static int check_emu_args(int N, int Cin, int Cout, int H, int W, int terms) {
    if (N < 0 || H < 1 || W < 1 || Cin < 1 || Cout < 1) return COALIGN_ERR_BAD_SHAPE;
    if (Cin % kKC || Cout % kCoutTile || (terms != 2 && terms != 3 && terms != 16)) return COALIGN_ERR_UNSUPPORTED;
    if ((int64_t)N * Cout * H * W > (int64_t)1 << 40 || (int64_t)Cin * H * W > (int64_t)1 << 30) return COALIGN_ERR_UNSUPPORTED;
    return COALIGN_OK;
}

extern "C" size_t coalign_conv3x3_emu_workspace_bytes(int N, int Cin, int Cout, int H, int W, int terms) {
    if (check_emu_args(N, Cin, Cout, H, W, terms) != COALIGN_OK || N == 0 || terms == 16) return 0;
    EmuArgs a{nullptr, nullptr, nullptr, nullptr, nullptr, N, Cin, Cout, H, W, 0, 0, 0, 0, H, W, nullptr, nullptr};
    Launch l{};
    const int rc = terms == 3 ? dispatch<3>(a, nullptr, 0, nullptr, &l) : dispatch<2>(a, nullptr, 0, nullptr, &l);
    return rc == COALIGN_OK && l.split ? l.ws_bytes : 0;
}

extern "C" size_t coalign_conv3x3_emu_workspace_bytes_ex(int N, int Cin, int Cout, int H, int W, int terms, int layout) {
    if (!(layout & kLayoutTapMajor)) return (layout & 3) == LAYOUT_NCHW ? coalign_conv3x3_emu_workspace_bytes(N, Cin, Cout, H, W, terms) : 0;
    if (check_emu_args(N, Cin, Cout, H, W, terms) != COALIGN_OK || N == 0) return 0;
    EmuArgs a{nullptr, nullptr, nullptr, nullptr, nullptr, N, Cin, Cout, H, W, 0, 0, 0, 0, H, W, nullptr, nullptr};
    Launch l{};
    const int lay = (layout & kLayoutOutSp) ? LAYOUT_OUT_SP : (layout & 3);
    if (lay == LAYOUT_OUT_SP && terms != 16) return 0;
    const int rc = terms == 3 ? dispatch_tapk<3>(a, lay, nullptr, 0, nullptr, &l) : terms == 16 ? dispatch_tapk<2, true>(a, lay, nullptr, 0, nullptr, &l) : dispatch_tapk<2>(a, lay, nullptr, 0, nullptr, &l);
    return rc == COALIGN_OK && l.split ? l.ws_bytes : 0;
}

static int emu_prio_mode() {
    static const int v = (coalign::lab_env("COALIGN_EMU_PRIO", 0) & 7) | (coalign::lab_env("COALIGN_EMU_XCD", 1) ? 8 : 0);      // bit 3: XCD-aware workgroup order (default on)
    return v;
}

extern "C" int coalign_conv3x3_emu_bias_act(const float *x, const void *w_split, const float *bias, const float *residual, float *y,
                                            int N, int Cin, int Cout, int H, int W, int relu, int terms, void *workspace,
                                            size_t workspace_bytes, void *stream) {
    using namespace coalign;
    if (!x || !w_split || !y || !bias) return COALIGN_ERR_NULL_POINTER;
    int rc = check_emu_args(N, Cin, Cout, H, W, terms);
    if (rc != COALIGN_OK) return rc;
    if (terms == 16) return COALIGN_ERR_UNSUPPORTED;       // the fp16 split serves the tap-major and the strided images only
    if (reinterpret_cast<uintptr_t>(w_split) & 15) return COALIGN_ERR_UNSUPPORTED;
    if (N == 0) return COALIGN_OK;
    EmuArgs a{x, static_cast<const uint4 *>(w_split), bias, residual, y, N, Cin, Cout, H, W, relu, 0, 0, 0, H, W, nullptr, nullptr};
    a.prio_mode = emu_prio_mode();
#ifdef EMU_TRACE
    a.trace = g_emu_trace;
    a.ablate = g_emu_ablate;
#endif
    hipStream_t s = static_cast<hipStream_t>(stream);
    rc = terms == 3 ? dispatch<3>(a, workspace, workspace_bytes, s, nullptr) : dispatch<2>(a, workspace, workspace_bytes, s, nullptr);
    return rc != COALIGN_OK ? rc : check_launch();
}

extern "C" int coalign_conv3x3_emu_ex(const float *x, const void *w_split, const float *bias, const float *residual, float *y, int N, int Cin,
                                      int Cout, int Hin, int Win, int stride, int relu, int terms, int layout, int32_t *range_flag, void *workspace,
                                      size_t workspace_bytes, void *stream) {
    using namespace coalign;
    if (stride == 1 && layout == LAYOUT_NCHW)
        return coalign_conv3x3_emu_bias_act(x, w_split, bias, residual, y, N, Cin, Cout, Hin, Win, relu, terms, workspace, workspace_bytes, stream);
    if (!x || !w_split || !y || !bias) return COALIGN_ERR_NULL_POINTER;
    if (layout >= 0 && (layout & kLayoutTapMajor)) {          // tap-major weight image: stride 1, NCHW in, NCHW or channels-last out
        int lay = layout & 3;
        if (stride != 1 || (lay != LAYOUT_NCHW && lay != LAYOUT_OUT_NHWC) || layout > (kLayoutTapMajor | kLayoutOutSp | 3)) return COALIGN_ERR_UNSUPPORTED;
        if (layout & kLayoutOutSp) {                          // SP map output: fp16 split, NCHW input
            if (terms != 16 || lay != LAYOUT_NCHW || residual || (reinterpret_cast<uintptr_t>(y) & 15)) return COALIGN_ERR_UNSUPPORTED;
            lay = LAYOUT_OUT_SP;
        }
        int rc = check_emu_args(N, Cin, Cout, Hin, Win, terms);
        if (rc != COALIGN_OK) return rc;
        if ((reinterpret_cast<uintptr_t>(w_split) & 15) || (lay != LAYOUT_NCHW && (reinterpret_cast<uintptr_t>(y) & 15))) return COALIGN_ERR_UNSUPPORTED;
        if (N == 0) return COALIGN_OK;
        EmuArgs a{x, static_cast<const uint4 *>(w_split), bias, residual, y, N, Cin, Cout, Hin, Win, relu, 0, 0, 0, Hin, Win, nullptr, nullptr};
        a.prio_mode = emu_prio_mode();
        a.wscale = emu_wscale(w_split, Cin, Cout, terms, 1);
        a.range_flag = range_flag;
#ifdef EMU_TRACE
        a.trace = g_emu_trace;
        a.ablate = g_emu_ablate;
#endif
        hipStream_t s = static_cast<hipStream_t>(stream);
        rc = terms == 3 ? dispatch_tapk<3>(a, lay, workspace, workspace_bytes, s, nullptr) : terms == 16 ? dispatch_tapk<2, true>(a, lay, workspace, workspace_bytes, s, nullptr) : dispatch_tapk<2>(a, lay, workspace, workspace_bytes, s, nullptr);
        return rc != COALIGN_OK ? rc : check_launch();
    }
    if (stride != 1 && stride != 2) return COALIGN_ERR_UNSUPPORTED;
    if (layout >= 0 && (layout & kLayoutOutSp)) {             // SP map output of the strided fp16 variants (NCHW or channels-last input)
        if (terms != 16 || stride != 2 || residual || (layout & ~(kLayoutOutSp | LAYOUT_IN_NHWC))) return COALIGN_ERR_UNSUPPORTED;
        layout = (layout & LAYOUT_IN_NHWC) | LAYOUT_OUT_SP;
    } else if (layout != LAYOUT_NCHW && layout != LAYOUT_OUT_NHWC && layout != LAYOUT_IN_NHWC && !(layout == LAYOUT_NHWC && stride == 2)) return COALIGN_ERR_UNSUPPORTED;
    const int H = (Hin + stride - 1) / stride, W = (Win + stride - 1) / stride;          // 3x3, pad 1: floor((n + 2 - 3) / s) + 1
    int rc = check_emu_args(N, Cin, Cout, Hin, Win, terms);
    if (rc != COALIGN_OK) return rc;
    if ((reinterpret_cast<uintptr_t>(w_split) & 15) || (layout != LAYOUT_NCHW && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15)))
        return COALIGN_ERR_UNSUPPORTED;
    if (N == 0) return COALIGN_OK;
    EmuArgs a{x, static_cast<const uint4 *>(w_split), bias, residual, y, N, Cin, Cout, H, W, relu, 0, 0, 0, Hin, Win, nullptr, nullptr};
    a.prio_mode = emu_prio_mode();
    a.wscale = emu_wscale(w_split, Cin, Cout, terms, 0);
    a.range_flag = range_flag;
#ifdef EMU_TRACE
    a.trace = g_emu_trace;
    a.ablate = g_emu_ablate;
#endif
    hipStream_t s = static_cast<hipStream_t>(stream);
    rc = terms == 3 ? dispatch_variant<3>(a, stride, layout, s) : terms == 16 ? dispatch_variant<2, true>(a, stride, layout, s) : dispatch_variant<2>(a, stride, layout, s);
    return rc != COALIGN_OK ? rc : check_launch();
}

// Round 4: the strided first convolution of the first ResNet stage reading a SPARSE canvas (csrc/pillar_sparse.hip) instead of the dense zero-filled one:
// same kernel, the pixel -> row lookup happens once per tile in the patch plan.  Output channels-last (out_nhwc != 0) or NCHW.
extern "C" int coalign_conv3x3_emu_sparse(const float *feats, int M_rows, const void *stamps, const int32_t *state, const void *w_split, const float *bias, float *y, int N,
                                          int Cin, int Cout, int Hin, int Win, int relu, int terms, int out_nhwc, int32_t *range_flag, void *stream) {
    using namespace coalign;
    if (!feats || !stamps || !state || !w_split || !y || !bias) return COALIGN_ERR_NULL_POINTER;
    const int H = (Hin + 1) / 2, W = (Win + 1) / 2;
    int rc = check_emu_args(N, Cin, Cout, Hin, Win, terms);
    if (rc != COALIGN_OK) return rc;
    if ((reinterpret_cast<uintptr_t>(w_split) | reinterpret_cast<uintptr_t>(feats) | reinterpret_cast<uintptr_t>(y)) & 15) return COALIGN_ERR_UNSUPPORTED;
    if (reinterpret_cast<uintptr_t>(stamps) & 7) return COALIGN_ERR_UNSUPPORTED;
    if (N == 0) return COALIGN_OK;
    EmuArgs a{feats, static_cast<const uint4 *>(w_split), bias, nullptr, y, N, Cin, Cout, H, W, relu, 0, 0, 0, Hin, Win, nullptr, nullptr};
    a.prio_mode = emu_prio_mode();
    a.wscale = emu_wscale(w_split, Cin, Cout, terms, 0);
    a.stamps = static_cast<const unsigned long long *>(stamps);
    a.tag_ptr = state;
    if (M_rows < 0) return COALIGN_ERR_BAD_SHAPE;
    a.sparse_rows = (unsigned)M_rows;
    a.range_flag = range_flag;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (out_nhwc == 2 && terms != 16) return COALIGN_ERR_UNSUPPORTED;      // SP map output: fp16 split only
    const int layout = out_nhwc == 2 ? (LAYOUT_IN_NHWC | LAYOUT_OUT_SP) : out_nhwc ? LAYOUT_NHWC : LAYOUT_IN_NHWC;
    rc = terms == 3 ? dispatch_variant<3>(a, 2, layout, s) : terms == 16 ? dispatch_variant<2, true>(a, 2, layout, s) : dispatch_variant<2>(a, 2, layout, s);
    return rc != COALIGN_OK ? rc : check_launch();
}
