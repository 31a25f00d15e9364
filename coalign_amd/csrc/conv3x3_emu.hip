// 3x3 / stride 1 / pad 1 convolution with fused bias (+ residual) (+ ReLU), fp32 in / fp32 out, computed on the BF16 matrix
// cores by error-free operand splitting ("fp32 emulation"), NCHW, gfx950.  OPT-IN (COALIGN_CONV_EMU, see backbone.py): the
// default product path keeps every product in native fp32.
//
// Same layers and semantics as conv3x3.hip (opencood/models/sub_modules/resblock.py:53-69, base_bev_backbone_resnet.py:59-138,
// downsample_conv.py:7-50).  The fp32 MFMA runs at the VALU rate (157 TFLOP/s); v_mfma_f32_32x32x16_bf16 is 16x faster and
// accumulates in fp32.  Every fp32 operand is written as an exact sum of bf16 numbers
//     x = x_h + x_m + x_l,   x_h = bf16(x),  x_m = bf16(x - x_h),  x_l = bf16(x - x_h - x_m)      (the subtractions are exact)
// and the product w * x is evaluated as the sum of the cross terms, smallest first, each one exact in the fp32 accumulator:
//   TERMS = 3:  w_h x_l + w_m x_m + w_l x_h + w_h x_m + w_m x_h + w_h x_h     dropped terms <= 2^-24 |w x|: fp32-level accuracy
//   TERMS = 2:  w_h x_l + w_l x_h + w_h x_h   (x_l = bf16(x - x_h))           dropped terms <= 2^-16 |w x|
// i.e. 6 (or 3) bf16 MFMAs replace 8 fp32 MFMAs of the same K: 2.7x (5.3x) less matrix-pipe time.  The weights are split on the
// host once; the input pixels are split in registers right after their LDS read (v_cvt_pk_bf16_f32 + exact subtractions).
//
// GEMM view per image:  D[cout, pixel] = sum_{cin, tap} W[cout, cin, tap] * X[cin, pixel + tap].  One MFMA has K = 16 = two taps
// x the 8 input channels of the LDS chunk: lanes 0-31 (k 0..7) carry tap 2s, lanes 32-63 (k 8..15) tap 2s + 1, s = 0..4 (the
// tenth tap is zero weights).  Tiling, persistent workgroups, LDS-DMA double buffering and the epilogue are those of conv3x3.hip.
#include "common.h"
#include <cstdlib>

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kKC = 8;               // input channels per LDS chunk = k values per lane
constexpr int kCoutTile = 64;        // output channels per workgroup
constexpr int kSteps = 5;            // MFMA steps per chunk: taps (0,1) (2,3) (4,5) (6,7) (8,-)

typedef __attribute__((address_space(1))) const void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

struct EmuArgs {
    const float *__restrict__ x;
    const uint4 *__restrict__ wt;     // [Cout / 64][Cin / 8][5 steps][TERMS][2 k-groups][64 cout][8 bf16]
    const float *__restrict__ bias, *__restrict__ residual, *__restrict__ zero;
    float *__restrict__ y;
    int N, Cin, Cout, H, W, relu, tiles_x, tiles_per_img, total_tiles;
#ifdef EMU_TRACE
    long long *trace;                 // profiling aid (tools/trace_conv_emu.py): [2 workgroups][waves][64 chunks][5 stamps]
#endif
};

#ifdef EMU_TRACE
#define EMU_STAMP(k)                                                                                              \
    if ((g == 0 || g == 100) && lane == 0 && L < 64)                                                              \
        a.trace[((((g ? 1 : 0) * G::WAVES + wave) * 64) + L) * 5 + (k)] = (long long)__builtin_amdgcn_s_memtime()
#else
#define EMU_STAMP(k)
#endif

constexpr int pick_stride(int pw, int bh, int bw) {
    int s = (pw + 3) / 4 * 4;
    if (bh == 1) return s;
    while (s % 32 != bw % 32) s += 4;
    return s;
}

template <int BH, int BW, int NPB, int TERMS>
struct Geo {
    static constexpr int NCO = NPB >= 4 ? 2 : 1;
    static constexpr int WAVES = NPB >= 4 ? NPB : 2 * NPB;
    static constexpr int THREADS = 64 * WAVES;
    static constexpr int TH = BH * NPB, TW = BW, PH = TH + 2, PW = TW + 8;
    static constexpr int STR = pick_stride(PW, BH, BW);
    static constexpr int CS = PH * STR;
    static constexpr int PGROUPS = kKC * CS / 4;
    static constexpr int PINSTR = (PGROUPS + 63) / 64;
    static constexpr int PLDS = (PGROUPS * 4 + 255) / 256 * 256;             // floats of the patch image (whole wave transfers)
    static constexpr int WQ = kSteps * TERMS * 2 * kCoutTile;                 // 16-byte groups of one weight chunk
    static constexpr int WINSTR = WQ / 64;
    static constexpr int PIX = PH * PW;                                       // pixel slots of the patch
    // LDS map (floats): three weight images | two fp32 staging patches | two split patches [term][y][x][8 cin] bf16
    static constexpr int W_OFF = 0, S_OFF = 3 * WQ * 4, B_OFF = S_OFF + 2 * PLDS, BSZ = TERMS * PIX * 4;
    static constexpr size_t LDS_BYTES = ((size_t)B_OFF + 2 * (size_t)BSZ) * 4;
};

struct Tile {
    int n, cg, y0, x0;
};

// error-free split of the 8 input channels of one pixel: out[t] = term t of each channel, 8 bf16 = one MFMA operand
template <int TERMS>
__device__ __forceinline__ void split_pixel(const float (&v)[8], bf16x8 (&out)[TERMS]) {
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

template <int BH, int BW, int NPB, int TERMS>
__global__ __launch_bounds__(64 * (NPB >= 4 ? NPB : 2 * NPB)) void conv3x3_emu_kernel(const EmuArgs a) {
    using G = Geo<BH, BW, NPB, TERMS>;
    extern __shared__ __attribute__((aligned(1024))) float lds[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, half = lane >> 5, p = lane & 31;      // wave: scalar
    const size_t plane = (size_t)a.H * a.W;
    const int groups = a.Cout / kCoutTile, chunks = a.Cin / kKC;
    auto decode = [&](int t) {
        Tile c;
        c.cg = t % groups;
        const int sp = t / groups;
        c.n = sp / a.tiles_per_img;
        const int r = sp - c.n * a.tiles_per_img, ty = r / a.tiles_x;
        c.y0 = ty * G::TH;
        c.x0 = (r - ty * a.tiles_x) * G::TW;
        return c;
    };
    const int pb = wave % NPB, cb = (G::NCO == 2 ? 0 : wave / NPB) * 32;
    const int py = pb * BH + p / BW, px = p % BW;
    int boff[kSteps];                                     // pixel slot of this lane's tap in step s (tap 9 -> tap 8, zeroed below)
#pragma unroll
    for (int s = 0; s < kSteps; ++s) {
        const int t = 2 * s + half < 9 ? 2 * s + half : 8;
        boff[s] = (py + t / 3) * G::PW + px + 3 + t % 3;
    }
    const int wlane = half * kCoutTile + cb + p;          // 16-byte group of this lane inside one (step, term) weight block

    // LDS-DMA plan of a tile: transfer j of this wave fills 16-byte group (wave + WAVES * j) * 64 + lane of the patch image; its
    // source offset inside the chunk's 8 input planes (or "nothing to fetch": zero word) depends on the tile only, so it is
    // computed once per tile, not per chunk
    constexpr int PJ = (G::PINSTR + G::WAVES - 1) / G::WAVES, WJ = (G::WINSTR + G::WAVES - 1) / G::WAVES;
    struct Plan {
        const float *src[PJ];  // this lane's source of patch transfer j for the NEXT chunk to issue (the zero word: nothing to fetch)
        size_t step[PJ];       // floats to advance per chunk (0 for the zero word)
        const uint4 *wsrc;     // this lane's source inside the next weight chunk
    };
    auto make_plan = [&](const Tile &t) {
        Plan pl;
        const float *xin = a.x + (size_t)t.n * a.Cin * plane;
#pragma unroll
        for (int j = 0; j < PJ; ++j) {
            const int ins = wave + G::WAVES * j;
            const int e = (ins * 64 + lane) * 4;
            const int c = e / G::CS, rem = e - c * G::CS, r = rem / G::STR, xx = rem - r * G::STR;
            const int gy = t.y0 - 1 + r, gx = t.x0 - 4 + xx;           // gx % 4 == 0: the group is inside the row or outside
            const bool ok = c < kKC && r < G::PH && xx < G::PW && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
            pl.src[j] = ok ? xin + (size_t)c * plane + (size_t)gy * a.W + gx : a.zero;
            pl.step[j] = ok ? (size_t)kKC * plane : 0;
        }
        pl.wsrc = a.wt + (size_t)t.cg * chunks * G::WQ + lane;
        return pl;
    };
    // transfer k (patch transfers first, then weight transfers) of chunk `step` of this workgroup (S slot step % 2, W slot
    // step % 3); every lane of every transfer is active (the staging image is padded to whole 1 KiB transfers), the LDS
    // addresses are scalar
    constexpr int NPART = PJ + WJ;
    auto issue_part = [&](const Plan &pl, int step, int k) {
        float *dst = lds + G::S_OFF + (step & 1) * G::PLDS, *wdst = lds + G::W_OFF + (step % 3) * (G::WQ * 4);
        if (k < PJ) {
            const int ins = wave + G::WAVES * k;
            if (ins < G::PINSTR) __builtin_amdgcn_global_load_lds((gptr_t)pl.src[k < PJ ? k : 0], (lptr_t)(dst + ins * 256), 16, 0, 0);
        } else {
            const int ins = wave + G::WAVES * (k - PJ);
            if (ins < G::WINSTR) __builtin_amdgcn_global_load_lds((gptr_t)(pl.wsrc + ins * 64), (lptr_t)(wdst + ins * 256), 16, 0, 0);
        }
    };
    auto advance = [&](Plan &pl) {                 // to the tile's next chunk
#pragma unroll
        for (int j = 0; j < PJ; ++j) pl.src[j] += pl.step[j];
        pl.wsrc += G::WQ;
    };

    const int g = blockIdx.x, n_wg = gridDim.x;
    const int my_tiles = (a.total_tiles - g + n_wg - 1) / n_wg;        // tiles g, g + n_wg, ...
    if (my_tiles <= 0) return;
#ifdef EMU_TRACE
    if (tid == 0) a.trace[2 * 16 * 64 * 5 + 2 * g] = wall_clock64();   // 100 MHz wall clock: start / end of every workgroup
#endif
    const int n_local = my_tiles * chunks;

    // Chunk-level software pipeline (one barrier per chunk).  In iteration L a wavefront
    //   * issues the LDS-DMA of chunk L + 2          (staging slot L % 2, weight slot (L + 2) % 3),
    //   * splits the staged fp32 patch of chunk L + 1 into bf16 operands        (split-patch slot (L + 1) % 2),
    //   * runs the MFMA steps of chunk L             (split-patch slot L % 2, weight slot L % 3),
    // all in one instruction stream, so the matrix pipe works while the VALU splits and the DMA lands.  The barrier at the top of
    // iteration L + 1 (with vmcnt / lgkmcnt 0) closes all three: every slot written in iteration L had its last reader in
    // iteration L - 1.
    int it = 0, ic = 0, li = 0;                    // issue cursor: tile ordinal, chunk, chunk number of this workgroup
    Tile itile = decode(g);
    Plan iplan = make_plan(itile);
    // take the next chunk to issue off the cursor: its plan and chunk number are returned, the cursor moves on
    auto take_next = [&](Plan &pl, int &step) {
        if (it >= my_tiles) return false;
        pl = iplan;
        step = li++;
        if (++ic == chunks) {
            ic = 0;
            if (++it < my_tiles) {
                itile = decode(g + it * n_wg);
                iplan = make_plan(itile);
            }
        } else {
            advance(iplan);
        }
        return true;
    };
    // the pixel slot this lane splits (slots beyond the patch: none)
    const bool cv_on = tid < G::PIX || G::PIX > G::THREADS;
    auto cv_load = [&](int step, int i, float (&v)[8]) {
        const float *st = lds + G::S_OFF + (step & 1) * G::PLDS;
        const int y = i / G::PW, xq = i - y * G::PW;
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = st[c * G::CS + y * G::STR + xq];
    };
    auto cv_store = [&](int step, int i, const bf16x8 (&o)[TERMS]) {
        uint4 *bt = reinterpret_cast<uint4 *>(lds + G::B_OFF + (step & 1) * G::BSZ);
#pragma unroll
        for (int t = 0; t < TERMS; ++t) bt[t * G::PIX + i] = __builtin_bit_cast(uint4, o[t]);
    };
    auto convert_all = [&](int step) {             // whole split pass (prologue, and patches larger than the workgroup)
        for (int i = tid; i < G::PIX; i += G::THREADS) {
            float v[8];
            bf16x8 o[TERMS];
            cv_load(step, i, v);
            split_pixel<TERMS>(v, o);
            cv_store(step, i, o);
        }
    };

    for (int k = 0; k < 2; ++k) {
        Plan pl;
        int step;
        if (take_next(pl, step)) {
#pragma unroll
            for (int q = 0; q < NPART; ++q) issue_part(pl, step, q);
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    convert_all(0);
    if (wave >= G::WAVES / 2) __builtin_amdgcn_s_setprio(1);           // the later-dispatched half loses every arbitration otherwise
    int L = 0;
    Tile cur = decode(g);
    for (int ti = 0; ti < my_tiles; ++ti) {
        const int gy = cur.y0 + py, gx = cur.x0 + px;
        const bool live = gy < a.H && gx < a.W;
        const size_t obase = ((size_t)cur.n * a.Cout + cur.cg * kCoutTile + cb + 4 * half) * plane + (live ? (size_t)gy * a.W + gx : 0);
        const float *bias = a.bias + cur.cg * kCoutTile + cb + 4 * half;
        floatx16 acc[G::NCO];
        float res[16 * G::NCO];                    // residual, fetched at the start of the tile and added in the epilogue
#pragma unroll
        for (int q = 0; q < 16 * G::NCO; ++q) {
            const int c = (q / 16) * 32 + 8 * ((q % 16) / 4) + (q % 4);
            acc[q / 16][q % 16] = bias[c];
            res[q] = a.residual ? a.residual[obase + (size_t)c * plane] : 0.f;
        }
        for (int chunk = 0; chunk < chunks; ++chunk, ++L) {
            EMU_STAMP(0);
            __builtin_amdgcn_s_waitcnt(0);
            EMU_STAMP(1);
            __syncthreads();
            EMU_STAMP(2);
            Plan dpl;
            int dstep = 0;
            // LDS-DMA of chunk L + 2, all transfers up front (spread between the MFMA steps they were measured 10 % slower: each
            // transfer stalls the wavefront's MFMA stream)
            if (take_next(dpl, dstep)) {
#pragma unroll
                for (int q = 0; q < NPART; ++q) issue_part(dpl, dstep, q);
            }
            EMU_STAMP(3);
            const bool more = L + 1 < n_local;
            float cv[8];
            const bool cv_inline = G::PIX <= G::THREADS && more && cv_on;
            if (cv_inline) cv_load(L + 1, tid, cv);
            const uint4 *bq = reinterpret_cast<const uint4 *>(lds + G::B_OFF + (L & 1) * G::BSZ);
            const uint4 *wq = reinterpret_cast<const uint4 *>(lds + G::W_OFF + (L % 3) * (G::WQ * 4)) + wlane;
            auto load_b = [&](int s, bf16x8 (&b)[TERMS]) {
#pragma unroll
                for (int t = 0; t < TERMS; ++t) {
                    uint4 v = bq[t * G::PIX + boff[s]];
                    if (s == kSteps - 1 && half) v = uint4{0, 0, 0, 0};       // the tenth tap does not exist
                    b[t] = __builtin_bit_cast(bf16x8, v);
                }
            };
            auto load_w = [&](int s, bf16x8 (&w)[G::NCO][TERMS]) {
#pragma unroll
                for (int q = 0; q < G::NCO; ++q)
#pragma unroll
                    for (int t = 0; t < TERMS; ++t) w[q][t] = __builtin_bit_cast(bf16x8, wq[((s * TERMS + t) * 2) * kCoutTile + q * 32]);
            };
            bf16x8 bc[TERMS], wc[G::NCO][TERMS], cvo[TERMS];
            load_b(0, bc);
            load_w(0, wc);
#pragma unroll
            for (int s = 0; s < kSteps; ++s) {
                bf16x8 wn[G::NCO][TERMS], bn[TERMS];
                constexpr int NT = TERMS == 3 ? 6 : 3;
                constexpr int wi[6] = {0, 1, TERMS == 3 ? 2 : 0, 0, 1, 0};                       // weight term of product i
                constexpr int bi[6] = {TERMS == 3 ? 2 : 1, TERMS == 3 ? 1 : 0, 0, 1, 0, 0};      // pixel term of product i
                if (s + 1 < kSteps) {                  // operands of the next step are in flight while this step's MFMAs issue
                    load_b(s + 1, bn);
                    load_w(s + 1, wn);
                }
#pragma unroll
                for (int i = 0; i < NT; ++i)
#pragma unroll
                    for (int q = 0; q < G::NCO; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wc[q][wi[i]], bc[bi[i]], acc[q], 0, 0, 0);
                if (s == 1 && cv_inline) split_pixel<TERMS>(cv, cvo);          // VALU work of the split beside the MFMAs
                if (s == 3 && cv_inline) cv_store(L + 1, tid, cvo);
                if (s + 1 < kSteps) {
#pragma unroll
                    for (int t = 0; t < TERMS; ++t) {
                        bc[t] = bn[t];
#pragma unroll
                        for (int q = 0; q < G::NCO; ++q) wc[q][t] = wn[q][t];
                    }
                }
            }
            if (G::PIX > G::THREADS && more) convert_all(L + 1);
            EMU_STAMP(4);
        }
        if (live) {
#pragma unroll
            for (int q = 0; q < 16 * G::NCO; ++q) {
                const int c = (q / 16) * 32 + 8 * ((q % 16) / 4) + (q % 4);
                const float v = acc[q / 16][q % 16] + res[q];
                a.y[obase + (size_t)c * plane] = a.relu ? fmaxf(v, 0.f) : v;
            }
        }
        if (ti + 1 < my_tiles) cur = decode(g + (ti + 1) * n_wg);
    }
#ifdef EMU_TRACE
    if (tid == 0) a.trace[2 * 16 * 64 * 5 + 2 * g + 1] = wall_clock64();
#endif
}

template <int BH, int BW, int NPB, int TERMS>
int launch(const EmuArgs &a0, hipStream_t s) {
    using G = Geo<BH, BW, NPB, TERMS>;
    static int resident = 0, cus = 0;
    if (!resident) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) prop.multiProcessorCount = 256;
        cus = prop.multiProcessorCount;
        const int rc = coalign::hip_call(hipFuncSetAttribute(reinterpret_cast<const void *>(conv3x3_emu_kernel<BH, BW, NPB, TERMS>),
                                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES));
        if (rc != COALIGN_OK) {                    // geometry does not fit this device's LDS: report, leave no sticky error behind
            (void)hipGetLastError();
            return rc;
        }
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv3x3_emu_kernel<BH, BW, NPB, TERMS>, G::THREADS, G::LDS_BYTES) != hipSuccess || n < 1) n = 1;
        resident = n;
    }
    EmuArgs a = a0;
    a.tiles_x = (a.W + G::TW - 1) / G::TW;
    a.tiles_per_img = a.tiles_x * ((a.H + G::TH - 1) / G::TH);
    a.total_tiles = a.tiles_per_img * (a.Cout / kCoutTile) * a.N;
    const int grid = a.total_tiles < cus * resident ? a.total_tiles : cus * resident;
    hipLaunchKernelGGL((conv3x3_emu_kernel<BH, BW, NPB, TERMS>), dim3(grid), dim3(G::THREADS), G::LDS_BYTES, s, a);
    return COALIGN_OK;
}

template <int TERMS>
int dispatch(const EmuArgs &a, hipStream_t s) {
    static const int force = getenv("COALIGN_EMU_GEO") ? atoi(getenv("COALIGN_EMU_GEO")) : -1;      // experiments only
    // measured on the backbone shapes (tools/bench_conv_emu_geo.py): 8 row segments of 32 pixels per workgroup win or tie on every
    // map size -- the weight image is shared by 8 wavefronts and both accumulator tiles amortise the operand split; the 3-way split
    // with long K prefers 4 segments (its 53 KB buffers leave room for two such workgroups per CU)
    // measured (tools/bench_conv_emu_geo.py): 12 row segments per workgroup (3 wavefronts per SIMD) on the large maps when the
    // LDS holds them (2-way split), 8 otherwise
    int geo = (TERMS == 2 && a.H >= 64) ? 5 : 0;
    if (force >= 0) geo = force;
    switch (geo) {
        case 0: return launch<1, 32, 8, TERMS>(a, s);
        case 1: return launch<1, 32, 4, TERMS>(a, s);
        case 2: return launch<2, 16, 4, TERMS>(a, s);
        case 3: return launch<2, 16, 2, TERMS>(a, s);
        case 5: return launch<1, 32, 12, TERMS>(a, s);
        default: return launch<1, 32, 2, TERMS>(a, s);
    }
}

}  // namespace

#ifdef EMU_TRACE
static long long *g_emu_trace = nullptr;
extern "C" void coalign_conv3x3_emu_set_trace(long long *p) { g_emu_trace = p; }
#endif

extern "C" size_t coalign_conv3x3_emu_weight_bytes(int Cin, int Cout, int terms) {
    if (Cin < 1 || Cout < 1 || Cin % kKC || Cout % kCoutTile || (terms != 2 && terms != 3)) return 0;
    return (size_t)(Cout / kCoutTile) * (Cin / kKC) * kSteps * terms * 2 * kCoutTile * 16 + 16;      // + one zero group
}

extern "C" int coalign_conv3x3_emu_bias_act(const float *x, const void *w_split, const float *bias, const float *residual, float *y,
                                            int N, int Cin, int Cout, int H, int W, int relu, int terms, void *stream) {
    using namespace coalign;
    if (!x || !w_split || !y || !bias) return COALIGN_ERR_NULL_POINTER;
    if (N < 0 || H < 1 || W < 1 || Cin < 1 || Cout < 1) return COALIGN_ERR_BAD_SHAPE;
    if (Cin % kKC || Cout % kCoutTile || W % 4 || (terms != 2 && terms != 3) ||
        ((reinterpret_cast<uintptr_t>(w_split) | reinterpret_cast<uintptr_t>(x)) & 15))
        return COALIGN_ERR_UNSUPPORTED;
    if ((int64_t)N * Cout * H * W > (int64_t)1 << 40) return COALIGN_ERR_UNSUPPORTED;
    if (N == 0) return COALIGN_OK;
    EmuArgs a{x, static_cast<const uint4 *>(w_split), bias, residual, nullptr, y, N, Cin, Cout, H, W, relu, 0, 0, 0};
#ifdef EMU_TRACE
    a.trace = g_emu_trace;
#endif
    // the 16 zero bytes appended to the packed weights: source of every out-of-image / padding group of the halo patch
    a.zero = reinterpret_cast<const float *>(static_cast<const char *>(w_split) + coalign_conv3x3_emu_weight_bytes(Cin, Cout, terms) - 16);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int rc = terms == 3 ? dispatch<3>(a, s) : dispatch<2>(a, s);
    return rc != COALIGN_OK ? rc : check_launch();
}
