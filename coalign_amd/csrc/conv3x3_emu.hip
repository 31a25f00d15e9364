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
    static constexpr int BUF = PLDS + WQ * 4;                                 // floats of one LDS buffer
    static constexpr size_t LDS_BYTES = 2 * (size_t)BUF * 4;
};

struct Tile {
    int n, cg, y0, x0;
};

template <int TERMS>
__device__ __forceinline__ void split(const float (&v)[8], bf16x8 (&out)[TERMS]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const __bf16 h = (__bf16)v[i];
        const float r = v[i] - (float)h;
        out[0][i] = h;
        if (TERMS == 2) {
            out[1][i] = (__bf16)r;
        } else {
            const __bf16 m = (__bf16)r;
            out[1][i] = m;
            out[TERMS - 1][i] = (__bf16)(r - (float)m);
        }
    }
}

template <int TERMS>
__device__ __forceinline__ void split_pair(float v0, float v1, bf16x8 (&out)[TERMS], int pr) {
    const __bf16 h0 = (__bf16)v0, h1 = (__bf16)v1;
    const float r0 = v0 - (float)h0, r1 = v1 - (float)h1;
    out[0][2 * pr] = h0; out[0][2 * pr + 1] = h1;
    const __bf16 m0 = (__bf16)r0, m1 = (__bf16)r1;
    out[1][2 * pr] = m0; out[1][2 * pr + 1] = m1;
    if (TERMS == 3) {
        out[TERMS - 1][2 * pr] = (__bf16)(r0 - (float)m0);
        out[TERMS - 1][2 * pr + 1] = (__bf16)(r1 - (float)m1);
    }
}

template <int BH, int BW, int NPB, int TERMS>
__global__ __launch_bounds__(64 * (NPB >= 4 ? NPB : 2 * NPB)) void conv3x3_emu_kernel(const EmuArgs a) {
    using G = Geo<BH, BW, NPB, TERMS>;
    extern __shared__ __attribute__((aligned(1024))) float lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, half = lane >> 5, p = lane & 31;
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
    int boff[kSteps];                                     // LDS offset of this lane's tap in step s (tap 9 -> tap 8, zeroed below)
#pragma unroll
    for (int s = 0; s < kSteps; ++s) {
        const int t = 2 * s + half < 9 ? 2 * s + half : 8;
        boff[s] = (py + t / 3) * G::STR + px + 3 + t % 3;
    }
    const int wlane = half * kCoutTile + cb + p;          // 16-byte group of this lane inside one (step, term) weight block

    // LDS-DMA plan of a tile: transfer j of this wave fills 16-byte group (wave + WAVES * j) * 64 + lane of the patch image; its
    // source offset inside the chunk's 8 input planes (or "nothing to fetch": zero word) depends on the tile only, so it is
    // computed once per tile, not per chunk
    constexpr int PJ = (G::PINSTR + G::WAVES - 1) / G::WAVES, WJ = (G::WINSTR + G::WAVES - 1) / G::WAVES;
    struct Plan {
        int off[PJ];           // float offset from the chunk's first plane; < 0: out of the image / padding slot
    };
    auto make_plan = [&](const Tile &t) {
        Plan pl;
#pragma unroll
        for (int j = 0; j < PJ; ++j) {
            const int ins = wave + G::WAVES * j;
            const int e = (ins * 64 + lane) * 4;
            const int c = e / G::CS, rem = e - c * G::CS, r = rem / G::STR, xx = rem - r * G::STR;
            const int gy = t.y0 - 1 + r, gx = t.x0 - 4 + xx;           // gx % 4 == 0: the group is inside the row or outside
            const bool ok = c < kKC && r < G::PH && xx < G::PW && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
            pl.off[j] = ok ? c * (int)plane + gy * a.W + gx : -1;
        }
        return pl;
    };
    auto issue = [&](const Tile &t, const Plan &pl, int chunk, int buf) {
        float *dst = lds + buf * G::BUF;
        const float *xin = a.x + ((size_t)t.n * a.Cin + (size_t)chunk * kKC) * plane;
#pragma unroll
        for (int j = 0; j < PJ; ++j) {
            const int ins = wave + G::WAVES * j;
            if (ins < G::PINSTR) {
                if (ins * 64 + lane >= G::PGROUPS) continue;               // partial last transfer: masked lanes write nothing
                const float *src = pl.off[j] >= 0 ? xin + pl.off[j] : a.zero;
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(dst + ins * 256), 16, 0, 0);
            }
        }
        const uint4 *wsrc = a.wt + ((size_t)t.cg * chunks + chunk) * G::WQ + lane;
#pragma unroll
        for (int j = 0; j < WJ; ++j) {
            const int ins = wave + G::WAVES * j;
            if (ins < G::WINSTR) __builtin_amdgcn_global_load_lds((gptr_t)(wsrc + ins * 64), (lptr_t)(dst + G::PLDS + ins * 256), 16, 0, 0);
        }
    };

    const int g = blockIdx.x, n_wg = gridDim.x;
    const int my_tiles = (a.total_tiles - g + n_wg - 1) / n_wg;        // tiles g, g + n_wg, ...
    if (my_tiles <= 0) return;
#ifdef EMU_TRACE
    if (tid == 0) a.trace[2 * 16 * 64 * 5 + 2 * g] = wall_clock64();   // 100 MHz wall clock: start / end of every workgroup
#endif
    const int n_local = my_tiles * chunks;
    int L = 0, buf = 0;
    Tile cur = decode(g);
    Plan plan = make_plan(cur);
    issue(cur, plan, 0, 0);
    if (wave >= G::WAVES / 2) __builtin_amdgcn_s_setprio(1);           // the later-dispatched half loses every arbitration otherwise
    for (int ti = 0; ti < my_tiles; ++ti) {
        const int tile = g + ti * n_wg;
        const int gy = cur.y0 + py, gx = cur.x0 + px;
        const bool live = gy < a.H && gx < a.W;
        const size_t obase = ((size_t)cur.n * a.Cout + cur.cg * kCoutTile + cb + 4 * half) * plane + (live ? (size_t)gy * a.W + gx : 0);
        const float *bias = a.bias + cur.cg * kCoutTile + cb + 4 * half;
        floatx16 acc[G::NCO];
        Tile next = cur;
        if (a.residual) {
#pragma unroll
            for (int q = 0; q < 16 * G::NCO; ++q) {
                const int c = (q / 16) * 32 + 8 * ((q % 16) / 4) + (q % 4);
                acc[q / 16][q % 16] = a.residual[obase + (size_t)c * plane] + bias[c];
            }
        } else {
#pragma unroll
            for (int q = 0; q < 16 * G::NCO; ++q) acc[q / 16][q % 16] = bias[(q / 16) * 32 + 8 * ((q % 16) / 4) + (q % 4)];
        }
        for (int chunk = 0; chunk < chunks; ++chunk, ++L) {
            EMU_STAMP(0);
            __builtin_amdgcn_s_waitcnt(0);
            __syncthreads();
            EMU_STAMP(1);
            if (chunk + 1 < chunks) {
                issue(cur, plan, chunk + 1, buf ^ 1);
            } else if (L + 1 < n_local) {                              // first chunk of this workgroup's next tile
                next = decode(tile + n_wg);
                plan = make_plan(next);
                issue(next, plan, 0, buf ^ 1);
            }
            EMU_STAMP(2);
            const float *pl = lds + buf * G::BUF;
            const uint4 *wq = reinterpret_cast<const uint4 *>(pl + G::PLDS) + wlane;
            // software pipeline over the five steps: the LDS reads and the operand split of step s + 1 are issued beside the
            // matrix instructions of step s (hipcc alone schedules read -> split -> MFMA strictly one step at a time and the
            // matrix pipe idles through every LDS round trip and split)
            auto load_b = [&](int s, float (&v)[8]) {
#pragma unroll
                for (int c = 0; c < 8; ++c) v[c] = pl[boff[s] + c * G::CS];
            };
            auto load_w = [&](int s, bf16x8 (&w)[G::NCO][TERMS]) {
#pragma unroll
                for (int q = 0; q < G::NCO; ++q)
#pragma unroll
                    for (int t = 0; t < TERMS; ++t) w[q][t] = __builtin_bit_cast(bf16x8, wq[((s * TERMS + t) * 2) * kCoutTile + q * 32]);
            };
            float vn[8];
            bf16x8 bc[TERMS], wc[G::NCO][TERMS];
            load_b(0, vn);
            load_w(0, wc);
            split<TERMS>(vn, bc);
            EMU_STAMP(3);
#pragma unroll
            for (int s = 0; s < kSteps; ++s) {
                bf16x8 wn[G::NCO][TERMS], bn[TERMS];
                constexpr int NT = TERMS == 3 ? 6 : 3;
                constexpr int wi[6] = {0, 1, TERMS == 3 ? 2 : 0, 0, 1, 0};                       // weight term of product i
                constexpr int bi[6] = {TERMS == 3 ? 2 : 1, TERMS == 3 ? 1 : 0, 0, 1, 0, 0};      // pixel term of product i
                constexpr int n_mfma = NT * G::NCO, lead = n_mfma >= 8 ? 4 : (n_mfma >= 6 ? 2 : 1), rest = n_mfma - lead;
                auto mfma = [&](int j) {                                                          // product j / NCO on accumulator j % NCO
                    const int i = j / G::NCO, q = j % G::NCO;
                    acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wc[q][wi[i]], bc[bi[i]], acc[q], 0, 0, 0);
                };
                if (s + 1 < kSteps) {
                    // exact issue order (sched_barrier fences): every LDS read of step s + 1, `lead` bare MFMAs to cover their
                    // latency, then the split of one pixel pair at a time between the remaining MFMAs
                    load_b(s + 1, vn);
                    load_w(s + 1, wn);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < lead; ++j) mfma(j);
                    __builtin_amdgcn_sched_barrier(0);
                    int done = lead;
#pragma unroll
                    for (int pr = 0; pr < 4; ++pr) {
                        if (s + 1 == kSteps - 1) {
                            vn[2 * pr] = half ? 0.f : vn[2 * pr];
                            vn[2 * pr + 1] = half ? 0.f : vn[2 * pr + 1];
                        }
                        split_pair<TERMS>(vn[2 * pr], vn[2 * pr + 1], bn, pr);
                        const int upto = lead + (pr + 1) * rest / 4;
#pragma unroll
                        for (int j = lead + pr * rest / 4; j < upto; ++j) mfma(j);
                        done = upto;
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    (void)done;
#pragma unroll
                    for (int t = 0; t < TERMS; ++t) {
                        bc[t] = bn[t];
#pragma unroll
                        for (int q = 0; q < G::NCO; ++q) wc[q][t] = wn[q][t];
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < n_mfma; ++j) mfma(j);
                }
            }
            EMU_STAMP(4);
            buf ^= 1;
        }
        if (live) {
#pragma unroll
            for (int q = 0; q < 16 * G::NCO; ++q) {
                const int c = (q / 16) * 32 + 8 * ((q % 16) / 4) + (q % 4);
                const float v = acc[q / 16][q % 16];
                a.y[obase + (size_t)c * plane] = a.relu ? fmaxf(v, 0.f) : v;
            }
        }
        cur = next;
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
        if (rc != COALIGN_OK) return rc;
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
    int geo = (TERMS == 3 && a.Cin / kKC >= 32 && a.H >= 64) ? 1 : 0;
    if (force >= 0) geo = force;
    switch (geo) {
        case 0: return launch<1, 32, 8, TERMS>(a, s);
        case 1: return launch<1, 32, 4, TERMS>(a, s);
        case 2: return launch<2, 16, 4, TERMS>(a, s);
        case 3: return launch<2, 16, 2, TERMS>(a, s);
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
