// 3x3 / stride 1 / pad 1 convolution with fused bias (+ residual) (+ ReLU) on the fp32 matrix cores, NCHW, gfx950.
//
// Reference semantics (see include/coalign_amd.h): the Conv2d(3x3, stride 1, padding 1) + BatchNorm2d (folded into the weights
// on the host) + ReLU / residual groups of the BEV backbone and the shrink header
// (opencood/models/sub_modules/resblock.py:53-69, base_bev_backbone_resnet.py:59-138, downsample_conv.py:7-50): 94 % of the
// frame's arithmetic.  MIOpen serves these shapes with a gfx9-generation VALU Winograd kernel (80-110 TFLOP/s of direct-
// convolution-equivalent work, profiles/round1) followed by a separate bias / ReLU pass; this is a direct implicit GEMM on
// v_mfma_f32_32x32x2_f32 -- exact fp32 products and accumulation, the same rounding model as an fmaf chain -- with the epilogue
// in registers.
//
// STATUS (round 1): deterministic and correct to 4e-6 of torch's conv on every tested shape; 87-117 TFLOP/s of real work, 0.87-1.01x
// MIOpen's time (Winograd + separate epilogue pass) per layer (profiles/round1/conv3x3_bench.json).  It wins clearly where the
// input-channel count is small (the 64-channel stage: 144 vs 165 us) and the detector uses it there; the other stages stay on
// MIOpen.
//
// GEMM view per image:  D[cout, pixel] = sum_{cin, tap} W[cout, cin, tap] * X[cin, pixel + tap].
//   A operand (32 x 2)  weights: 32 output channels x 2 input channels of one tap      (from LDS, [cin][tap][cout])
//   B operand (2 x 32)  input:   2 input channels x 32 pixels shifted by the tap        (from LDS, halo patch [cin][y][x])
//   D (32 x 32)         16 accumulators per lane: lane % 32 = pixel (coalesced NCHW stores), 16 output channels.
// A workgroup (4 wavefronts) owns NPB vertically stacked pixel blocks (BH x BW = 32 pixels each) and 64 output channels.  The
// input channels stream through a double-buffered LDS image in chunks of 8 with the gfx950 LDS-DMA (global_load_lds): while the
// matrix cores work on chunk c out of one buffer, chunk c + 1 lands in the other -- no staging registers, no ds_write pass, one
// barrier per chunk.  LDS-DMA writes lane-linearly (wave-uniform base + lane x size), so the LDS image IS the layout: the packed
// weights carry their bank padding in global memory, and every float of the halo patch (padding slots and out-of-image taps
// included) is fetched by the lane that owns its LDS slot, pointed at a zero word when there is nothing to fetch.
// f32 MFMA issues one instruction per 64 cycles per SIMD, so two LDS reads per MFMA is far from the LDS limit: the kernel is
// bound by the matrix pipe as long as the staging stays out of the wavefronts' way.
#include "common.h"

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int kKC = 8;               // input channels per LDS chunk
constexpr int kCoutTile = 64;        // output channels per workgroup
constexpr int kWStride = 9 * kCoutTile + 32;   // floats between input channels of the weight chunk (+32: the two k-halves of a wave hit different banks)

typedef __attribute__((address_space(1))) const void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

struct ConvArgs {
    const float *__restrict__ x, *__restrict__ wt, *__restrict__ bias, *__restrict__ residual;
    float *__restrict__ y;
    int N, Cin, Cout, H, W, relu, tiles_x, tiles_per_img, total_tiles;
    float *__restrict__ partial;      // [grid][16 * NCO][threads]: accumulators of a tile whose chunks are split over two workgroups
    int *flags;                       // [grid], zeroed per launch: flags[g] = 1 once workgroup g has published its partial tile
};

constexpr int pick_stride(int pw, int bh, int bw) {      // smallest row stride >= pw (multiple of 4) whose BH row segments tile the 32 banks
    int s = (pw + 3) / 4 * 4;
    if (bh == 1) return s;
    while (s % 32 != bw % 32) s += 4;
    return s;
}

constexpr int kSplitChunks = 32;                          // tiles with at least this many chunks are load-balanced by splitting
constexpr int kWChunk = kKC * kWStride;                   // floats of one packed weight chunk (4864 = 19 x 256)
constexpr int kWInstr = kWChunk * 4 / 1024;               // dwordx4 LDS-DMA instructions per chunk (64 lanes x 16 B each)
static_assert(kWChunk * 4 % 1024 == 0, "weight chunk must be a whole number of dwordx4 wave transfers");

// BH x BW = 32 pixels per block; the workgroup stacks NPB blocks vertically:
//   NPB = 4: wave w owns pixel block w and both 32-channel halves (two accumulator tiles),
//   NPB = 2: wave w owns pixel block w % 2 and the 32-channel half w / 2 (one tile) -- smaller work units for small maps.
template <int BH, int BW, int NPB>
struct Geo {
    static constexpr int NCO = NPB >= 4 ? 2 : 1;                              // accumulator tiles per wave
    static constexpr int WAVES = NPB >= 4 ? NPB : 2 * NPB;                    // wavefronts per workgroup
    static constexpr int THREADS = 64 * WAVES;
    static constexpr int TH = BH * NPB, TW = BW, PH = TH + 2, PW = TW + 8;  // rows cover x0 - 4 .. x0 + TW + 3: whole 16-byte groups
    static constexpr int STR = pick_stride(PW, BH, BW);
    static_assert(STR % 4 == 0, "row stride must keep the 16-byte groups row-aligned");
    static constexpr int CS_RAW = PH * STR;
    static constexpr int CS = CS_RAW;                                         // channel stride (multiple of 4)
    static constexpr int PGROUPS = kKC * CS / 4;                              // 16-byte groups of the patch image
    static constexpr int PINSTR = (PGROUPS + 63) / 64;                        // dwordx4 LDS-DMA instructions (the last may be partial)
    static constexpr int PLDS = PGROUPS * 4;                                  // floats of the patch image
    static constexpr int BUF = PLDS + kWChunk;                                // floats of one LDS buffer (patch | weights)
};

struct Tile {
    int n, cg, y0, x0;
};

template <int BH, int BW, int NPB, bool SPLIT>
__global__ __launch_bounds__(64 * (NPB >= 4 ? NPB : 2 * NPB)) __attribute__((amdgpu_waves_per_eu(NPB == 8 ? 4 : 3, NPB == 8 ? 4 : 3)))
void conv3x3_kernel(const ConvArgs a) {
    using G = Geo<BH, BW, NPB>;
    __shared__ __attribute__((aligned(1024))) float lds[2 * G::BUF];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, half = lane >> 5, p = lane & 31;   // wave: scalar
    const size_t plane = (size_t)a.H * a.W;
    const int groups = a.Cout / kCoutTile, chunks = a.Cin / kKC;
    const float *zero = a.wt + 9 * kCoutTile;            // first padding word of the packed weights: always 0
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

    // this lane's pixel inside the tile, channel half, LDS read bases
    const int pb = wave % NPB, cb = (G::NCO == 2 ? 0 : wave / NPB) * 32;
    const int py = pb * BH + p / BW, px = p % BW;
    const int pbase = half * G::CS + py * G::STR + px + 3;      // tap (dy, dx) of pixel (py, px) sits at row py + dy, column px + 3 + dx
    const int wbase = G::PLDS + half * kWStride + cb + p;

    // LDS-DMA of one chunk into buffer `buf`: wave w issues transfers w, w + WAVES, ...  The source of every lane's 16-byte group
    // of the halo patch (or the zero word: padding slot / outside the image) depends on the tile only: it is computed once per
    // tile (`Plan`), a chunk adds its plane offset.  `wave` is scalar, so the LDS destinations are scalar too.
    constexpr int PJ = (G::PINSTR + G::WAVES - 1) / G::WAVES, WJ = (kWInstr + G::WAVES - 1) / G::WAVES;
    struct Plan {
        int off[PJ];           // float offset from the chunk's first input plane; < 0: nothing to fetch
        int n, cg;
    };
    auto make_plan = [&](const Tile &t) {
        Plan pl;
        pl.n = t.n;
        pl.cg = t.cg;
#pragma unroll
        for (int j = 0; j < PJ; ++j) {
            const int ins = wave + G::WAVES * j;
            const int e = (ins * 64 + lane) * 4;                       // first float of this lane's 16-byte group
            const int c = e / G::CS, rem = e - c * G::CS, r = rem / G::STR, xx = rem - r * G::STR;
            const int gy = t.y0 - 1 + r, gx = t.x0 - 4 + xx;           // gx % 4 == 0: the group is inside the row or outside
            const bool ok = ins * 64 + lane < G::PGROUPS && c < kKC && r < G::PH && xx < G::PW && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
            pl.off[j] = ok ? c * (int)plane + gy * a.W + gx : -1;
        }
        return pl;
    };
    auto issue = [&](const Plan &pl, int chunk, int buf) {
        float *dst = lds + buf * G::BUF;
        const float *xin = a.x + ((size_t)pl.n * a.Cin + (size_t)chunk * kKC) * plane;
#pragma unroll
        for (int j = 0; j < PJ; ++j) {
            const int ins = wave + G::WAVES * j;
            if (ins < G::PINSTR) {
                if (ins * 64 + lane >= G::PGROUPS) continue;               // partial last transfer: masked lanes write nothing
                const float *src = pl.off[j] >= 0 ? xin + pl.off[j] : zero;
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(dst + ins * 256), 16, 0, 0);
            }
        }
        const float *wsrc = a.wt + ((size_t)pl.cg * chunks + chunk) * kWChunk + lane * 4;
#pragma unroll
        for (int j = 0; j < WJ; ++j) {
            const int ins = wave + G::WAVES * j;
            if (ins < kWInstr) __builtin_amdgcn_global_load_lds((gptr_t)(wsrc + ins * 256), (lptr_t)(dst + G::PLDS + ins * 256), 16, 0, 0);
        }
    };

    // Work = total_tiles x chunks (tile, chunk) steps, cut into gridDim.x equal contiguous ranges (stream-K): every persistent
    // workgroup runs the same number of MFMAs (+-1 chunk) whatever the tile count -- with whole tiles per workgroup, 715 tiles
    // on 512 resident workgroups meant two rounds at 70 % fill.  A range may start in the middle of a tile (then this
    // workgroup computes the tile's last chunks from zero accumulators and publishes them) and may end in the middle of a tile
    // (then it owns that tile: bias / residual start, its own first chunks, plus the partial the next workgroup published at the
    // very beginning of its range, long before it is needed here).  Ranges are at least one tile long, so a tile has at most two
    // contributors and the split -- hence the summation order -- is a pure function of the shape: results stay deterministic.
    // Splitting pays when a tile is long (>= 32 chunks: the 256-channel layers and the shrink header, 13-27 % faster); short tiles
    // (64 / 128 input channels) would be split almost every time and the hand-over costs more than the imbalance.
    const long long S = (long long)a.total_tiles * chunks;
    const int g = blockIdx.x, n_wg = gridDim.x;
    constexpr bool split = SPLIT;            // decided on the host from the chunk count (kSplitChunks)
    // local step L of this workgroup -> global step: a contiguous range when splitting; whole tiles g, g + n_wg, ... otherwise
    // (strided: the workgroups resident at the same time then work on neighbouring tiles and share halos / weights in L2)
    const int s0 = split ? (int)(S * g / n_wg) : 0;
    const int n_local = split ? (int)(S * (g + 1) / n_wg) - s0 : ((a.total_tiles - g + n_wg - 1) / n_wg) * chunks;
    auto global_step = [&](int L) { return split ? s0 + L : (g + (L / chunks) * n_wg) * chunks + L % chunks; };
    if (n_local <= 0) return;
    int L = 0;
    floatx16 acc[G::NCO];
    int buf = 0;
    int plan_tile = global_step(0) / chunks;
    Plan plan = make_plan(decode(plan_tile));                         // plan of the tile the next issue belongs to
    issue(plan, global_step(0) - plan_tile * chunks, 0);
    while (L < n_local) {
        const int gs0 = global_step(L);
        const int tile = gs0 / chunks, c_begin = gs0 - tile * chunks;
        const int c_end = (n_local - L) < (chunks - c_begin) ? c_begin + (n_local - L) : chunks;
        const bool head = c_begin == 0, complete = c_end == chunks;
        const Tile cur = decode(tile);
        const int gy = cur.y0 + py, gx = cur.x0 + px;
        const bool live = gy < a.H && gx < a.W;
        // accumulator r of lane l is output channel 8 * (r / 4) + 4 * (l / 32) + r % 4 of its 32-block, pixel l % 32; dead lanes
        // read (and never write) pixel 0 of their plane
        const size_t obase = ((size_t)cur.n * a.Cout + cur.cg * kCoutTile + cb + 4 * half) * plane + (live ? (size_t)gy * a.W + gx : 0);
        const float *bias = a.bias + cur.cg * kCoutTile + cb + 4 * half;
        if (SPLIT && !head) {
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
            for (int q = 0; q < 16 * G::NCO; ++q) acc[q / 16][q % 16] = bias[(q / 16) * 32 + 8 * ((q % 16) / 4) + (q % 4)];
        }
        for (int chunk = c_begin; chunk < c_end; ++chunk, ++L) {
            // this step's image was issued into `buf` one step ago: wait for my transfers, then for everyone's (which also means
            // everyone has finished reading the other buffer), then start the next step's transfer into that other buffer
            __builtin_amdgcn_s_waitcnt(0);
            __syncthreads();
            if (L + 1 < n_local) {
                const int ns = global_step(L + 1), nt = ns / chunks;
                if (nt != plan_tile) {
                    plan_tile = nt;
                    plan = make_plan(decode(nt));
                }
                issue(plan, ns - nt * chunks, buf ^ 1);
            }
            const float *pl = lds + buf * G::BUF;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int dy = tap / 3, dx = tap % 3;
#pragma unroll
                for (int kp = 0; kp < kKC / 2; ++kp) {
                    const float b = pl[pbase + 2 * kp * G::CS + dy * G::STR + dx];
#pragma unroll
                    for (int q = 0; q < G::NCO; ++q) {
                        const float w = pl[wbase + 2 * kp * kWStride + tap * kCoutTile + q * 32];
                        acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(w, b, acc[q], 0, 0, 0);
                    }
                }
            }
            buf ^= 1;
        }
        // The hand-over of a split tile uses agent-scope *write-through* stores / L2-bypassing loads (relaxed atomics) and no
        // fences: an agent-scope release / acquire fence writes back and invalidates the whole L2 of the XCD -- with hundreds of
        // workgroups doing that, the weights and patches of everybody else kept being evicted (measured: 141 -> 382 us).
        // Ordering without fences: write-through stores -> s_waitcnt vmcnt(0) -> workgroup barrier -> flag store on the producer side;
        // flag load (L2 bypass) -> workgroup barrier -> L2-bypassing loads on the consumer side.  The step-by-step argument stands next to
        // the same code in conv3x3_emu.hip; tests/test_round3_gpu.py::test_stream_k_handover_stress[fp32] guards this kernel.
        // (slot layout [wave][q][lane]: one base pointer per 16 values + immediate offsets, so the addresses cost four registers)
        if (SPLIT && !head) {              // contributor: publish the partial sums of the tile's last chunks (slot g)
            float *slot = a.partial + (size_t)g * (16 * G::NCO * G::THREADS) + (size_t)wave * (16 * G::NCO * 64) + lane;
#pragma unroll
            for (int q = 0; q < 16 * G::NCO; ++q)
                __hip_atomic_store(slot + (q / 16) * 1024 + (q % 16) * 64, acc[q / 16][q % 16], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_s_waitcnt(0);     // my write-throughs are acknowledged ...
            __syncthreads();                   // ... and so are everyone's
            if (tid == 0) __hip_atomic_store(a.flags + g, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            continue;
        }
        if (SPLIT && !complete) {          // owner of a split tile: add what workgroup g + 1 published
            if (tid == 0)
                while (__hip_atomic_load(a.flags + g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) __builtin_amdgcn_s_sleep(2);
            __syncthreads();
            const float *slot = a.partial + (size_t)(g + 1) * (16 * G::NCO * G::THREADS) + (size_t)wave * (16 * G::NCO * 64) + lane;
#pragma unroll
            for (int q = 0; q < 16 * G::NCO; ++q)
                acc[q / 16][q % 16] += __hip_atomic_load(slot + (q / 16) * 1024 + (q % 16) * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (live) {
#pragma unroll
            for (int q = 0; q < 16 * G::NCO; ++q) {
                const int c = (q / 16) * 32 + 8 * ((q % 16) / 4) + (q % 4);
                const float v = acc[q / 16][q % 16];
                a.y[obase + (size_t)c * plane] = a.relu ? fmaxf(v, 0.f) : v;
            }
        }
    }
}

template <int BH, int BW, int NPB>
int plan(const ConvArgs &a0, ConvArgs *out, size_t *ws_bytes) {          // grid size; fills the tiling fields and the workspace size
    using G = Geo<BH, BW, NPB>;
    static int resident = 0;                    // workgroups one CU holds at once (occupancy query, once per shape family)
    static int cus = 0;
    if (!resident) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) prop.multiProcessorCount = 256;
        cus = prop.multiProcessorCount;
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv3x3_kernel<BH, BW, NPB, true>, G::THREADS, 0) != hipSuccess || n < 1) n = 1;
        resident = n;
    }
    ConvArgs a = a0;
    a.tiles_x = (a.W + G::TW - 1) / G::TW;
    a.tiles_per_img = a.tiles_x * ((a.H + G::TH - 1) / G::TH);
    a.total_tiles = a.tiles_per_img * (a.Cout / kCoutTile) * a.N;
    const int grid = a.total_tiles < cus * resident ? a.total_tiles : cus * resident;     // every range >= one tile; all resident
    if (out) *out = a;
    if (ws_bytes) *ws_bytes = coalign::align_up((size_t)(grid + 1) * sizeof(int), 256) + (size_t)(grid + 1) * 16 * G::NCO * G::THREADS * sizeof(float);
    return grid;
}

template <int BH, int BW, int NPB>
int launch(const ConvArgs &a0, void *workspace, size_t workspace_bytes, hipStream_t s) {
    using G = Geo<BH, BW, NPB>;
    ConvArgs a;
    size_t need = 0;
    const int grid = plan<BH, BW, NPB>(a0, &a, &need);
    if (!workspace) return COALIGN_ERR_NULL_POINTER;
    if (workspace_bytes < need) return COALIGN_ERR_WORKSPACE;
    const size_t flag_bytes = coalign::align_up((size_t)(grid + 1) * sizeof(int), 256);
    a.flags = static_cast<int *>(workspace);
    a.partial = reinterpret_cast<float *>(static_cast<char *>(workspace) + flag_bytes);
    const int rc = coalign::fill_words(workspace, flag_bytes / 4, 0u, s);
    if (rc != COALIGN_OK) return rc;
    if (a.Cin / kKC >= kSplitChunks) hipLaunchKernelGGL((conv3x3_kernel<BH, BW, NPB, true>), dim3(grid), dim3(G::THREADS), 0, s, a);
    else hipLaunchKernelGGL((conv3x3_kernel<BH, BW, NPB, false>), dim3(grid), dim3(G::THREADS), 0, s, a);
    return COALIGN_OK;
}

}  // namespace

static int check_conv_args(const float *x, const float *w_packed, int N, int Cin, int Cout, int H, int W) {
    if (N < 0 || H < 1 || W < 1 || Cin < 1 || Cout < 1) return COALIGN_ERR_BAD_SHAPE;
    if (Cin % kKC || Cout % kCoutTile || W % 4 || ((reinterpret_cast<uintptr_t>(w_packed) | reinterpret_cast<uintptr_t>(x)) & 15)) return COALIGN_ERR_UNSUPPORTED;
    if ((int64_t)N * Cout * H * W > (int64_t)1 << 40 || (int64_t)N * Cout * H * W * (Cin / kKC) > (int64_t)1 << 40) return COALIGN_ERR_UNSUPPORTED;
    return COALIGN_OK;
}

extern "C" size_t coalign_conv3x3_workspace_bytes(int N, int Cin, int Cout, int H, int W) {
    if (check_conv_args(nullptr, nullptr, N, Cin, Cout, H, W) != COALIGN_OK || N == 0) return 0;
    const ConvArgs a{nullptr, nullptr, nullptr, nullptr, nullptr, N, Cin, Cout, H, W, 0, 0, 0, 0, nullptr, nullptr};
    size_t need = 0;
    if ((W % 32 == 0 || W >= 256) && H >= 64) plan<1, 32, 8>(a, nullptr, &need);
    else if (W % 32 == 0 || W >= 256) plan<1, 32, 4>(a, nullptr, &need);
    else if (W % 16 == 0) plan<2, 16, 2>(a, nullptr, &need);
    else plan<1, 32, 2>(a, nullptr, &need);
    return need;
}

extern "C" int coalign_conv3x3_bias_act(const float *x, const float *w_packed, const float *bias, const float *residual, float *y,
                                        int N, int Cin, int Cout, int H, int W, int relu, void *workspace, size_t workspace_bytes,
                                        void *stream) {
    using namespace coalign;
    if (!x || !w_packed || !y || !bias) return COALIGN_ERR_NULL_POINTER;
    int rc = check_conv_args(x, w_packed, N, Cin, Cout, H, W);
    if (rc != COALIGN_OK) return rc;
    if (N == 0) return COALIGN_OK;
    const ConvArgs a{x, w_packed, bias, residual, y, N, Cin, Cout, H, W, relu, 0, 0, 0, nullptr, nullptr};
    hipStream_t s = static_cast<hipStream_t>(stream);
    // pixel-block shape by map size: 32-pixel row segments on wide maps, 2 x 16 blocks on the middle ones
    if ((W % 32 == 0 || W >= 256) && H >= 64) rc = launch<1, 32, 8>(a, workspace, workspace_bytes, s);       // 8 wavefronts share one weight image
    else if (W % 32 == 0 || W >= 256) rc = launch<1, 32, 4>(a, workspace, workspace_bytes, s);
    else if (W % 16 == 0) rc = launch<2, 16, 2>(a, workspace, workspace_bytes, s);
    else rc = launch<1, 32, 2>(a, workspace, workspace_bytes, s);
    return rc != COALIGN_OK ? rc : check_launch();
}
