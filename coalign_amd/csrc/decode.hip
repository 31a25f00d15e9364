// Anchor decode: sigmoid + threshold + box decode + direction fix + corners + projection + sanity filters,
// order-preserving compaction without atomics (deterministic), gfx950.
//
// Reference semantics (see include/coalign_amd.h): VoxelPostprocessor.post_process
// (opencood/data_utils/post_processor/voxel_postprocessor.py:275-377) and the box_utils helpers it calls.
// The reference runs ~40 tiny PyTorch ops plus boolean-mask compactions here; this is two launches:
//   count_kernel  one thread per anchor, sigmoid(cls) > thr, per-block popcounts
//   emit_kernel   same predicate, block offset = sum of the preceding block counts, wave-ballot prefix inside
//                 the block -> the candidate list keeps the reference's flat (h, w, anchor) order; only the
//                 passing anchors (a few %) run the full decode.
// The file is compiled with -ffp-contract=off: every a*b+c below rounds twice, like the op-by-op float32
// evaluation of the reference.
#include "common.h"

namespace {

constexpr int kBlock = 256;

struct DecodeArgs {
    const float *cls, *reg, *dir, *anchors, *T;
    int A, HW, nb, total, hwl, capacity;
    float thr, dir_offset;
    const int *count_in;
    int *count_out, *cand_index;
    float *cand_score, *cand_box7, *cand_corners;
    uint8_t *cand_keep;
    uint32_t *status;
    int *block_counts;
    uint32_t *clear;          // round 6: the frame's counters / status words, zeroed by the FIRST decode launch of a frame (no fill launch of their own)
    int n_clear;
};

// Correctly rounded float32 sigmoid (evaluated in float64): the value every faithful float32 implementation
// (the reference's vectorised CPU one included) is within an ulp or two of.
__device__ __forceinline__ float sigmoid_f32(float x) { return (float)(1.0 / (1.0 + exp(-(double)x))); }

__device__ __forceinline__ bool passes(const DecodeArgs &a, int i, float &prob) {
    const int hw = i / a.A, an = i - hw * a.A;
    prob = sigmoid_f32(a.cls[(size_t)an * a.HW + hw]);
    return prob > a.thr;
}

__device__ __forceinline__ float limit_period(float v, float offset, float period) {
    return v - floorf(v / period + offset) * period;  // common_utils.py:70-79
}

__device__ void decode_and_store(const DecodeArgs &a, int i, float prob, int pos) {
    const int hw = i / a.A, an = i - hw * a.A;
    float d[7], r[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        d[k] = a.reg[(size_t)(an * 7 + k) * a.HW + hw];
        r[k] = a.anchors[(size_t)i * 7 + k];
    }
    // delta_to_boxes3d, voxel_postprocessor.py:405-450
    const float diag = sqrtf(r[4] * r[4] + r[5] * r[5]);
    float b[7];
    b[0] = d[0] * diag + r[0];
    b[1] = d[1] * diag + r[1];
    b[2] = d[2] * r[3] + r[2];
    b[3] = expf(d[3]) * r[3];
    b[4] = expf(d[4]) * r[4];
    b[5] = expf(d[5]) * r[5];
    b[6] = d[6] + r[6];
    if (a.dir) {  // direction-bin correction, voxel_postprocessor.py:317-333
        int label = 0;
        float best = a.dir[(size_t)(an * a.nb) * a.HW + hw];
        for (int q = 1; q < a.nb; ++q) {
            const float v = a.dir[(size_t)(an * a.nb + q) * a.HW + hw];
            if (v > best) { best = v; label = q; }
        }
        const float period = (float)(2.0 * M_PI / a.nb);
        const float rot = limit_period(b[6] - a.dir_offset, 0.f, period);
        b[6] = (rot + a.dir_offset) + period * (float)label;
        b[6] = limit_period(b[6], 0.5f, (float)(2.0 * M_PI));
    }
    // boxes_to_corners_3d, box_utils.py:152-204: (l, w, h) scaled +-1/2 template, rotate about z, add centre
    const float L = a.hwl ? b[5] : b[3], Wd = b[4], Hh = a.hwl ? b[3] : b[5];
    const float ca = cosf(b[6]), sa = sinf(b[6]);
    const float sx[8] = {1, 1, -1, -1, 1, 1, -1, -1}, sy[8] = {-1, 1, 1, -1, -1, 1, 1, -1}, sz[8] = {-1, -1, -1, -1, 1, 1, 1, 1};
    float T[12];
    if (a.T) {
#pragma unroll
        for (int k = 0; k < 12; ++k) T[k] = a.T[k];
    }
    float xmin = INFINITY, xmax = -INFINITY, ymin = INFINITY, ymax = -INFINITY, zmin = INFINITY, zmax = -INFINITY;
    float *co = a.cand_corners + (size_t)pos * 24;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float lx = L * (sx[k] / 2), ly = Wd * (sy[k] / 2), lz = Hh * (sz[k] / 2);
        float X = (lx * ca + ly * (-sa)) + b[0];
        float Y = (lx * sa + ly * ca) + b[1];
        float Z = lz + b[2];
        if (a.T) {  // project_box3d, box_utils.py:278-316
            const float px = ((T[0] * X + T[1] * Y) + T[2] * Z) + T[3];
            const float py = ((T[4] * X + T[5] * Y) + T[6] * Z) + T[7];
            const float pz = ((T[8] * X + T[9] * Y) + T[10] * Z) + T[11];
            X = px; Y = py; Z = pz;
        }
        co[3 * k] = X; co[3 * k + 1] = Y; co[3 * k + 2] = Z;
        xmin = fminf(xmin, X); xmax = fmaxf(xmax, X);
        ymin = fminf(ymin, Y); ymax = fmaxf(ymax, Y);
        zmin = fminf(zmin, Z); zmax = fmaxf(zmax, Z);
    }
    // remove_large_pred_bbx (box_utils.py:840-869, its "z_len" is the y extent used as a truthy mask) and
    // remove_bbx_abnormal_z (:872-890)
    const float x_len = xmax - xmin, y_len = ymax - ymin;
    const bool keep = (x_len <= 6.f) && (y_len <= 6.f) && (y_len != 0.f) && (zmin >= -3.f) && (zmax <= 1.f);
    a.cand_keep[pos] = keep ? 1 : 0;
    a.cand_score[pos] = prob;
    if (a.cand_index) a.cand_index[pos] = i;
    if (a.cand_box7) {
#pragma unroll
        for (int k = 0; k < 7; ++k) a.cand_box7[(size_t)pos * 7 + k] = b[k];
    }
}

__global__ __launch_bounds__(kBlock) void count_kernel(DecodeArgs a) {
    __shared__ int wave_cnt[kBlock / 64];
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (blockIdx.x == 0 && (int)threadIdx.x < a.n_clear) a.clear[threadIdx.x] = 0u;      // (nothing of this launch reads them; emit_kernel, the next launch, does)
    float p;
    const bool ok = i < a.total && passes(a, i, p);
    const unsigned long long m = __ballot(ok);
    if ((threadIdx.x & 63) == 0) wave_cnt[threadIdx.x >> 6] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        int s = 0;
        for (int w = 0; w < kBlock / 64; ++w) s += wave_cnt[w];
        a.block_counts[blockIdx.x] = s;
    }
}

__global__ __launch_bounds__(kBlock) void emit_kernel(DecodeArgs a) {
    __shared__ int wave_cnt[kBlock / 64];
    __shared__ int red[kBlock / 64];
    __shared__ int s_before, s_all;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // candidates emitted by earlier blocks (and, in the last block, by everybody: the new total)
    int before = 0, all = 0;
    for (int bq = threadIdx.x; bq < (int)gridDim.x; bq += kBlock) {
        const int c = a.block_counts[bq];
        all += c;
        if (bq < (int)blockIdx.x) before += c;
    }
    for (int o = 32; o > 0; o >>= 1) { before += __shfl_xor(before, o); all += __shfl_xor(all, o); }
    if (lane == 0) { wave_cnt[wv] = before; red[wv] = all; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int sb = 0, sa = 0;
        for (int w = 0; w < kBlock / 64; ++w) { sb += wave_cnt[w]; sa += red[w]; }
        s_before = sb; s_all = sa;
    }
    __syncthreads();
    const int start = a.count_in ? *a.count_in : 0;
    const int base = start + s_before;
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
        const long tot = (long)start + s_all;
        *a.count_out = (int)(tot < a.capacity ? tot : a.capacity);
        if (tot > a.capacity && a.status) atomicOr(a.status, COALIGN_FLAG_CANDIDATE_OVERFLOW);
    }
    __syncthreads();

    const int i = blockIdx.x * kBlock + threadIdx.x;
    float p = 0.f;
    const bool ok = i < a.total && passes(a, i, p);
    const unsigned long long m = __ballot(ok);
    if (lane == 0) wave_cnt[wv] = __popcll(m);
    __syncthreads();
    int pos = base + __popcll(m & ((1ull << lane) - 1ull));
    for (int w = 0; w < wv; ++w) pos += wave_cnt[w];
    if (ok && pos < a.capacity) decode_and_store(a, i, p, pos);
}

}  // namespace

extern "C" {

size_t coalign_anchor_decode_workspace_bytes(int A, int H, int W) {
    if (A <= 0 || H <= 0 || W <= 0) return 0;
    const size_t total = (size_t)A * H * W;
    return coalign::align_up(((total + kBlock - 1) / kBlock) * sizeof(int), 256);
}

static int anchor_decode_impl(const float *cls, const float *reg, const float *dir, const float *anchors, int A, int H, int W,
                              int num_bins, float score_thr, float dir_offset, int order_hwl, const float *transform,
                              int capacity, const int32_t *count_in, int32_t *count_out, int32_t *cand_index,
                              float *cand_score, float *cand_box7, float *cand_corners, uint8_t *cand_keep, uint32_t *status,
                              void *workspace, size_t workspace_bytes, uint32_t *clear_words, int n_clear, void *stream_);

int coalign_anchor_decode(const float *cls, const float *reg, const float *dir, const float *anchors, int A, int H, int W,
                          int num_bins, float score_thr, float dir_offset, int order_hwl, const float *transform,
                          int capacity, const int32_t *count_in, int32_t *count_out, int32_t *cand_index,
                          float *cand_score, float *cand_box7, float *cand_corners, uint8_t *cand_keep, uint32_t *status,
                          void *workspace, size_t workspace_bytes, void *stream_) {
    return anchor_decode_impl(cls, reg, dir, anchors, A, H, W, num_bins, score_thr, dir_offset, order_hwl, transform, capacity, count_in, count_out, cand_index, cand_score, cand_box7,
                              cand_corners, cand_keep, status, workspace, workspace_bytes, nullptr, 0, stream_);
}

// Round 6: the FIRST decode call of a frame also zeroes the frame's counter / status words (n_clear <= 256 32-bit words at clear_words: count_in, count_out and status
// normally point into them) -- the fill launch in front of the post-processing chain is gone.
int coalign_anchor_decode_first(const float *cls, const float *reg, const float *dir, const float *anchors, int A, int H, int W,
                                int num_bins, float score_thr, float dir_offset, int order_hwl, const float *transform,
                                int capacity, const int32_t *count_in, int32_t *count_out, int32_t *cand_index,
                                float *cand_score, float *cand_box7, float *cand_corners, uint8_t *cand_keep, uint32_t *status,
                                void *workspace, size_t workspace_bytes, uint32_t *clear_words, int n_clear, void *stream_) {
    if (!clear_words) return COALIGN_ERR_NULL_POINTER;
    if (n_clear < 1 || n_clear > kBlock) return COALIGN_ERR_BAD_SHAPE;
    return anchor_decode_impl(cls, reg, dir, anchors, A, H, W, num_bins, score_thr, dir_offset, order_hwl, transform, capacity, count_in, count_out, cand_index, cand_score, cand_box7,
                              cand_corners, cand_keep, status, workspace, workspace_bytes, clear_words, n_clear, stream_);
}

static int anchor_decode_impl(const float *cls, const float *reg, const float *dir, const float *anchors, int A, int H, int W,
                              int num_bins, float score_thr, float dir_offset, int order_hwl, const float *transform,
                              int capacity, const int32_t *count_in, int32_t *count_out, int32_t *cand_index,
                              float *cand_score, float *cand_box7, float *cand_corners, uint8_t *cand_keep, uint32_t *status,
                              void *workspace, size_t workspace_bytes, uint32_t *clear_words, int n_clear, void *stream_) {
    using namespace coalign;
    hipStream_t stream = (hipStream_t)stream_;
    if (A <= 0 || H <= 0 || W <= 0 || capacity < 0 || (dir && num_bins <= 0)) return COALIGN_ERR_BAD_SHAPE;
    if ((size_t)A * H * W * 7 > (size_t)INT32_MAX) return COALIGN_ERR_BAD_SHAPE;
    if (!cls || !reg || !anchors || !count_out || !cand_score || !cand_corners || !cand_keep || !workspace)
        return COALIGN_ERR_NULL_POINTER;
    if (workspace_bytes < coalign_anchor_decode_workspace_bytes(A, H, W)) return COALIGN_ERR_WORKSPACE;
    DecodeArgs a;
    a.cls = cls; a.reg = reg; a.dir = dir; a.anchors = anchors; a.T = transform;
    a.A = A; a.HW = H * W; a.nb = num_bins; a.total = A * H * W; a.hwl = order_hwl; a.capacity = capacity;
    a.thr = score_thr; a.dir_offset = dir_offset;
    a.count_in = count_in; a.count_out = count_out; a.cand_index = cand_index; a.cand_score = cand_score;
    a.cand_box7 = cand_box7; a.cand_corners = cand_corners; a.cand_keep = cand_keep; a.status = status;
    a.block_counts = (int *)workspace;
    a.clear = clear_words;
    a.n_clear = clear_words ? n_clear : 0;
    const int blocks = (a.total + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(count_kernel, dim3(blocks), dim3(kBlock), 0, stream, a);
    int rc = check_launch();
    if (rc) return rc;
    hipLaunchKernelGGL(emit_kernel, dim3(blocks), dim3(kBlock), 0, stream, a);
    return check_launch();
}

}  // extern "C"
