// Points -> pillars voxeliser on the device, gfx950.
//
// Reference semantics (see include/coalign_amd.h): SpVoxelPreprocessor.preprocess + collate_batch
// (opencood/data_utils/pre_processor/sp_voxel_preprocessor.py:62-147) around spconv's sequential point-to-voxel loop, and
// the two point filters of opencood/utils/pcd_utils.py:41-88.  The sequential loop numbers voxels by first appearance
// and keeps each voxel's first `max_points` points *in point order*; both orders are reproduced exactly, so the output is
// bit-identical to the CPU loop and run-to-run deterministic.  Per-point global atomics are avoided (device-scope atomics
// cost ~100 us per 400 k points on this part, measured): every cell of the dense grid has exactly one owner workgroup.
//
//   split_kernel   per 1024-point block: filters, float32 cell coordinate (IEEE divide + floor, like the CPU loop) ->
//                  cell code per point, and the block's points grouped by owner (an LDS counting sort over the owners;
//                  per block a row of owner offsets).  This is the first level of a two-level bucket sort by cell.
//   owner_kernel   grid (owners, clouds); an owner holds 5120 cells (dealt round-robin, see cell_code()).
//                  It walks only its own slices of the split array -- a few thousand points -- and builds in LDS the
//                  per-cell point count and smallest point index, scans the counts into per-cell segment offsets (its
//                  slice of the bucket array starts where the lower owners' points end), writes (count, first index,
//                  segment start) of every occupied cell, walks its points again and drops each index into its cell's
//                  segment (LDS ticket; the order inside a segment is arbitrary scratch, sorted later).
//                  Finally it adds its cells' heads (a cell's smallest point index opens the voxel) to the per-1024-point-
//                  block head counts -- the only global atomics of the pipeline, one per (owner, block).
//   assign_kernel  per 1024-point block: a point is a head iff it is its cell's smallest index; head rank in point order
//                  (heads in earlier blocks + scan inside the block) = voxel number, clamped to max_voxels per cloud; writes
//                  coords (cloud, z, y, x), num_points, the voxel's (segment start, point count), and the voxel counts.
//   gather_kernel  the max_points smallest indices of each voxel's segment in ascending order: 16 lanes per voxel with
//                  DPP row rotations for the common <= 16-point voxels, a whole wavefront per entry of the big-cell list
//                  (cross-lane rank up to 64 points, candidate bound + LDS compaction beyond); float4 gather of those
//                  points, zero fill of the remaining slots.
//
// Everything is integer / index work plus copies: HBM-bound by the [M, max_points, 4] output it must write.
#include <algorithm>

#include "common.h"

namespace {

constexpr int kMaxClouds = 16;
constexpr int kBlock = 256;
constexpr int kPerThread = 4;
constexpr int kChunk = kBlock * kPerThread;   // points per workgroup in the per-point kernels
constexpr int kOwner = 1024;                  // threads of an owner workgroup
constexpr int kRange = 5 * kOwner;            // cells per owner (2 x 20 KB of LDS)
constexpr int kMaxOwners = 4096;              // LDS histogram of the split
constexpr int kMaxBlocks = 2048;              // 1024-point blocks per cloud (2 M points) the owner can index in LDS
constexpr int kSegLds = 1024;                 // per-wave LDS copy of a large voxel's segment (ints)
constexpr int kIntMax = 0x7fffffff;

struct VoxArgs {
    const float4 *pts;
    int n_clouds, max_blocks;
    int off[kMaxClouds + 1];      // first point of each cloud
    float lo[3], vs[3];
    int grid[3], ncell;
    int ranges, ncode;            // owner workgroups per cloud, ranges * kRange >= ncell cell codes per cloud
    int max_points, max_voxels, capacity;
    int flags;
    float flo[3], fhi[3];
    int *cell_of_point, *table, *blocksum, *bucket;
    int *biglist, *big_count;     // flat cellinfo indices of the cells holding > 16 points, and how many there are
    int2 *split;                  // per block: (cell code, point index in cloud) grouped by owner
    int4 *cellinfo;               // per cell code: (point count, smallest point index, segment start, voxel row or -1)
    int2 *seg;
    float4 *voxels;
    int4 *coords;
    int *num_points, *voxel_counts;
};

// Cell -> code: consecutive cells are dealt round-robin to the owner workgroups, so that the densely hit cells around the
// sensor spread evenly over all owners (measured with contiguous ownership: one owner got 40 % of the points).
// code = owner * kRange + slot; bijective on [0, ncell).
__device__ __forceinline__ int cell_code(const VoxArgs &a, int cell) {
    const int slot = cell / a.ranges;
    return (cell - slot * a.ranges) * kRange + slot;
}

__device__ __forceinline__ int code_cell(const VoxArgs &a, int code) {
    const int owner = code / kRange;
    return (code - owner * kRange) * a.ranges + owner;
}

__device__ __forceinline__ int cell_of(const VoxArgs &a, const float4 p) {
    if (a.flags & COALIGN_VOX_FILTER_EGO) {      // pcd_utils.py:69-88, closed box
        if (p.x >= -1.95f && p.x <= 2.95f && p.y >= -1.1f && p.y <= 1.1f) return -1;
    }
    if (a.flags & COALIGN_VOX_FILTER_RANGE) {    // pcd_utils.py:41-66, strict
        if (!(p.x > a.flo[0] && p.x < a.fhi[0] && p.y > a.flo[1] && p.y < a.fhi[1] && p.z > a.flo[2] && p.z < a.fhi[2])) return -1;
    }
    const float cx = floorf((p.x - a.lo[0]) / a.vs[0]);
    const float cy = floorf((p.y - a.lo[1]) / a.vs[1]);
    const float cz = floorf((p.z - a.lo[2]) / a.vs[2]);
    // NaN coordinates fail every comparison and are rejected (the CPU loop's int conversion of NaN is undefined)
    if (!(cx >= 0.0f && cx < (float)a.grid[0] && cy >= 0.0f && cy < (float)a.grid[1] && cz >= 0.0f && cz < (float)a.grid[2])) return -1;
    return cell_code(a, ((int)cz * a.grid[1] + (int)cy) * a.grid[0] + (int)cx);
}

__global__ __launch_bounds__(kBlock) void split_kernel(const VoxArgs a) {
    __shared__ int hist[kMaxOwners + 1];
    __shared__ int wave_tmp[kBlock / 64];
    const int cloud = blockIdx.y, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int begin = a.off[cloud], n = a.off[cloud + 1] - begin;
    if (tid == 0) {                   // accumulated by owner_kernel, next in the stream
        a.blocksum[cloud * a.max_blocks + blockIdx.x] = 0;
        if (blockIdx.x == 0 && blockIdx.y == 0) *a.big_count = 0;
    }
    if (blockIdx.x * kChunk >= n) return;
    const int R = a.ranges;
    for (int r = tid; r <= R; r += kBlock) hist[r] = 0;
    __syncthreads();
    int *cells = a.cell_of_point + begin;
    int code[kPerThread], ticket[kPerThread];
#pragma unroll
    for (int k = 0; k < kPerThread; ++k) {
        const int i = blockIdx.x * kChunk + k * kBlock + tid;
        code[k] = -1;
        if (i < n) {
            code[k] = cell_of(a, a.pts[begin + i]);
            cells[i] = code[k];
            if (code[k] >= 0) ticket[k] = atomicAdd(&hist[code[k] / kRange], 1);
        }
    }
    __syncthreads();
    // exclusive scan of the R owner counts (R <= 4096: up to 16 consecutive bins per thread)
    const int per = (R + kBlock - 1) / kBlock;
    int sum = 0;
    for (int k = 0; k < per; ++k) {
        const int r = tid * per + k;
        if (r < R) sum += hist[r];
    }
    int incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(incl, d);
        if (lane >= d) incl += t;
    }
    if (lane == 63) wave_tmp[wave] = incl;
    __syncthreads();
    int run = incl - sum;
    for (int w = 0; w < wave; ++w) run += wave_tmp[w];
    for (int k = 0; k < per; ++k) {
        const int r = tid * per + k;
        if (r < R) {
            const int c = hist[r];
            hist[r] = run;
            run += c;
        }
    }
    if (tid == kBlock - 1) hist[R] = run;
    __syncthreads();
    int *row = a.table + ((size_t)cloud * a.max_blocks + blockIdx.x) * (R + 1);
    for (int r = tid; r <= R; r += kBlock) row[r] = hist[r];
    int2 *out = a.split + begin + blockIdx.x * kChunk;
#pragma unroll
    for (int k = 0; k < kPerThread; ++k)
        if (code[k] >= 0) out[hist[code[k] / kRange] + ticket[k]] = make_int2(code[k], blockIdx.x * kChunk + k * kBlock + tid);
}

__global__ __launch_bounds__(kOwner) void owner_kernel(const VoxArgs a) {
    __shared__ int cnt[kRange];            // point count, later the exclusive segment offset of the cell
    __shared__ int aux[kRange];            // kIntMax - smallest point index, later the fill ticket of the cell
    __shared__ int pre[kMaxBlocks + 1];    // elements of this owner in the blocks before b
    __shared__ int from[kMaxBlocks];       // where this owner's slice starts inside block b
    __shared__ int wave_tmp[kOwner / 64], wave_big[kOwner / 64];
    __shared__ int below_total, big_base;
    const int cloud = blockIdx.y, owner = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int begin = a.off[cloud], n = a.off[cloud + 1] - begin, nb = (n + kChunk - 1) / kChunk;
    const int R = a.ranges, lo = owner * kRange;
    for (int c = tid; c < kRange; c += kOwner) { cnt[c] = 0; aux[c] = 0; }
    if (tid == 0) below_total = 0;
    // slices: block b holds this owner's points at [from[b], from[b] + size_b) of its 1024 split slots
    const int *table = a.table + (size_t)cloud * a.max_blocks * (R + 1);
    constexpr int kPerB = kMaxBlocks / kOwner;
    int size[kPerB], ssum = 0, below = 0;
#pragma unroll
    for (int k = 0; k < kPerB; ++k) {
        const int b = tid * kPerB + k;
        size[k] = 0;
        if (b < nb) {
            const int s = table[(size_t)b * (R + 1) + owner], e = table[(size_t)b * (R + 1) + owner + 1];
            from[b] = s;
            size[k] = e - s;
            below += s;
        }
        ssum += size[k];
    }
    __syncthreads();                  // below_total = 0 visible
    int incl = ssum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(incl, d);
        if (lane >= d) incl += t;
    }
    if (lane == 63) wave_tmp[wave] = incl;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) below += __shfl_xor(below, d);
    if (lane == 0) atomicAdd(&below_total, below);
    __syncthreads();
    {
        int run = incl - ssum;
        for (int w = 0; w < wave; ++w) run += wave_tmp[w];
#pragma unroll
        for (int k = 0; k < kPerB; ++k) {
            const int b = tid * kPerB + k;
            pre[b] = run;            // blocks past nb are empty: pre[nb] = total
            run += size[k];
        }
        if (tid == kOwner - 1) pre[kMaxBlocks] = run;
    }
    __syncthreads();
    const int total = pre[nb];
    const int2 *split = a.split + begin;
    auto element = [&](int e) {       // e-th point of this owner: binary search for its block
        int l = 0, h = nb;            // invariant: pre[l] <= e < pre[h]
        while (h - l > 1) {
            const int m = (l + h) >> 1;
            if (pre[m] <= e) l = m; else h = m;
        }
        return split[l * kChunk + from[l] + (e - pre[l])];
    };
    // a thread's first kKeep points stay in registers for both walks (a balanced owner has ~3 per thread): their
    // searches and loads are independent and overlap; the atomics follow in one burst
    constexpr int kKeep = 4;
    int2 mine[kKeep];
#pragma unroll
    for (int u = 0; u < kKeep; ++u) {
        const int e = tid + u * kOwner;
        mine[u] = e < total ? element(e) : make_int2(-1, 0);
    }
#pragma unroll
    for (int u = 0; u < kKeep; ++u)
        if (mine[u].x >= 0) {
            atomicAdd(&cnt[mine[u].x - lo], 1);
            atomicMax(&aux[mine[u].x - lo], kIntMax - mine[u].y);
        }
    for (int e = tid + kKeep * kOwner; e < total; e += kOwner) {
        const int2 p = element(e);
        atomicAdd(&cnt[p.x - lo], 1);
        atomicMax(&aux[p.x - lo], kIntMax - p.y);
    }
    __syncthreads();
    // exclusive scan of the counts: 5 consecutive cells per thread, wave scan, wave totals
    constexpr int kPer = kRange / kOwner;
    int local[kPer], sum = 0;
#pragma unroll
    for (int k = 0; k < kPer; ++k) { local[k] = cnt[tid * kPer + k]; sum += local[k]; }
    incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(incl, d);
        if (lane >= d) incl += t;
    }
    int nbig = 0;                     // cells of > 16 points go onto a work list: gather_kernel gives each a whole wavefront
#pragma unroll
    for (int k = 0; k < kPer; ++k) nbig += local[k] > 16 ? 1 : 0;
    int bincl = nbig;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(bincl, d);
        if (lane >= d) bincl += t;
    }
    if (lane == 63) { wave_tmp[wave] = incl; wave_big[wave] = bincl; }
    __syncthreads();
    int run = below_total + incl - sum, brun = bincl - nbig;
    for (int w = 0; w < wave; ++w) { run += wave_tmp[w]; brun += wave_big[w]; }
    if (tid == kOwner - 1) big_base = (brun + nbig) ? atomicAdd(a.big_count, brun + nbig) : 0;   // one global atomic per owner
    const int flat0 = cloud * a.ncode + lo;
    int4 *cellinfo = a.cellinfo + flat0;
    int firsts[kPer];
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
        const int c = tid * kPer + k;
        firsts[k] = kIntMax - aux[c];
        if (local[k] > 0) cellinfo[c] = make_int4(local[k], firsts[k], begin + run, -1);   // empty cells are never looked up
        cnt[c] = run;
        aux[c] = 0;
        run += local[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kPer; ++k)
        if (local[k] > 16) a.biglist[big_base + brun++] = flat0 + tid * kPer + k;
    int *bucket = a.bucket + begin;
#pragma unroll
    for (int u = 0; u < kKeep; ++u)
        if (mine[u].x >= 0) bucket[cnt[mine[u].x - lo] + atomicAdd(&aux[mine[u].x - lo], 1)] = begin + mine[u].y;
    for (int e = tid + kKeep * kOwner; e < total; e += kOwner) {
        const int2 p = element(e);
        bucket[cnt[p.x - lo] + atomicAdd(&aux[p.x - lo], 1)] = begin + p.y;
    }
    // voxel heads (= occupied cells, at their smallest point index) per 1024-point block, summed over the owners in
    // blocksum: assign_kernel ranks the heads in point order from it
    __syncthreads();
    int *heads = pre;
    for (int b = tid; b < nb; b += kOwner) heads[b] = 0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kPer; ++k)
        if (local[k] > 0) atomicAdd(&heads[firsts[k] / kChunk], 1);
    __syncthreads();
    for (int b = tid; b < nb; b += kOwner)
        if (heads[b]) atomicAdd(&a.blocksum[cloud * a.max_blocks + b], heads[b]);
}

// head flags of this thread's kPerThread consecutive points (bit k) and their cells
__device__ __forceinline__ unsigned head_flags(const VoxArgs &a, int cloud, int begin, int n, int i0, int (&cells)[kPerThread]) {
    const int4 *cellinfo = a.cellinfo + (size_t)cloud * a.ncode;
    unsigned h = 0;
#pragma unroll
    for (int k = 0; k < kPerThread; ++k) {
        const int i = i0 + k;
        int c = -1;
        if (i < n) c = a.cell_of_point[begin + i];
        cells[k] = c;
        if (c >= 0 && cellinfo[c].y == i) h |= 1u << k;
    }
    return h;
}

__global__ __launch_bounds__(kBlock) void assign_kernel(const VoxArgs a) {
    __shared__ int wave_sum[kBlock / 64];
    __shared__ int heads_of[kMaxClouds], before_me;
    const int cloud = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // every workgroup sums the per-block head counts itself (a few hundred ints): the heads of each cloud -> its voxel total
    // and base row, the heads in the blocks before this one -> the rank of this block's first head
    for (int c = wave; c < a.n_clouds; c += kBlock / 64) {
        const int nb = (a.off[c + 1] - a.off[c] + kChunk - 1) / kChunk;
        int all = 0, before = 0;
        for (int b = lane; b < nb; b += 64) {
            const int h = a.blocksum[c * a.max_blocks + b];
            all += h;
            before += (c == cloud && b < (int)blockIdx.x) ? h : 0;
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) { all += __shfl_xor(all, d); before += __shfl_xor(before, d); }
        if (lane == 0) {
            heads_of[c] = all < a.max_voxels ? all : a.max_voxels;
            if (c == cloud) before_me = before;
        }
    }
    __syncthreads();
    int base = 0;
    for (int c = 0; c < cloud; ++c) base += heads_of[c];
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        int sum = 0;
        for (int c = 0; c < a.n_clouds; ++c) { a.voxel_counts[c] = heads_of[c]; sum += heads_of[c]; }
        a.voxel_counts[a.n_clouds] = sum;
    }
    const int begin = a.off[cloud], n = a.off[cloud + 1] - begin;
    if (blockIdx.x * kChunk >= n) return;
    int cells[kPerThread];
    const unsigned h = head_flags(a, cloud, begin, n, blockIdx.x * kChunk + threadIdx.x * kPerThread, cells);
    const int mine = __popc(h);
    int incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(incl, d);
        if (lane >= d) incl += t;
    }
    if (lane == 63) wave_sum[wave] = incl;
    __syncthreads();
    int rank = before_me + incl - mine;
    for (int w = 0; w < wave; ++w) rank += wave_sum[w];
    if (!h) return;
    int4 *cellinfo = a.cellinfo + (size_t)cloud * a.ncode;
#pragma unroll
    for (int k = 0; k < kPerThread; ++k) {
        if (!(h >> k & 1)) continue;
        if (rank < a.max_voxels) {     // a cell that would open voxel number >= max_voxels is dropped with all its points
            const int code = cells[k], v = base + rank;
            const int4 info = cellinfo[code];
            const int cnt = info.x;
            const int c = code_cell(a, code);
            const int x = c % a.grid[0], yz = c / a.grid[0];
            a.coords[v] = make_int4(cloud, yz / a.grid[1], yz % a.grid[1], x);
            a.num_points[v] = cnt < a.max_points ? cnt : a.max_points;
            a.seg[v] = make_int2(info.z, cnt);
            cellinfo[code].w = v;
        }
        ++rank;
    }
}

__device__ __forceinline__ int wave_min(int v) {    // DPP only (no LDS crossbar): row minimum by rotations, then the 4 rows
    int t;
    t = __builtin_amdgcn_update_dpp(0, v, 0x128, 0xf, 0xf, false); v = t < v ? t : v;   // row_ror:8
    t = __builtin_amdgcn_update_dpp(0, v, 0x124, 0xf, 0xf, false); v = t < v ? t : v;   // row_ror:4
    t = __builtin_amdgcn_update_dpp(0, v, 0x122, 0xf, 0xf, false); v = t < v ? t : v;   // row_ror:2
    t = __builtin_amdgcn_update_dpp(0, v, 0x121, 0xf, 0xf, false); v = t < v ? t : v;   // row_ror:1
    const int r0 = __builtin_amdgcn_readlane(v, 0), r1 = __builtin_amdgcn_readlane(v, 16);
    const int r2 = __builtin_amdgcn_readlane(v, 32), r3 = __builtin_amdgcn_readlane(v, 48);
    const int m01 = r0 < r1 ? r0 : r1, m23 = r2 < r3 ? r2 : r3;
    return m01 < m23 ? m01 : m23;
}

template <int N>
__device__ __forceinline__ int rank_in_row(int idx) {   // # of the other 15 lanes of this 16-lane row holding a smaller value
    if constexpr (N == 0) {
        return 0;
    } else {
        const int other = __builtin_amdgcn_update_dpp(0, idx, 0x120 + N, 0xf, 0xf, false);   // row_ror:N
        return (other < idx ? 1 : 0) + rank_in_row<N - 1>(idx);
    }
}

// gather_kernel, two kinds of workgroups in one launch (both grid-stride: the counts are only known on the device):
//   blockIdx.x <  big_blocks    one wavefront per entry of the big-cell list (> 16 points) -- first in the grid, they run longest;
//   blockIdx.x >= big_blocks    16 lanes per voxel (16 voxels per workgroup step): zero fill of every voxel's unused slots and
//                               the complete job for voxels of <= 16 points (rank by DPP row rotation, float4 gather).
// A big cell's segment holds its point indices in arbitrary order; wanted are the `keep` smallest, ascending.  Up to 64
// entries: one per lane, rank = number of smaller entries (readlane sweep).  Beyond: every lane's minimum over its strided
// share gives 64 distinct candidates, the keep-th smallest of which bounds the keep smallest of the whole segment; the
// entries <= that bound (keep..~64 of them unless the order is adversarial) are compacted into LDS and ranked as before.
__global__ __launch_bounds__(kBlock) void gather_kernel(const VoxArgs a, const int big_blocks, const int list_cap) {
    __shared__ int stage[kBlock / 64][kSegLds];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if ((int)blockIdx.x >= big_blocks) {
        const int group = lane >> 4, gl = lane & 15, small_blocks = gridDim.x - big_blocks;
        const int total = a.voxel_counts[a.n_clouds];
        for (int v = (((int)blockIdx.x - big_blocks) * (kBlock / 64) + wave) * 4 + group; v < ((total + 3) & ~3); v += small_blocks * 16) {
            const bool live = v < total;
            const int2 seg = live ? a.seg[v] : make_int2(0, 0);
            const int cnt = seg.y;
            const int keep = cnt < a.max_points ? cnt : a.max_points;
            float4 *row = a.voxels + (size_t)v * a.max_points;
            const bool small = live && cnt <= 16;
            const int idx = (small && gl < cnt) ? a.bucket[seg.x + gl] : kIntMax;
            const int rank = rank_in_row<15>(idx);          // all 64 lanes take part: the loop bound is wave-uniform
            if (small && gl < cnt && rank < keep) row[rank] = a.pts[idx];
            if (live)
                for (int slot = keep + gl; slot < a.max_points; slot += 16) row[slot] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        return;
    }
    const int waves = big_blocks * (kBlock / 64);
    int item = blockIdx.x * (kBlock / 64) + wave;
    int entry = item < list_cap ? a.biglist[item] : 0;       // fetched beside the count, garbage past it
    const int nbig = *a.big_count;
    int *st = stage[wave];
    for (; item < nbig; item += waves, entry = item < nbig ? a.biglist[item] : 0) {
        const int4 info = a.cellinfo[entry];
        const int gcnt = info.x, gstart = info.z, v = info.w;
        if (v < 0) continue;                                 // cell dropped by max_voxels
        const int keep = gcnt < a.max_points ? gcnt : a.max_points;
        float4 *row = a.voxels + (size_t)v * a.max_points;
        int cand = gcnt;                                     // candidates to rank, wave-uniform
        int idx = kIntMax;
        if (gcnt <= 64) {
            if (lane < gcnt) idx = a.bucket[gstart + lane];
        } else {
            constexpr int kRegs = kSegLds / 64;
            const bool inreg = gcnt <= kSegLds;
            int vals[kRegs], mine = kIntMax;
            if (inreg) {
#pragma unroll
                for (int u = 0; u < kRegs; ++u) {
                    const int e = lane + u * 64;
                    vals[u] = e < gcnt ? a.bucket[gstart + e] : kIntMax;
                }
#pragma unroll
                for (int u = 0; u < kRegs; ++u) mine = vals[u] < mine ? vals[u] : mine;
            } else {
                for (int e = lane; e < gcnt; e += 64) {
                    const int t = a.bucket[gstart + e];
                    mine = t < mine ? t : mine;
                }
            }
            int mrank = 0;
            for (int j = 0; j < 64; ++j) mrank += __builtin_amdgcn_readlane(mine, j) < mine ? 1 : 0;
            const int bound = wave_min(mrank == keep - 1 ? mine : kIntMax);
            int filled = 0;
            auto push = [&](int t) {
                const unsigned long long take = __ballot(t <= bound);
                const int pos = filled + __popcll(take & ((1ull << lane) - 1ull));
                if (t <= bound && pos < kSegLds) st[pos] = t;
                filled += __popcll(take);
            };
            if (inreg) {
#pragma unroll
                for (int u = 0; u < kRegs; ++u) push(vals[u]);
            } else {
                for (int e0 = 0; e0 < gcnt; e0 += 64) push(e0 + lane < gcnt ? a.bucket[gstart + e0 + lane] : kIntMax);
            }
            coalign::wave_lds_sync();
            cand = filled;
            if (cand <= 64) {
                if (lane < cand) idx = st[lane];
            } else {
                // more than 64 candidates (adversarial segment order): selection rounds, over the LDS copy if it was complete
                const bool staged = cand <= kSegLds;
                const int m_cnt = staged ? cand : gcnt;
                int last = -1, chosen = -1;                  // lane r ends up with the r-th smallest point index
                for (int r = 0; r < keep; ++r) {
                    int m = kIntMax;
                    for (int e = lane; e < m_cnt; e += 64) {
                        const int t = staged ? st[e] : a.bucket[gstart + e];
                        m = (t > last && t < m) ? t : m;
                    }
                    last = wave_min(m);
                    if (lane == r) chosen = last;
                }
                if (lane < keep) row[lane] = a.pts[chosen];
                coalign::wave_lds_sync();
                continue;
            }
        }
        int rank = 0;
        for (int j = 0; j < cand; ++j) rank += __builtin_amdgcn_readlane(idx, j) < idx ? 1 : 0;
        if (lane < cand && rank < keep) row[rank] = a.pts[idx];
        coalign::wave_lds_sync();                            // the next item reuses the stage
    }
}

struct Workspace {
    size_t cellinfo, cell_of_point, split, table, bucket, blocksum, seg, biglist, big_count, total;
};

Workspace layout(int n_clouds, int64_t n_points, int64_t ncell, int64_t capacity, int max_blocks) {
    Workspace w;
    const int64_t ranges = (ncell + kRange - 1) / kRange;      // cell codes: ranges * kRange per cloud, see cell_code()
    size_t o = 0;
    auto take = [&](size_t ints) { size_t at = o; o = coalign::align_up(o + ints * sizeof(int), 256); return at; };
    w.cellinfo = take((size_t)n_clouds * ranges * kRange * 4);
    w.cell_of_point = take((size_t)n_points);
    w.split = take((size_t)n_points * 2);
    w.table = take((size_t)n_clouds * max_blocks * (ranges + 1));
    w.bucket = take((size_t)n_points);
    w.blocksum = take((size_t)n_clouds * max_blocks);
    w.seg = take((size_t)capacity * 2);
    w.biglist = take((size_t)n_points / 17 + 1);
    w.big_count = take(1);
    w.total = o;
    return w;
}

int grid_of(const double *voxel_size, const double *range, int g[3]) {
    for (int j = 0; j < 3; ++j) {   // float32 round((max - min) / voxel), sp_voxel_preprocessor.py:40-42
        const float q = ((float)range[3 + j] - (float)range[j]) / (float)voxel_size[j];
        if (!(q >= 0.5f && q < 65536.0f)) return COALIGN_ERR_BAD_SHAPE;
        g[j] = (int)nearbyintf(q);
    }
    if ((int64_t)g[0] * g[1] * g[2] > (int64_t)kMaxOwners * kRange) return COALIGN_ERR_UNSUPPORTED;   // 21 M cells
    return COALIGN_OK;
}

int max_blocks_of(const int64_t *offsets, int n_clouds) {
    int64_t m = 1;
    for (int c = 0; c < n_clouds; ++c) m = std::max<int64_t>(m, (offsets[c + 1] - offsets[c] + kChunk - 1) / kChunk);
    return (int)m;
}

}  // namespace

extern "C" int64_t coalign_voxelize_capacity(int64_t n_points, int n_clouds, const double *voxel_size, const double *range,
                                             int max_voxels) {
    int g[3];
    if (n_points < 0 || n_clouds < 1 || !voxel_size || !range || max_voxels < 1 || grid_of(voxel_size, range, g) != COALIGN_OK) return -1;
    const int64_t ncell = (int64_t)g[0] * g[1] * g[2];
    return std::min<int64_t>(n_points, (int64_t)n_clouds * std::min<int64_t>(ncell, max_voxels));
}

extern "C" size_t coalign_voxelize_workspace_bytes(const int64_t *cloud_offsets, int n_clouds, const double *voxel_size,
                                                   const double *range, int max_voxels) {
    int g[3];
    if (!cloud_offsets || n_clouds < 1 || n_clouds > kMaxClouds || grid_of(voxel_size, range, g) != COALIGN_OK) return 0;
    const int64_t n = cloud_offsets[n_clouds], ncell = (int64_t)g[0] * g[1] * g[2];
    const int64_t cap = std::min<int64_t>(n, (int64_t)n_clouds * std::min<int64_t>(ncell, max_voxels));
    return layout(n_clouds, n, ncell, cap, max_blocks_of(cloud_offsets, n_clouds)).total;
}

extern "C" int coalign_voxelize(const float *points, const int64_t *cloud_offsets, int n_clouds, const double *voxel_size,
                                const double *range, int max_points, int max_voxels, int flags, const double *filter_range,
                                float *voxels, int32_t *coords, int32_t *num_points, int64_t capacity, int32_t *voxel_counts,
                                void *workspace, size_t workspace_bytes, void *stream) {
    using namespace coalign;
    if (!cloud_offsets || !voxel_size || !range || !voxel_counts) return COALIGN_ERR_NULL_POINTER;
    if (n_clouds < 1 || max_points < 1 || max_voxels < 1) return COALIGN_ERR_BAD_SHAPE;
    if (n_clouds > kMaxClouds || max_points > 64) return COALIGN_ERR_UNSUPPORTED;
    if (flags & ~(COALIGN_VOX_FILTER_EGO | COALIGN_VOX_FILTER_RANGE)) return COALIGN_ERR_UNSUPPORTED;
    if ((flags & COALIGN_VOX_FILTER_RANGE) && !filter_range) return COALIGN_ERR_NULL_POINTER;
    if (cloud_offsets[0] != 0) return COALIGN_ERR_BAD_SHAPE;
    for (int c = 0; c < n_clouds; ++c)
        if (cloud_offsets[c + 1] < cloud_offsets[c]) return COALIGN_ERR_BAD_SHAPE;
    const int64_t n = cloud_offsets[n_clouds];
    if (n > (int64_t)1 << 30) return COALIGN_ERR_UNSUPPORTED;
    VoxArgs a{};
    int rc = grid_of(voxel_size, range, a.grid);
    if (rc != COALIGN_OK) return rc;
    const int64_t ncell = (int64_t)a.grid[0] * a.grid[1] * a.grid[2];
    const int64_t need = std::min<int64_t>(n, (int64_t)n_clouds * std::min<int64_t>(ncell, max_voxels));
    if (capacity < need) return COALIGN_ERR_BAD_SHAPE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (n == 0) return fill_words(voxel_counts, (size_t)(n_clouds + 1), 0u, s);
    if (!points || !voxels || !coords || !num_points || !workspace) return COALIGN_ERR_NULL_POINTER;
    a.max_blocks = max_blocks_of(cloud_offsets, n_clouds);
    if (a.max_blocks > kMaxBlocks) return COALIGN_ERR_UNSUPPORTED;     // > 2 M points in one cloud
    const Workspace w = layout(n_clouds, n, ncell, need, a.max_blocks);
    if (workspace_bytes < w.total) return COALIGN_ERR_WORKSPACE;
    if ((reinterpret_cast<uintptr_t>(points) | reinterpret_cast<uintptr_t>(voxels) | reinterpret_cast<uintptr_t>(coords)) & 15)
        return COALIGN_ERR_UNSUPPORTED;     // float4 / int4 accesses
    char *ws = static_cast<char *>(workspace);
    a.pts = reinterpret_cast<const float4 *>(points);
    a.n_clouds = n_clouds;
    for (int c = 0; c <= n_clouds; ++c) a.off[c] = (int)cloud_offsets[c];
    for (int j = 0; j < 3; ++j) {
        a.lo[j] = (float)range[j];
        a.vs[j] = (float)voxel_size[j];
        if (flags & COALIGN_VOX_FILTER_RANGE) {
            a.flo[j] = (float)filter_range[j];
            a.fhi[j] = (float)filter_range[3 + j];
        }
    }
    a.ncell = (int)ncell;
    a.ranges = (int)((ncell + kRange - 1) / kRange);
    a.ncode = a.ranges * kRange;
    a.max_points = max_points;
    a.max_voxels = max_voxels;
    a.capacity = (int)need;
    a.flags = flags;
    a.cellinfo = reinterpret_cast<int4 *>(ws + w.cellinfo);
    a.split = reinterpret_cast<int2 *>(ws + w.split);
    a.table = reinterpret_cast<int *>(ws + w.table);
    a.cell_of_point = reinterpret_cast<int *>(ws + w.cell_of_point);
    a.bucket = reinterpret_cast<int *>(ws + w.bucket);
    a.blocksum = reinterpret_cast<int *>(ws + w.blocksum);
    a.seg = reinterpret_cast<int2 *>(ws + w.seg);
    a.biglist = reinterpret_cast<int *>(ws + w.biglist);
    a.big_count = reinterpret_cast<int *>(ws + w.big_count);
    a.voxels = reinterpret_cast<float4 *>(voxels);
    a.coords = reinterpret_cast<int4 *>(coords);
    a.num_points = num_points;
    a.voxel_counts = voxel_counts;
    const dim3 per_point(a.max_blocks, n_clouds);
    // gather grid: enough workgroups for the worst case, capped -- both kinds of workgroup stride over the device-side counts
    const unsigned small_blocks = (unsigned)std::min<int64_t>((need + 15) / 16, 8192);
    const int64_t list_cap = n / 17 + 1;
    const unsigned big_blocks = (unsigned)std::min<int64_t>((list_cap + 3) / 4, 8192);
    hipLaunchKernelGGL(split_kernel, per_point, dim3(kBlock), 0, s, a);
    hipLaunchKernelGGL(owner_kernel, dim3(a.ranges, n_clouds), dim3(kOwner), 0, s, a);
    hipLaunchKernelGGL(assign_kernel, per_point, dim3(kBlock), 0, s, a);
    hipLaunchKernelGGL(gather_kernel, dim3(big_blocks + small_blocks), dim3(kBlock), 0, s, a, (int)big_blocks, (int)list_cap);
    return check_launch();
}
