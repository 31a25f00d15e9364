/*
 * coalign_amd.h -- C ABI of the MI355X (gfx950) implementation of the CoAlign per-frame detection hot path.
 *
 * Drop-in boundary.  The reference (yifanlu0227/CoAlign) is pure Python/PyTorch on this path,
 * so "what its FFI would bind" is a ctypes stub inside the opencood modules named below; INTEGRATION.md shows
 * those stubs.  Every entry point
 *   - takes plain device pointers and sizes (no torch types), all tensors dense row-major, float32 unless noted;
 *   - never allocates, never synchronises, never exits: outputs and workspaces are caller-allocated,
 *     kernels are enqueued on the hipStream_t passed as `stream` (NULL = the null stream);
 *   - returns 0 (COALIGN_OK) or a negative coalign_status; coalign_status_string() names it and
 *     coalign_last_hip_error() returns the hipError_t string captured by the calling thread's last failure;
 *   - is re-entrant and keeps no mutable global state (one workspace per stream is the caller's business).
 *
 * Paths in comments are relative to the reference repository root.
 */
#ifndef COALIGN_AMD_H
#define COALIGN_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define COALIGN_ABI_VERSION 2      /* round 6: M_rows / range_flag parameters of (9b) _ex, (9d), (10b) */

typedef enum coalign_status {
    COALIGN_OK = 0,
    COALIGN_ERR_NULL_POINTER = -1,  /* a required pointer argument is NULL                     */
    COALIGN_ERR_BAD_SHAPE = -2,     /* negative / inconsistent dimensions                      */
    COALIGN_ERR_UNSUPPORTED = -3,   /* valid request outside what the kernels implement        */
    COALIGN_ERR_WORKSPACE = -4,     /* workspace_bytes smaller than the *_workspace_bytes query */
    COALIGN_ERR_HIP = -5            /* a HIP runtime call failed; see coalign_last_hip_error() */
} coalign_status;

/* bits of the device-side status word written by the post-processing kernels */
#define COALIGN_FLAG_CANDIDATE_OVERFLOW 1u /* more candidates passed the score threshold than `capacity` */

int coalign_abi_version(void);
const char *coalign_status_string(int status);
const char *coalign_last_hip_error(void);

/* ---------------------------------------------------------------------------------------------------------
 * (1) Pillar feature encoder + scatter to the dense BEV canvas.
 *     Replaces PillarVFE.forward        opencood/models/sub_modules/pillar_vfe.py:105-155 (PFNLayer :31-53)
 *          and PointPillarScatter.forward opencood/models/sub_modules/point_pillar_scatter.py:15-72.
 *
 * voxel_features  [M, P, 4]  (x, y, z, intensity), rows >= num_points are padding
 * voxel_num_points[M] int32, voxel_coords [M, 4] int32 = (agent, z, y, x)
 * pfn_weight      [C, Cin]   Cin = 4*use_absolute_xyz + 1*(!use_absolute_xyz) + 6 + with_distance  (10 by default)
 * pfn_bias        [C] or NULL (only when the layer was built with use_norm = false)
 * bn_weight/bn_bias/bn_mean/bn_var [C] or all NULL; bn_eps (1e-3 in the reference, pillar_vfe.py:25), eval mode
 * voxel_size[3], range_min[3]: HOST float64; pillar centre = coord * voxel + (voxel / 2 + range_min), the offset
 *                 being evaluated in float64 and rounded to float32 once, like pillar_vfe.py:84-89
 * pillar_features [M, C] out (required; the reference exposes it as batch_dict['pillar_features'])
 * canvas          [n_agents, C, ny, nx] out; cell index = z + y * nx + x (point_pillar_scatter.py:54), every
 *                 element is written (zeros where no pillar).  If two pillars of one agent share a cell the one
 *                 with the larger row index wins (the reference's sequential CPU indexing semantics).
 * workspace       coalign_pillar_scatter_workspace_bytes(n_agents, ny, nx) bytes
 */
size_t coalign_pillar_scatter_workspace_bytes(int n_agents, int ny, int nx);
int coalign_pillar_vfe_scatter(const float *voxel_features, const int32_t *voxel_num_points, const int32_t *voxel_coords,
                               int M, int P, const float *pfn_weight, const float *pfn_bias, const float *bn_weight,
                               const float *bn_bias, const float *bn_mean, const float *bn_var, float bn_eps, int C,
                               int use_absolute_xyz, int with_distance, const double *voxel_size, const double *range_min,
                               int n_agents, int ny, int nx, float *pillar_features, float *canvas, void *workspace,
                               size_t workspace_bytes, void *stream);
/* The same op writing the canvas CHANNELS-LAST, [n_agents, ny, nx, C] (the memory of a torch channels_last tensor of logical shape
 * [n_agents, C, ny, nx]; P <= 32, C <= 64): a pillar's feature row is its canvas cell, the scatter is one C-float store per pillar on
 * top of a memset.  Read in place by the strided convolutions of the first ResNet stage (coalign_conv3x3_emu_ex layout 2,
 * coalign_pointwise_conv_ex in_nhwc). */
int coalign_pillar_vfe_scatter_nhwc(const float *voxel_features, const int32_t *voxel_num_points, const int32_t *voxel_coords, int M, int P,
                                    const float *pfn_weight, const float *pfn_bias, const float *bn_weight, const float *bn_bias,
                                    const float *bn_mean, const float *bn_var, float bn_eps, int C, int use_absolute_xyz, int with_distance,
                                    const double *voxel_size, const double *range_min, int n_agents, int ny, int nx, float *pillar_features,
                                    float *canvas, void *workspace, size_t workspace_bytes, void *stream);
/* The persistent-canvas form, two launches per frame and no memset: canvas [n_agents, ny, nx, C] and cellmap [n_agents * ny * nx] int32
 * are kept by the caller across frames -- canvas zeroed and cellmap filled with -1 ONCE -- and dest [>= max(M, M_prev)] holds the slots the
 * previous call wrote (M_prev of them; 0 on the first call).  Launch 1 zeroes those rows and enters the new pillars into the cell map,
 * launch 2 encodes, writes pillar_features, the winners' canvas rows and the new dest, and leaves the cell map all -1 again. */
int coalign_pillar_encode_persistent(const float *voxel_features, const int32_t *voxel_num_points, const int32_t *voxel_coords, int M, int P,
                                     const float *pfn_weight, const float *pfn_bias, const float *bn_weight, const float *bn_bias,
                                     const float *bn_mean, const float *bn_var, float bn_eps, int C, int use_absolute_xyz, int with_distance,
                                     const double *voxel_size, const double *range_min, int n_agents, int ny, int nx, float *pillar_features,
                                     int32_t *dest, int M_prev, float *canvas, int32_t *cellmap, void *stream);
/* The same for a producer that leaves its pillar count ON THE DEVICE (coalign_voxelize's voxel_counts word): nothing on the host depends on
 * the count, so the call can sit behind the voxeliser in one stream / one captured HIP graph -- the reference's loop reads the count back
 * through collate_batch (sp_voxel_preprocessor.py:145-174) and train_utils.to_device (train_utils.py:249-258) before the model starts.
 * The arrays hold M_capacity rows, *M_dev (clamped to [0, M_capacity]) of them valid.  dest_state [M_capacity + 1] int32 is the caller's
 * persistent list: element 0 = rows the previous call wrote (zero it once), elements 1.. = their canvas slots.  pillar_features may be NULL
 * (the canvas is the only consumer in the inference loop).  unique_cells != 0: the caller guarantees at most one pillar per cell (true of
 * every voxeliser's output) -- no cell map is consulted and cellmap may be NULL; otherwise the duplicate rule of (1) applies. */
int coalign_pillar_encode_stream(const float *voxel_features, const int32_t *voxel_num_points, const int32_t *voxel_coords, int M_capacity,
                                 const int32_t *M_dev, int P, const float *pfn_weight, const float *pfn_bias, const float *bn_weight,
                                 const float *bn_bias, const float *bn_mean, const float *bn_var, float bn_eps, int C, int use_absolute_xyz,
                                 int with_distance, const double *voxel_size, const double *range_min, int n_agents, int ny, int nx,
                                 float *pillar_features, int32_t *dest_state, float *canvas, int32_t *cellmap, int unique_cells, void *stream);
/* Scatter only (PointPillarScatter.forward on already-encoded pillars): pillar_features [M, C] -> canvas.
 * Same cell rule, duplicate rule and workspace as above. */
int coalign_scatter_to_bev(const float *pillar_features, const int32_t *voxel_coords, int M, int C, int n_agents, int ny,
                           int nx, float *canvas, void *workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------------------------
 * (2) Pose-aware affine warp + multi-agent fusion, one launch per feature scale.
 *     Replaces warp_affine_simple   opencood/models/sub_modules/torch_transformation_utils.py:322-331
 *              AttFusion.forward    opencood/models/fuse_modules/fusion_in_one.py:96-136
 *                (ScaledDotProductAttention opencood/models/fuse_modules/att_fuse.py:43-47)
 *              MaxFusion.forward    opencood/models/fuse_modules/fusion_in_one.py:51-89
 *              warp_feature         opencood/models/fuse_modules/fusion_in_one.py:26-45
 *
 * x          [n_total, C, H, W]   agents of all frames concatenated (record_len order)
 * theta      [n_total, 2, 3] float64, DEVICE: for agent j of frame b the row normalized_affine[b, 0, j]
 *            (ego -> agent j in affine_grid's normalised coordinates).  The sampling grid is evaluated in
 *            float64 and then cast to float32, like F.affine_grid on a float64 theta followed by .to(src).
 * group_len  [n_groups] HOST int32, agents per frame (record_len); sum == n_total; each 1..8
 * mode       COALIGN_FUSE_ATT : out [n_groups, C, Ho, Wo]  softmax_j(<X0,Xj>/sqrt(C)) weighted sum, ego row only
 *            COALIGN_FUSE_MAX : out [n_groups, C, Ho, Wo]  elementwise max over the frame's agents
 *            COALIGN_FUSE_NONE: out [n_total,  C, Ho, Wo]  the warped maps themselves
 * Sampling: bilinear, zeros padding, align_corners = False.  C <= 256 for ATT/MAX.
 */
/* normalize_pairwise_tfm (opencood/utils/transformation_utils.py:69-91) on the device in one launch: pairwise [n_matrices, 4, 4]
 * float64 (T_{j<-i}, any leading shape flattened) -> out [n_matrices, 2, 3] float64 in affine_grid's normalised coordinates;
 * den_x = downsample_rate * discrete_ratio * W, den_y = ... * H, evaluated by the caller in double like the reference's Python scalars.
 * Bit-identical to the reference's tensor expression. */
int coalign_normalize_pairwise(const double *pairwise, int n_matrices, int H, int W, double den_x, double den_y, double *out, void *stream);

enum { COALIGN_FUSE_ATT = 0, COALIGN_FUSE_MAX = 1, COALIGN_FUSE_NONE = 2 };
int coalign_warp_fuse(const float *x, int n_total, int C, int H, int W, const double *theta, const int32_t *group_len,
                      int n_groups, int mode, float *out, int Ho, int Wo, void *stream);
/* The same op when the agents of a frame do not sit in x in agent order (agent-sharded execution, coalign_amd/sharded.py: the
 * receive buffer of the feature exchange is source-rank major).  rows [n_total] HOST int32 or NULL (= identity): for logical
 * agent i of frame b (i = 0 is the ego), rows[off_b + i] is the row of that frame's block of x that holds it -- a permutation of
 * 0..group_len[b]-1.  theta and the NONE-mode output stay in logical agent order; softmax / weighted sum run in logical order,
 * so the result is bit-identical to the un-routed call (fusion_in_one.py:125-132 fixes that order through regroup). */
int coalign_warp_fuse_rows(const float *x, int n_total, int C, int H, int W, const double *theta, const int32_t *group_len,
                           int n_groups, const int32_t *rows, int mode, float *out, int Ho, int Wo, void *stream);
/* The same op on CHANNELS-LAST maps, up to three feature scales of ONE frame in one launch (csrc/warp_fuse_nhwc.hip): x[i] is
 * [n, H[i], W[i], C[i]] (the memory of a torch channels_last tensor of logical shape [n, C, H, W]), out[i] is [Ho[i], Wo[i], C[i]]
 * (ATT / MAX) or [n, Ho[i], Wo[i], C[i]] (NONE), C[i] in {64, 128, 256}, 16-byte aligned.  x, C, H, W, out, Ho, Wo: HOST arrays of
 * n_scales entries; theta [n, 2, 3] float64 DEVICE (shared by the scales: the normalised affine is resolution independent,
 * point_pillar_baseline_multiscale.py:108-109); rows [n] HOST or NULL as in coalign_warp_fuse_rows. */
int coalign_warp_fuse_nhwc(int n_scales, const float *const *x, const int32_t *C, const int32_t *H, const int32_t *W, float *const *out,
                           const int32_t *Ho, const int32_t *Wo, int n, const double *theta, const int32_t *rows, int mode, void *stream);

/* ---------------------------------------------------------------------------------------------------------
 * (3) Anchor decode: sigmoid + score threshold + box decode + direction-bin fix + 8 corners + projection +
 *     size / z sanity filters, order-preserving compaction.
 *     Replaces VoxelPostprocessor.post_process opencood/data_utils/post_processor/voxel_postprocessor.py:275-377
 *              delta_to_boxes3d :405-450, limit_period opencood/utils/common_utils.py:70-79,
 *              boxes_to_corners_3d / project_box3d / remove_large_pred_bbx / remove_bbx_abnormal_z
 *              opencood/utils/box_utils.py:152-204, 278-316, 840-869 (including its y-for-z quirk), 872-890.
 *
 * cls [A, H, W], reg [7A, H, W], dir [num_bins*A, H, W] or NULL: one agent's head outputs (batch 1)
 * anchors [H*W*A, 7] float32 (x, y, z, h, w, l, yaw) in flat (h, w, anchor) order
 * transform [4, 4] float32 DEVICE row-major agent->ego, or NULL for identity
 * Candidates (score > score_thr) are appended in flat anchor order starting at position *count_in
 * (count_in == NULL -> 0); *count_out receives the new total, clamped to `capacity`, so several agents
 * (late fusion) can be chained without a host round trip.
 * cand_index [capacity] int32 flat anchor index, cand_score [capacity], cand_box7 [capacity, 7] (after the
 * direction fix), cand_corners [capacity, 8, 3] (projected), cand_keep [capacity] uint8 (passes both sanity
 * filters).  status: device uint32, COALIGN_FLAG_* bits are OR-ed in.  Any of cand_index/cand_box7 may be NULL.
 */
size_t coalign_anchor_decode_workspace_bytes(int A, int H, int W);
int coalign_anchor_decode(const float *cls, const float *reg, const float *dir, const float *anchors, int A, int H, int W,
                          int num_bins, float score_thr, float dir_offset, int order_hwl, const float *transform,
                          int capacity, const int32_t *count_in, int32_t *count_out, int32_t *cand_index,
                          float *cand_score, float *cand_box7, float *cand_corners, uint8_t *cand_keep, uint32_t *status,
                          void *workspace, size_t workspace_bytes, void *stream);
/* Round 6: the FIRST decode call of a frame also zeroes the frame's counter / status words (n_clear <= 256 32-bit words at clear_words; count_in, count_out and
 * status normally point into them: the per-frame state of voxel_postprocessor.py:243-402) -- one launch less in front of the post-processing chain. */
int coalign_anchor_decode_first(const float *cls, const float *reg, const float *dir, const float *anchors, int A, int H, int W,
                                int num_bins, float score_thr, float dir_offset, int order_hwl, const float *transform,
                                int capacity, const int32_t *count_in, int32_t *count_out, int32_t *cand_index,
                                float *cand_score, float *cand_box7, float *cand_corners, uint8_t *cand_keep, uint32_t *status,
                                void *workspace, size_t workspace_bytes, uint32_t *clear_words, int n_clear, void *stream);

/* ---------------------------------------------------------------------------------------------------------
 * (4) Rotated NMS.  Replaces nms_rotated opencood/utils/box_utils.py:693-738 (+ compute_iou / convert_format
 *     opencood/utils/common_utils.py:196-236, i.e. the Shapely polygon loop) and supersedes the bitmask NMS of
 *     opencood/pcdet_utils/iou3d_nms (src/iou3d_nms_kernel.cu:267-311 + host scan src/iou3d_nms.cpp:90-136).
 *
 * boxes  [K, rows, cols] float32, rows >= 4, cols >= 2: corners 0..3, columns (x, y) form the polygon
 *        ([K, 8, 3] and [K, 4, 2] are the two layouts the reference passes)
 * scores [K]; valid [K] uint8 or NULL; K_dev: device int32 holding the live K (<= K) or NULL
 * Semantics: consider valid boxes only, order by score descending (ties: larger index first), keep the first
 * `top` (1000 in the reference), greedy suppression with float32(IoU) > iou_thr where IoU is evaluated in
 * float64 (convex clipping, union = |A| + |B| - inter).  keep [top] int32 receives indices into `boxes` in pick
 * order, *keep_count their number.  top <= 4096.
 */
size_t coalign_nms_rotated_workspace_bytes(int K, int top);
int coalign_nms_rotated(const float *boxes, int rows, int cols, const float *scores, const uint8_t *valid, int K,
                        const int32_t *K_dev, float iou_thr, int top, int32_t *keep, int32_t *keep_count, void *workspace,
                        size_t workspace_bytes, void *stream);

/* (4) + (5) in one call: rank, suppression bitmask, greedy walk AND the in-range gather of coalign_gather_in_range (three launches
 * instead of four; the walk and the gather share one workgroup that holds the whole bitmask in LDS).  corners [K, 8, 3] are the boxes the
 * NMS clips (their first four corners' x, y, box_utils.py:716-717) and the rows the gather copies.  top <= 1024 (the reference uses 1000).
 * keep / keep_count as in coalign_nms_rotated; out_* as in coalign_gather_in_range. */
int coalign_nms_rotated_gather(const float *corners, const float *scores, const uint8_t *valid, int K, const int32_t *K_dev, float iou_thr, int top,
                               int32_t *keep, int32_t *keep_count, const double *range6_host, float *out_corners, float *out_scores,
                               int32_t *out_count, void *workspace, size_t workspace_bytes, void *stream);
/* Gather the kept boxes and drop those with a corner outside `range` (xmin, ymin, zmin, xmax, ymax, zmax;
 * compared in float64 like mask_boxes_outside_range_numpy, opencood/utils/box_utils.py:384-421 with
 * min_num_corners = 8), preserving pick order (voxel_postprocessor.py:385-397).
 * corners [*, 8, 3], scores [*], keep [keep_cap] + *keep_count from coalign_nms_rotated
 * out_corners [keep_cap, 8, 3], out_scores [keep_cap], *out_count.  keep_cap <= 4096. */
int coalign_gather_in_range(const float *corners, const float *scores, const int32_t *keep, const int32_t *keep_count,
                            int keep_cap, const double *range6_host, float *out_corners, float *out_scores,
                            int32_t *out_count, void *stream);

/* Rotated IoU matrix with the same float64 clipping as the NMS (union = |A| + |B| - inter, result rounded to float32):
 * the numbers opencood/utils/eval_utils.py:45-96 (caluclate_tp_fp) obtains from common_utils.compute_iou, one Shapely
 * call per (detection, ground truth) pair.  boxes_a [Na, rows_a, cols_a], boxes_b [Nb, rows_b, cols_b] as in (4)
 * -> iou [Na, Nb]. */
int coalign_iou_rotated_matrix(const float *boxes_a, int rows_a, int cols_a, int Na, const float *boxes_b, int rows_b, int cols_b,
                               int Nb, float *iou, void *stream);

/* ---------------------------------------------------------------------------------------------------------
 * (5) OpenPCDet-semantics BEV IoU (fp32 overlap with its 1e-2 corner margin) for callers of
 *     opencood/pcdet_utils/iou3d_nms/iou3d_nms_utils.py (boxes_iou_bev :32-46, boxes_iou3d_gpu :147-181, nms_gpu :255-271, nms_normal_gpu :274-289).
 * boxes_a [Na, 7], boxes_b [Nb, 7] = (x, y, z, dx, dy, dz, heading) -> iou [Na, Nb]
 */
int coalign_boxes_iou_bev(const float *boxes_a, int Na, const float *boxes_b, int Nb, float *iou, void *stream);
/* same pairing, the overlap AREA instead of the IoU (iou3d_nms_cuda.boxes_overlap_bev_gpu, used by boxes_iou3d_gpu :147-181) */
int coalign_boxes_overlap_bev(const float *boxes_a, int Na, const float *boxes_b, int Nb, float *overlap, void *stream);

/* Device bitmask NMS with OpenPCDet semantics -- replaces iou3d_nms_cuda.nms_gpu / nms_normal_gpu
 * (opencood/pcdet_utils/iou3d_nms/src/iou3d_nms.cpp:90-137 + the host scan of the mask :121-133; kernels
 * src/iou3d_nms_kernel.cu:267-311 nms_kernel, :328-372 nms_normal_kernel).  boxes_sorted [n, 7] must already be in descending
 * score order (the Python caller sorts, like iou3d_nms_utils.py:264-269); a box is kept unless an earlier KEPT box has
 * fp32 IoU > thresh with it (normal != 0: heading ignored, axis-aligned IoU of iou_normal :313-325).
 * -> keep [n] int32 positions into boxes_sorted, ascending; *keep_count.  Nothing leaves the device (the reference copies the
 * n x n/64 mask to the host and walks it there).  n <= 16384. */
size_t coalign_pcdet_nms_workspace_bytes(int n);
int coalign_pcdet_nms(const float *boxes_sorted, int n, float thresh, int normal, int32_t *keep, int32_t *keep_count, void *workspace,
                      size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------------------------
 * (6) Fused convolution epilogue for the dense stages (rows E / I): in place  y = act(y + bias[c] (+ residual)).
 *     With eval-mode BatchNorm folded into the convolution weights on the host, this one pass replaces the
 *     BatchNorm2d / ReLU / residual-add kernels of BasicBlock.forward (opencood/models/sub_modules/resblock.py:53-69),
 *     of the deblocks (base_bev_backbone_resnet.py:121-138) and the bias + ReLU of DoubleConv (downsample_conv.py:7-27).
 * y [N, C, HW] in/out, bias [C] or NULL, residual [N, C, HW] or NULL, relu 0/1.
 */
int coalign_bias_act(float *y, const float *bias, const float *residual, int N, int C, int HW, int relu, void *stream);

/* ---------------------------------------------------------------------------------------------------------
 * (7) Points -> pillars voxeliser, the step in front of (1).
 *     Replaces SpVoxelPreprocessor.preprocess + collate_batch
 *     (opencood/data_utils/pre_processor/sp_voxel_preprocessor.py:62-85, 107-174), i.e. spconv's CPU
 *     VoxelGeneratorV2 / Point2VoxelCPU3d run per cav in the DataLoader workers, and optionally the two point filters in
 *     front of it (opencood/utils/pcd_utils.py:41-66 mask_points_by_range, :69-88 mask_ego_points; call sites
 *     intermediate_fusion_dataset.py:96-117, late_fusion_dataset.py:157-170).  shuffle_points (a random permutation) stays
 *     with the caller: the output is a deterministic function of the point ORDER, exactly like the sequential CPU loop.
 *
 * points         [N, 4] float32 (x, y, z, intensity), the clouds of one batch concatenated; 16-byte aligned
 * cloud_offsets  HOST int64 [n_clouds + 1], cloud c = points[cloud_offsets[c] : cloud_offsets[c + 1]], n_clouds <= 16
 * voxel_size[3], range[6] (xmin ymin zmin xmax ymax zmax): HOST float64, used as float32 like spconv does;
 *                grid = round((max - min) / voxel); a point is kept iff 0 <= floor((p - min) / voxel) < grid on all axes
 * max_points     points kept per voxel (first come, point order), <= 64;  max_voxels  voxels opened per cloud (cells met
 *                after that are dropped, points of already open cells are still taken)
 * flags          COALIGN_VOX_FILTER_EGO and / or COALIGN_VOX_FILTER_RANGE (filter_range: HOST float64 [6], strict
 *                inequalities evaluated in float32); a filtered point is skipped, which equals filtering first
 * voxels         [capacity, max_points, 4] out, rows past a voxel's count are zero     (`voxel_features`)
 * coords         [capacity, 4] int32 out = (cloud, z, y, x)                           (`voxel_coords` after collate)
 * num_points     [capacity] int32 out                                                 (`voxel_num_points`)
 * voxel_counts   [n_clouds + 1] int32 out (device): voxels of each cloud, then their sum M; cloud c owns the rows
 *                [sum(counts[:c]), sum(counts[:c + 1])), numbered by first appearance in point order.  Rows >= M are
 *                not written.  capacity >= coalign_voxelize_capacity(...) = min(N, n_clouds * min(cells, max_voxels)).
 * Limits (COALIGN_ERR_UNSUPPORTED beyond): 16 clouds, 2 M points per cloud, 20 971 520 grid cells, max_points 64.
 * workspace      coalign_voxelize_workspace_bytes(...) bytes (dense per-cell table 16 B x cells x clouds + 20 B per point).
 */
#define COALIGN_VOX_FILTER_EGO 1
#define COALIGN_VOX_FILTER_RANGE 2
int64_t coalign_voxelize_capacity(int64_t n_points, int n_clouds, const double *voxel_size, const double *range, int max_voxels);
size_t coalign_voxelize_workspace_bytes(const int64_t *cloud_offsets, int n_clouds, const double *voxel_size,
                                        const double *range, int max_voxels);
int coalign_voxelize(const float *points, const int64_t *cloud_offsets, int n_clouds, const double *voxel_size,
                     const double *range, int max_points, int max_voxels, int flags, const double *filter_range,
                     float *voxels, int32_t *coords, int32_t *num_points, int64_t capacity, int32_t *voxel_counts,
                     void *workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------------------------
 * (8) Agent-object pose-graph optimisation, the solver inside CoAlign's box alignment.
 *     Replaces PoseGraphOptimization2D.optimize (opencood/models/sub_modules/pose_graph_optim.py:11-60: g2o SparseOptimizer,
 *     Levenberg-Marquardt, dense SE2 block solver) as called by box_alignment_relative_sample_np
 *     (opencood/models/sub_modules/box_align_v2.py:287-372; hook intermediate_fusion_dataset.py:301-328).
 *     A batch of independent graphs (one per frame) is solved in one launch, float64.
 *
 * Graph g owns vertices [vertex_offsets[g], vertex_offsets[g+1]) and edges [edge_offsets[g], edge_offsets[g+1]) (device int32).
 * Its first n_agents[g] (<= 8) vertices are the agents, the rest (<= 256) the landmarks; every edge joins an agent to a
 * landmark (g2o EdgeSE2 / EdgeSE2PointXY) and the edges of one landmark are contiguous (the reference adds them that way).
 * vertices   [V, 3] float64 in/out: (x, y, theta in radians); point landmarks use (x, y, -)
 * kinds      [V] int32: 0 = fixed SE2 (the ego), 1 = free SE2, 2 = free point (landmarks only)
 * edge_agent / edge_landmark  [E] int32, vertex numbers LOCAL to the graph
 * edge_meas  [E, 3] float64: the landmark in the agent's frame (x, y, theta; theta ignored for point landmarks)
 * edge_info  [E, 3] float64: diagonal of the information matrix (the reference only ever builds diagonal ones)
 * stats      [n_graphs, 4] float64 out: LM iterations run (-1: graph outside the limits above, left untouched),
 *            chi2 before, chi2 after, final lambda
 * Damping schedule, update rule (X <- X * dx), termination (rho == 0, ten rejected trials, max_iterations) follow g2o's
 * OptimizationAlgorithmLevenberg; vertices without edges are not moved.
 */
size_t coalign_pose_graph_workspace_bytes(int total_vertices);
int coalign_pose_graph_optimize(int n_graphs, const int32_t *vertex_offsets, const int32_t *edge_offsets, const int32_t *n_agents,
                                int total_vertices, double *vertices, const int32_t *kinds, const int32_t *edge_agent,
                                const int32_t *edge_landmark, const double *edge_meas, const double *edge_info,
                                int max_iterations, double *stats, void *workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------------------------
 * (9) 3x3 / stride 1 / padding 1 convolution with the epilogue of (6) fused, on the fp32 matrix cores (rows E / I).
 *     y = act(conv3x3(x, w) + bias[c] (+ residual)): Conv2d + eval BatchNorm2d (folded into w, bias on the host) + ReLU of
 *     BasicBlock.forward (opencood/models/sub_modules/resblock.py:53-69), the ResNet stages
 *     (base_bev_backbone_resnet.py:59-119) and DoubleConv of the shrink header (downsample_conv.py:7-50).
 * x [N, Cin, H, W], y / residual [N, Cout, H, W] float32 NCHW, Cin % 8 == 0, Cout % 64 == 0, x and y distinct buffers;
 * bias [Cout] required (zeros for none), residual may be NULL.
 * w_packed [Cout / 64][Cin / 8][8][608] float32, 16-byte aligned: the LDS image of one 8-input-channel chunk,
 *   w_packed[g][ci / 8][ci % 8][(ky * 3 + kx) * 64 + j] = w[g * 64 + j][ci][ky][kx],  entries [..][576..607] = 0.
 * workspace: coalign_conv3x3_workspace_bytes(...) bytes (flags + partial tiles of the stream-K work split; flags are
 * cleared by the call itself).  W % 4 == 0.
 * Products and sums are exact float32 (v_mfma_f32_32x32x2_f32); the summation order differs from a sequential loop but is a
 * pure function of the shape (deterministic).
 */
size_t coalign_conv3x3_workspace_bytes(int N, int Cin, int Cout, int H, int W);
int coalign_conv3x3_bias_act(const float *x, const float *w_packed, const float *bias, const float *residual, float *y,
                             int N, int Cin, int Cout, int H, int W, int relu, void *workspace, size_t workspace_bytes, void *stream);

/* (9b) The DEFAULT route of the detector's 3x3 convolutions since round 2 (terms = 3; COALIGN_CONV_EMU=0 selects (9)): the same layers, fp32 in / fp32 out, with every fp32 product evaluated on the bf16 matrix cores by
 * error-free operand splitting (x = x_h + x_m + x_l, bf16 each; cross terms accumulated in fp32, smallest first).
 *   terms = 3: six bf16 products per fp32 product, dropped terms <= 2^-24 |w x| -- fp32-level accuracy;
 *   terms = 2: three products, dropped terms <= 2^-16 |w x|.
 * w_split: coalign_conv3x3_emu_weight_bytes(Cin, Cout, terms) bytes, 16-byte aligned:
 *   [Cout / 64][Cin / 8][5 steps][terms][2 k-groups][64 cout][8 cin] bf16 with tap = 2 * step + k-group (the tenth tap zero),
 *   term 0 = bf16(w), term 1 = bf16(w - term 0), term 2 = bf16(w - term 0 - term 1), followed by 16 zero bytes.
 * Cin % 8 == 0, Cout % 64 == 0, any H, W (no alignment requirement on x).  bias is required; residual may be NULL.
 * workspace: coalign_conv3x3_emu_workspace_bytes(...) bytes of device scratch (0 = none needed for that shape) for the stream-K
 * hand-over of tiles split between two workgroups; one workspace per stream, not shared between concurrent launches.  The split is a
 * pure function of the shape (deterministic).
 * terms = 3 is the detector's default arithmetic, terms = 2 is opt-in (COALIGN_CONV_EMU, coalign_amd/backbone.py).
 */
size_t coalign_conv3x3_emu_weight_bytes(int Cin, int Cout, int terms);
size_t coalign_conv3x3_emu_workspace_bytes(int N, int Cin, int Cout, int H, int W, int terms);
int coalign_conv3x3_emu_bias_act(const float *x, const void *w_split, const float *bias, const float *residual, float *y,
                                 int N, int Cin, int Cout, int H, int W, int relu, int terms, void *workspace, size_t workspace_bytes,
                                 void *stream);
/* The same convolution with a stride and a memory layout: stride 1 or 2 (3x3, pad 1: output ceil(Hin / stride) x ceil(Win / stride);
 * the strided first convolution of a ResNet stage, resblock.py:150-174), layout 0 = NCHW in and out, 1 = NCHW in / channels-last
 * (NHWC) out (stride 1 only: the last convolution of a stage, whose map the fusion kernel and the next stage read), 2 = NHWC in /
 * NCHW out (stride 2 only).  residual (stride 1) is always NCHW.  workspace as coalign_conv3x3_emu_workspace_bytes for
 * (stride 1, layout 0); the other variants need none. */
enum { COALIGN_LAYOUT_NCHW = 0, COALIGN_LAYOUT_OUT_NHWC = 1, COALIGN_LAYOUT_IN_NHWC = 2, COALIGN_LAYOUT_W_TAPMAJOR = 4, COALIGN_LAYOUT_OUT_SP = 8 };
/* COALIGN_LAYOUT_OUT_SP (round 5; flag, terms = 16 only, no residual): y is an SP map (9e) instead of a float32 tensor -- with stride 2 (NCHW or
 * channels-last input, tap-pair image) and with COALIGN_LAYOUT_W_TAPMAJOR (stride 1, NCHW input): the layers in front of a chain of coalign_conv3x3_sp. */
/* Round 4: layout 3 (= IN_NHWC | OUT_NHWC) with stride 2: channels-last in AND out -- the strided first convolution of a ResNet stage in front of
 * the Winograd layers (9c), which read and write channels-last. */
/* COALIGN_LAYOUT_W_TAPMAJOR (flag, or-ed into layout 0 or 1, stride 1, Cin % 16 == 0): w_split is the TAP-MAJOR image
 *   [Cout / 64][Cin / 16][9 taps][terms][2 channel halves][64 cout][8 cin] bf16 (+ 16 zero bytes),
 * coalign_conv3x3_emu_weight_bytes_ex(Cin, Cout, terms, 1) bytes: one matrix instruction = the 16 channels of one tap, nine per 16
 * channels instead of ten (no zero tenth tap).  Workspace: coalign_conv3x3_emu_workspace_bytes_ex(..., layout). */
size_t coalign_conv3x3_emu_weight_bytes_ex(int Cin, int Cout, int terms, int tap_major);
size_t coalign_conv3x3_emu_workspace_bytes_ex(int N, int Cin, int Cout, int H, int W, int terms, int layout);
int coalign_conv3x3_emu_ex(const float *x, const void *w_split, const float *bias, const float *residual, float *y, int N, int Cin,
                           int Cout, int Hin, int Win, int stride, int relu, int terms, int layout, int32_t *range_flag, void *workspace,
                           size_t workspace_bytes, void *stream);      /* range_flag (SP map outputs; may be NULL): as in (9e) */

/* terms = 16 (round 4; scale free since round 5): the fp16 split.  Every operand is an "sp16 pair" (csrc/common.h): the value rounded to 22 significant bits,
 * h = its leading 11 bits as fp16, l = the following 11 bits times 2^10 (a normal fp16 number for every |value| >= 2^-14); products w_h x_h in one fp32
 * accumulator, w_h x_l + w_l x_h in a second one that enters with 2^-10; w_l x_l (< 2^-20 |w x|) dropped.  The weight image is built from the weights
 * multiplied per OUTPUT CHANNEL by the power of two that puts the channel's largest weight into [2^13, 2^14), and ends in [Cout] float32 2^-k_c and [Cout]
 * float32 2^k_c (coalign_conv3x3_emu_weight_bytes* include them): exact, so the result does not depend on the scale of the weights.  Operating range of the
 * activations: |x| <= 65504 (beyond it operands clamp, finite).  Against a float64 convolution the error is <= the native fp32 kernel's (9) over weight scales
 * 2e-4 ... 2e-1 x activation scales 1e-2 ... 1e2, Gaussian and Student-t (tests/test_round5_gpu.py). */

/* (9e) Round 5: 3x3 / stride 1 convolutions whose input (and usually output) is an SP MAP -- the activations stored as sp16 pairs in the matrix
 * instruction's operand order, so that the consumer's K loop is LDS-DMA + matrix instructions only (csrc/conv3x3_sp.hip).  Same layers as (9b):
 * BasicBlock.forward opencood/models/sub_modules/resblock.py:53-69, the ResNet stages base_bev_backbone_resnet.py:59-119, DoubleConv downsample_conv.py:7-27.
 *   SP map of a logical [N, C, H, W] tensor, C % 16 == 0: [N][C / 16][4][H][W][8] fp16, plane index = 2 * (channel / 8 % 2) + term; element e of a
 *   16-byte group = channel 16 (c / 16) + 8 (plane / 2) + e; coalign_sp_map_bytes = N * C * H * W * 4 bytes, 16-byte aligned.
 *   coalign_sp_pack / coalign_sp_unpack: float32 (NCHW, or channels-last if *_nhwc != 0) <-> SP map (unpack(pack(x)) = x rounded to 22 significant bits).
 *   coalign_conv3x3_sp: y = relu?(conv3x3(x, w) + bias + residual);  w_split = the TAP-MAJOR terms-16 image of (9b);
 *     residual_kind 0 none | 1 SP map [N, Cout, H, W] | 2 channels-last float32;  out_kind 1 SP map | 2 channels-last float32;
 *     geometry 0 = chosen from the shape, 81 / 121 / 124 / 148 = a fixed tile geometry (8 x 32, 12 x 32, 24 x 16, 8 x 32 in 4 x 8 blocks); 326 (round 6) = 8 x 48
 *     tiles of 32 output channels on 12 wavefronts: never chosen by 0 (a layer's latency -9 ... -20 % on 25-row maps, the frame pipeline's throughput -1.6 %);
 *     range_flag (may be NULL): bit 0 is set when a value written to an SP map exceeded 65504 in magnitude.
 *     workspace: coalign_conv3x3_sp_workspace_bytes(...) bytes (0 = none needed for that shape), 16-byte aligned, ZERO-INITIALISED ONCE by the caller and
 *     then owned by the launches of ONE stream: the stream-K hand-over of tiles cut between workgroups (shapes whose whole tiles would leave the chip badly
 *     filled: 192 tiles on 256 CUs, 552 tiles = 2.2 rounds); every launch leaves its flag words zero again.  The cut is a pure function of the shape.
 *   Bit-identical to coalign_conv3x3_emu_ex(terms 16, tap-major) on the unpacked input. */
size_t coalign_sp_map_bytes(int N, int C, int H, int W);
int coalign_sp_pack(const float *x, int in_nhwc, void *y_sp, int N, int C, int H, int W, int32_t *range_flag, void *stream);
int coalign_sp_unpack(const void *x_sp, float *y, int out_nhwc, int N, int C, int H, int W, void *stream);
size_t coalign_conv3x3_sp_workspace_bytes(int N, int Cin, int Cout, int H, int W, int geometry);
int coalign_conv3x3_sp(const void *x_sp, const void *w_split, const float *bias, const void *residual, int residual_kind, void *y, int out_kind,
                       int N, int Cin, int Cout, int H, int W, int relu, int geometry, int32_t *range_flag, void *workspace, size_t workspace_bytes, void *stream);

/* (9e') Round 6: coalign_conv3x3_sp writing its map TWICE -- channels-last float32 in y_nhwc (what the fusion kernel, the exchange and the 1 x 1 skip
 * convolution read) and the SP map in y_sp (what the next stage's strided convolution (9f) reads): the last 3x3 convolution of a ResNet stage
 * (resblock.py:53-69, base_bev_backbone_resnet.py:95-119).  y_sp holds coalign_sp_pack(y_nhwc), bit for bit; everything else as coalign_conv3x3_sp. */
int coalign_conv3x3_sp_both(const void *x_sp, const void *w_split, const float *bias, const void *residual, int residual_kind, float *y_nhwc, void *y_sp,
                            int N, int Cin, int Cout, int H, int W, int relu, int geometry, int32_t *range_flag, void *workspace, size_t workspace_bytes, void *stream);

/* (9f) Round 6: the STRIDED 3x3 convolution (stride 2, pad 1, bias + ReLU: the first convolution of a ResNet stage, resblock.py:53-69 with stride 2 and
 * :150-174, base_bev_backbone_resnet.py:59-119) in the form of (9e): input already split, K loop = LDS-DMA + matrix instructions (csrc/conv3x3_sp_s2.hip).
 *   coalign_conv3x3_sp_s2: x_sp = SP map [N, Cin, H, W] -> y_sp = SP map [N, Cout, ceil(H/2), ceil(W/2)]; w_split = the TAP-MAJOR terms-16 image of (9b).
 *   coalign_conv3x3_sp_s2_sparse: the same layer reading the sparse canvas of (1b): rows_sp = coalign_sp_pack_rows of the feature rows, stamps / state as (9d);
 *     a stamp naming a row >= M_rows reads as empty.  Cin >= 32.
 *   coalign_sp_pack_rows: float32 feature rows [M_capacity, C] (C % 16 == 0) -> sp16 rows [M][C / 16][4 planes][8] fp16 (coalign_sp_rows_bytes = M * C * 4
 *     bytes, 16-byte aligned): a row's 16-byte group (c16, plane) = one matrix operand, fetched by one lane of an LDS-DMA instruction.  M_dev (may be NULL):
 *     the row count on the device; rows at and beyond it are not touched.
 *   Output (y, x) is bit-identical to output (2 y, 2 x) of coalign_conv3x3_sp on the same map (same operations in the same order); Cin % 16 == 0,
 *   Cout % 64 == 0; range_flag as (9e); no workspace. */
size_t coalign_sp_rows_bytes(int M, int C);
int coalign_sp_pack_rows(const float *rows, int M_capacity, const int32_t *M_dev, int C, void *rows_sp, int32_t *range_flag, void *stream);
int coalign_conv3x3_sp_s2(const void *x_sp, const void *w_split, const float *bias, void *y_sp, int N, int Cin, int Cout, int H, int W, int relu,
                          int32_t *range_flag, void *stream);
int coalign_conv3x3_sp_s2_sparse(const void *rows_sp, int M_rows, const void *stamps, const int32_t *state, const void *w_split, const float *bias, void *y_sp,
                                 int N, int Cin, int Cout, int H, int W, int relu, int32_t *range_flag, void *stream);

/* (9g) Round 6: (9f) carrying the block's 1 x 1 / stride-2 DOWN-SAMPLING convolution (resblock.py:165-174: `downsample`, BatchNorm folded, no ReLU; its input pixel (2y, 2x)
 * is the centre tap of the strided 3x3 convolution) as a tenth tap: ONE launch computes conv1 (SP map out) and the skip map y_skip [N, ceil(H/2), ceil(W/2), Cout]
 * channels-last float32 -- what (10) / (10b) computed in a launch of their own (20-30 us for 0.5 us of matrix work).
 *   w_skip: coalign_conv1x1_sp_weight_bytes(Cin, Cout) bytes, 16-byte aligned: [Cout / 64][Cin / 16][2 terms][2 channel halves][64 cout][8 cin] fp16 sp16 pairs of the
 *   per-output-channel scaled 1 x 1 weights, 16 zero bytes, [Cout] float32 2^-k_c, [Cout] float32 2^k_c (the centre-tap slice of the image of (9b): terms 16, tap-major).
 *   No bias on the skip (the caller adds its BatchNorm shift to the second convolution's bias, as (10) does); Cout <= 512.  conv1's output is that of (9f), bit for bit. */
size_t coalign_conv1x1_sp_weight_bytes(int Cin, int Cout);
int coalign_conv3x3_sp_s2_skip(const void *x_sp, const void *w_split, const float *bias, const void *w_skip, void *y_sp, float *y_skip, int N, int Cin, int Cout, int H, int W,
                               int relu, int32_t *range_flag, void *stream);
int coalign_conv3x3_sp_s2_skip_sparse(const void *rows_sp, int M_rows, const void *stamps, const int32_t *state, const void *w_split, const float *bias, const void *w_skip,
                                      void *y_sp, float *y_skip, int N, int Cin, int Cout, int H, int W, int relu, int32_t *range_flag, void *stream);

/* (9c) The Winograd F(2x2, 3x3) convolution of round 4 (measured, not adopted) is exported by the LABORATORY library only: include/coalign_amd_lab.h. */

/* Fill `n_words` 32-bit words at `p` (4-byte aligned) with `value`, as a kernel on `stream` (the per-frame counters of the post-processing
 * buffers, voxel_postprocessor.py:243-402's per-frame state; graph-capture safe, no library launch). */
int coalign_fill_words(void *p, size_t n_words, uint32_t value, void *stream);

/* (1b) Round 4: PillarVFE + PointPillarScatter as ONE launch with a SPARSE canvas (csrc/pillar_sparse.hip).  Replaces the same reference modules as (1)
 * (pillar_vfe.py:31-53,105-155, point_pillar_scatter.py:15-72) for callers that consume the canvas through (9d) / (10b) below.
 *   pillar_features [M_capacity, C] out: the feature rows (C <= 64, P <= 32; the linearised PFN evaluated exactly in fp32 on v_mfma_f32_32x32x2_f32).
 *   stamps: coalign_sparse_canvas_stamp_bytes(n_agents, ny, nx) bytes, 8-byte aligned, ZERO-INITIALISED ONCE by the caller and then owned by this
 *     sequence of calls: 64-bit words (frame tag << 32) | pillar row per cell, entered by atomicMax -- the larger row of a cell wins, the reference's rule.
 *   state: coalign_sparse_canvas_state_bytes() bytes (int32[528] since round 6; [2] before), zero-initialised once: state[0] = tag of the last completed call, the rest
 *     the launch's arrival counters (two levels: one counter for every workgroup cost 3.8 of the launch's 18 us).  A cell holds pillar row
 *     (stamp & 0xffffffff) of THIS frame iff (stamp >> 32) == state[0] after the call; nothing is cleared between frames.  The tag is 32 bits and stamps are
 *     ordered by (tag, row): the caller re-zeroes stamps and state before 2^32 - 1 calls have gone through one stamp map (coalign_amd/ops.py does after 2^31).
 *   M_dev (may be NULL): the pillar count on the device (int32), M_capacity then sizes the arrays.  No distance feature (with_distance configs use (1)).
 *   folded: coalign_pillar_folded_param_bytes() bytes (16-byte aligned) written ONCE per weight set by coalign_pillar_fold_params from the PFN layer's
 *     Linear weight [C, 10 or 7] (+ bias when there is no BatchNorm) and its eval BatchNorm tensors: the channel parameters in the form the kernel's pair
 *     loop uses (the three weight groups summed, BatchNorm folded to scale / shift, signs folded) -- the counterpart of the convolutions' split weight images. */
size_t coalign_sparse_canvas_stamp_bytes(int n_agents, int ny, int nx);
size_t coalign_sparse_canvas_state_bytes(void);
size_t coalign_pillar_folded_param_bytes(void);
int coalign_pillar_fold_params(const float *pfn_weight, const float *pfn_bias, const float *bn_weight, const float *bn_bias, const float *bn_mean,
                               const float *bn_var, float bn_eps, int C, int use_absolute_xyz, float *folded, void *stream);
int coalign_pillar_encode_sparse(const float *voxel_features, const int32_t *voxel_num_points, const int32_t *voxel_coords, int M_capacity,
                                 const int32_t *M_dev, int P, const float *folded, int C, int use_absolute_xyz, const double *voxel_size,
                                 const double *range_min, int n_agents, int ny, int nx, float *pillar_features, void *stamps, int32_t *state, void *stream);
/* (1c) Round 5: the same launch reading its inputs through a FRAME RECORD in device memory.  A captured HIP graph bakes the pointers of its launches; a
 * caller that replays one graph per frame on frames living at different addresses either copies every frame into the graph's buffers (20 MB per 5-agent
 * frame) or -- this entry -- writes 32 bytes: the kernel loads the three array pointers and the pillar count from `frame` when it starts.  The arrays
 * named by the record must stay valid and unchanged until the launch has completed; M_capacity sizes the grid and pillar_features ([M_capacity, C]);
 * rows at and beyond frame->M are never read.  Everything else as (1b). */
typedef struct coalign_pillar_frame {
    const float *voxel_features;          /* [M, P, 4] float32 (processed_lidar.voxel_features, pillar_vfe.py:105-155) */
    const int32_t *voxel_num_points;      /* [M] */
    const int32_t *voxel_coords;          /* [M, 4] (agent, z, y, x) */
    int32_t M;                            /* pillars of this frame, clamped to [0, M_capacity] by the kernel */
    int32_t reserved;
} coalign_pillar_frame;
int coalign_pillar_encode_sparse_frame(const coalign_pillar_frame *frame, int M_capacity, int P, const float *folded, int C, int use_absolute_xyz,
                                       const double *voxel_size, const double *range_min, int n_agents, int ny, int nx, float *pillar_features, void *stamps,
                                       int32_t *state, void *stream);
/* (9d) The strided 3x3 convolution of (9b) (stride 2, pad 1, bias + ReLU; resblock.py:150-174) reading the sparse canvas of (1b): feats [M, Cin], pixel
 * (n, y, x) of the logical [N, Hin, Win, Cin] input = feats row (stamp & 0xffffffff) where the cell's stamp carries state[0] and names a row < M_rows (the rows
 * behind feats: a stamp map shared with a later, larger frame can never index past the caller's array), else zero.
 * y: [N, Cout, ceil(Hin/2), ceil(Win/2)] NCHW, channels-last if out_nhwc == 1, an SP map (9e) if out_nhwc == 2 (terms 16).  terms in {2, 3, 16} with the tap-pair weight image of (9b). */
int coalign_conv3x3_emu_sparse(const float *feats, int M_rows, const void *stamps, const int32_t *state, const void *w_split, const float *bias, float *y, int N, int Cin,
                               int Cout, int Hin, int Win, int relu, int terms, int out_nhwc, int32_t *range_flag, void *stream);

/* ---------------------------------------------------------------------------------------------------------
 * (10) Pointwise layers of the BEV backbone as one GEMM launch each, bias (+ ReLU) fused, NCHW float32:
 *   up in {1, 2, 4}, in_stride = 1:  ConvTranspose2d(kernel = stride = up) + eval BatchNorm (folded) + ReLU of the up-sampling heads
 *        (opencood/models/sub_modules/base_bev_backbone_resnet.py:47-87, 121-138), written into channels
 *        [c_off, c_off + Cout) of a [N, Ctot, Hin * up, Win * up] tensor -- the concatenation of :134-138 without a copy;
 *   up = 1, in_stride = 2:           the 1 x 1 / stride-2 down-sampling convolution + BatchNorm of a ResNet stage's first block
 *        (opencood/models/sub_modules/resblock.py:53-69, 165-174), output [N, Ctot, ceil(Hin / 2), ceil(Win / 2)].
 * x [N, Cin, Hin, Win], Cin even and <= 256.
 * w [Cin, M_padded]: row ci holds the GEMM rows m = co * up * up + ky * up + kx, i.e. ConvTranspose2d's [Cin, Cout, k, k] weight
 *   as it is (M_padded = Cout * up * up) or, for up = 1, the transposed Conv2d weight zero-padded to a multiple of 32 columns.
 * bias [Cout].  y must be 16-byte aligned for up = 4, 8-byte for up = 2.
 */
int coalign_pointwise_conv(const float *x, const float *w, const float *bias, float *y, int N, int Cin, int Hin, int Win,
                           int in_stride, int Cout, int up, int M_padded, int Ctot, int c_off, int relu, void *stream);
/* in_nhwc bit 0: x is channels-last, [N, Hin, Win, Cin] (Cin % 4 == 0, 16-byte aligned).  in_nhwc bit 1 (round 4, up = 1 only): y is channels-last,
 * [N, Hp, Wp, Ctot] (Ctot % 4 == 0, c_off % 4 == 0, y and bias 16-byte aligned) -- the skip convolution whose result is the residual of a Winograd layer. */
int coalign_pointwise_conv_ex(const float *x, const float *w, const float *bias, float *y, int N, int Cin, int Hin, int Win, int in_stride,
                              int Cout, int up, int M_padded, int Ctot, int c_off, int relu, int in_nhwc, void *stream);
/* The same layers on the bf16 matrix cores by error-free 3-way operand splitting (fp32-width products, fp32 accumulation; the arithmetic of
 * coalign_conv3x3_emu_ex, used together with it).  Replaces the same reference modules as coalign_pointwise_conv: the up-sampling heads
 * (opencood/models/sub_modules/base_bev_backbone_resnet.py:47-87, 121-138) and the stride-2 skip convolutions (resblock.py:53-69, 165-174).
 * Cin % 16 == 0, Cin <= 256.  w_split: the pre-split weight image, coalign_pointwise_emu_weight_bytes(Cin, M_padded) bytes, 16-byte
 * aligned: uint4 [M_padded / 32][Cin / 16][3 terms][64 lanes]; lane l of (row tile, step, term) = term `term` (0: bf16(w), 1: bf16 of the
 * remainder, 2: bf16 of what is left, round-to-nearest-even) of W[k = 16 step + 8 (l / 32) + j][m = 32 tile + l % 32], j = 0..7, as 8
 * bf16 -- W being the [Cin][M_padded] matrix coalign_pointwise_conv takes.  Every other argument as coalign_pointwise_conv_ex. */
size_t coalign_pointwise_emu_weight_bytes(int Cin, int M_padded);
int coalign_pointwise_conv_emu(const float *x, const void *w_split, const float *bias, float *y, int N, int Cin, int Hin, int Win, int in_stride,
                               int Cout, int up, int M_padded, int Ctot, int c_off, int relu, int in_nhwc, void *stream);

/* (10b) The 1 x 1 / stride-2 skip convolution of (10) on the split-bf16 matrix cores reading the sparse canvas of (1b) (resblock.py:165-174).
 * y: [N, Cout, ceil(Hin/2), ceil(Win/2)] NCHW, or channels-last if out_nhwc != 0 (Cout % 4 == 0). */
int coalign_pointwise_conv_emu_sparse(const float *feats, int M_rows, const void *stamps, const int32_t *state, const void *w_split, const float *bias, float *y, int N,
                                      int Cin, int Hin, int Win, int Cout, int M_padded, int relu, int out_nhwc, void *stream);

/* (10c) Round 5: (10) on the split-bf16 matrix cores writing its channel slice [c_off, c_off + Cout) of an SP map (9e) of Ctot channels instead of float32 --
 * the three up-sampling heads of base_bev_backbone_resnet.py:121-138 hand the concatenated map to the shrink header's first 3 x 3 convolution
 * (downsample_conv.py:7-27) already split, which then runs on coalign_conv3x3_sp.  Same values, bit for bit, as coalign_sp_pack of (10)'s float32 result.
 * Cout, Ctot, c_off multiples of 16; M_padded == Cout * up * up; in_nhwc: 0 / 1 (input layout); range_flag as in (9e), may be NULL. */
int coalign_pointwise_conv_emu_sp(const float *x, const void *w_split, const float *bias, void *y_sp, int N, int Cin, int Hin, int Win, int in_stride, int Cout,
                                  int up, int M_padded, int Ctot, int c_off, int relu, int in_nhwc, int32_t *range_flag, void *stream);
/* (10e) Round 6: the merged cls / reg / dir 1 x 1 heads (point_pillar_baseline_multiscale.py:123-133, point_pillar.py: three nn.Conv2d(C, k, 1) on the shrink header's map,
 * their weights concatenated to M <= 32 rows) reading that map as an SP MAP (9e): y [N, M, H, W] float32 NCHW = W x + bias, no activation.  w_sp: the image of (9g) built from
 * the weights padded to 64 rows (coalign_conv1x1_sp_weight_bytes(Cin, 64) bytes).  One 16-byte load per lane and operand, no LDS, the sp16 arithmetic of the 3x3 layers. */
int coalign_heads_sp(const void *x_sp, const void *w_sp, const float *bias, float *y, int N, int Cin, int M, int H, int W, void *stream);
/* (10d) Round 6: (10c) for SEVERAL layers in ONE launch: the up-sampling heads of the backbone's scales (base_bev_backbone_resnet.py:121-138) are independent GEMMs on
 * one fused map each (2 200 ... 35 200 pixels), latency bound one after the other; as one launch their workgroups run side by side.  Layer i: x[i] float32
 * [N, Cin[i], Hin[i], Win[i]] (channels-last if in_nhwc[i]), split weight image w_split[i], bias[i]; writes channels [c_off[i], c_off[i] + Cout[i]) of the SP map
 * y_sp [N, Ctot, Hin[i] * up[i], Win[i] * up[i]] (one output size for all).  n_layers <= 4.  Bit-identical to n_layers calls of (10c). */
int coalign_pointwise_conv_emu_sp_multi(int n_layers, const void *const *x, const void *const *w_split, const void *const *bias, const int32_t *Cin, const int32_t *Hin,
                                        const int32_t *Win, const int32_t *Cout, const int32_t *up, const int32_t *c_off, const int32_t *in_nhwc, void *y_sp, int N, int Ctot,
                                        int relu, int32_t *range_flag, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* COALIGN_AMD_H */
