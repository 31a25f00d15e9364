/* TEST INFRASTRUCTURE ONLY -- CPU restatement of the points -> pillars voxeliser.
 *
 * The reference calls a third-party dependency here that is NOT in /root/reference and not installed in this image:
 * spconv (v1.x `spconv.utils.VoxelGeneratorV2` -> points_to_voxel_3d_np, or v2.x `Point2VoxelCPU3d.point_to_voxel`;
 * opencood/data_utils/pre_processor/sp_voxel_preprocessor.py:22-85 accepts either, no version is pinned by the repo).
 * Both published implementations are the same sequential loop, restated below:  PARITY UNPINNED (no spconv to run, the
 * reference holds no golden vectors for it); anchored on the call site's contract -- float32 [N, 4] points in,
 * voxels [M, max_points, 4] / coordinates [M, 3] (z, y, x) / num_points [M] out.
 *
 *   for each point, in order:  c_j = floor((p_j - range_min_j) / voxel_j)  (float32), reject unless 0 <= c_j < grid_j;
 *   first point of a cell opens voxel number `voxel_num` unless max_voxels are open already (then the point is skipped,
 *   later points of already-open cells are still taken); a voxel keeps its first max_points points.
 */
#include <math.h>
#include <string.h>

int oracle_points_to_voxel(const float *pts, int n, const float *vsize, const float *range, const int *grid /* x y z */,
                           int max_points, int max_voxels, float *voxels, int *coors, int *num, int *cell_to_voxel) {
    const long cells = (long)grid[0] * grid[1] * grid[2];
    for (long i = 0; i < cells; ++i) cell_to_voxel[i] = -1;
    int voxel_num = 0;
    for (int i = 0; i < n; ++i) {
        int c[3], failed = 0;
        for (int j = 0; j < 3; ++j) {
            const float q = floorf((pts[4 * i + j] - range[j]) / vsize[j]);
            if (!(q >= 0.0f && q < (float)grid[j])) { failed = 1; break; }
            c[j] = (int)q;
        }
        if (failed) continue;
        const long cell = ((long)c[2] * grid[1] + c[1]) * grid[0] + c[0];
        int v = cell_to_voxel[cell];
        if (v == -1) {
            if (voxel_num >= max_voxels) continue;
            v = voxel_num++;
            cell_to_voxel[cell] = v;
            coors[3 * v + 0] = c[2]; coors[3 * v + 1] = c[1]; coors[3 * v + 2] = c[0];
        }
        if (num[v] < max_points) {
            memcpy(voxels + ((long)v * max_points + num[v]) * 4, pts + 4 * i, 4 * sizeof(float));
            num[v] += 1;
        }
    }
    return voxel_num;
}
