/* Entry points of the LABORATORY library only (coalign_amd/lib/libcoalign_hip_lab.so = the product sources + -DCOALIGN_LAB + the kernels below): variants that were
 * measured against the product's and not adopted.  They stay buildable and tested (tests/test_round4_gpu.py through coalign_amd.hip.lab_lib()); the product
 * library libcoalign_hip.so does not contain them.  Same conventions as include/coalign_amd.h. */
#ifndef COALIGN_AMD_LAB_H
#define COALIGN_AMD_LAB_H

#include "coalign_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* (9c) Round 4: the stride-1 3x3 convolutions as Winograd F(2x2, 3x3) on the bf16 matrix cores -- 16 instead of 36 products per 2 x 2 outputs and
 * (cin, cout), fp32 operands by the same 3-way error-free bf16 split, fp32 accumulation (csrc/conv3x3_wino.hip).  Same layers as (9b):
 * BasicBlock.forward opencood/models/sub_modules/resblock.py:53-69 and DoubleConv downsample_conv.py:7-27.
 *   x, residual (may be NULL), y: CHANNELS-LAST, [N][H][W][C] float32, 16-byte aligned; Cin % 16 == 0, Cout % 64 == 0, any H, W.
 *   u_split: coalign_conv3x3_wino_weight_bytes(Cin, Cout) bytes: the transformed weights U = G g G^T (float64 on the host, G = [1 0 0; .5 .5 .5;
 *   .5 -.5 .5; 0 0 1]) as three bf16 terms in the order the kernel's wavefronts load them:
 *   [Cout / 64][Cin / 16][h 2][wave 8 = (c, i)][jj 2][term 3][lane 64][8] bf16 = term of U[i][2 h + jj][64 g + 32 c + lane % 32][16 k + 8 (lane / 32) + e].
 *   tile_block_w: 0 = chosen from W, 8 = blocks of 8 x 8 Winograd tiles, 16 = 4 x 16 tiles per workgroup.
 * No workspace.  y = relu?(conv3x3(x, g) + bias (+ residual)); against a float64 convolution the error is of the size of the direct
 * split-bf16 kernel's (measured: tests/test_round4_gpu.py). */
size_t coalign_conv3x3_wino_weight_bytes(int Cin, int Cout);
int coalign_conv3x3_wino(const float *x, const void *u_split, const float *bias, const float *residual, float *y, int N, int Cin, int Cout,
                         int H, int W, int relu, int tile_block_w, void *stream);

#ifdef __cplusplus
}
#endif

#endif
