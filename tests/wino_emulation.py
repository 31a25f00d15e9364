"""Host-side walk through the data flow of ``csrc/conv3x3_wino.hip`` with the kernel's own index formulas (numpy, float64 arithmetic on the
bf16 terms): the operand image of ``ops.pack_conv3x3_wino_weight`` is read the way the wavefronts read it, the raw patch / V buffer / exchange
layouts are addressed the way the producer, consumer and epilogue address them.  It pins the LAYOUT and TRANSFORM logic on the CPU (the
matrix instruction's lane maps are taken from the CDNA4 guide); the GPU tests pin the kernel itself.  Test infrastructure only."""
from __future__ import annotations

import numpy as np
import torch


def _bf16_terms_to_f64(img_u8: torch.Tensor) -> np.ndarray:
    """uint8 image -> float64 values of the bf16 entries, flat."""
    return img_u8[:-16].view(torch.bfloat16).double().numpy()      # (the image ends in 16 zero bytes: the kernel's zero-padding source)


def emulate(x_nhwc: np.ndarray, u_img: torch.Tensor, bias: np.ndarray, residual, relu: bool, tbw: int) -> np.ndarray:
    N, H, W, Cin = x_nhwc.shape
    Cout = bias.shape[0]
    K, groups = Cin // 16, Cout // 64
    TBH = 64 // tbw
    PR, PC = 2 * TBH + 2, 2 * tbw + 2
    PCH = 24 if tbw == 16 else 12
    pitch = H + (1 if H % 2 else 2)
    tile_rows, tile_cols = N * pitch // 2, (W + 1) // 2
    blocks_x = (tile_cols + tbw - 1) // tbw
    blocks = blocks_x * ((tile_rows + TBH - 1) // TBH)
    U = _bf16_terms_to_f64(u_img).reshape(groups, K, 2, 8, 2, 3, 64, 8)          # [g][k][h][wave][jj][term][lane][e]
    y = np.zeros((N, H, W, Cout))
    lanes = np.arange(64)
    for unit in range(blocks * groups):
        cg, blk = unit % groups, unit // groups
        by, bx = blk // blocks_x, blk % blocks_x
        ty0, tx0 = by * TBH, bx * tbw
        acc = np.zeros((8, 4, 2, 32, 32))                                        # [wave][position j][tb][cout row m][tile col n]
        for k in range(K):
            # raw patch in the LDS layout [q][parity][row][PCH][4]
            raw = np.zeros((4, 2, PR, PCH, 4))
            for e in range(PR * PC * 4):
                pi, q = e >> 2, e & 3
                pr, pc = pi // PC, pi % PC
                s, xg = 2 * ty0 - 1 + pr, 2 * tx0 - 1 + pc
                n = s // pitch if s >= 0 else 0
                r = s - n * pitch
                ok = s >= 0 and n < N and r < H and 0 <= xg < W
                if ok:
                    raw[q, pc & 1, pr, pc >> 1] = x_nhwc[n, r, xg, 16 * k + 4 * q: 16 * k + 4 * q + 4]
            for h in range(2):
                # producers: V buffer [jj][i][g][tile][8] (terms summed: the split is exact)
                V = np.zeros((2, 4, 2, 64, 8))
                for wave in range(8):
                    wi, wc = wave & 3, wave >> 2
                    ra = 0 if wi == 0 else (2 if wi == 2 else 1)
                    rb = 2 if wi in (0, 1) else (1 if wi == 2 else 3)
                    sg = 1.0 if wi == 1 else -1.0
                    for lane in range(64):
                        pty, ptx = lane // tbw, lane % tbw
                        for q2 in range(2):
                            q = 2 * wc + q2
                            R = []
                            for ci in range(3):
                                c = h + ci
                                da = raw[q, c & 1, 2 * pty + ra, ptx + (c >> 1)]
                                db = raw[q, c & 1, 2 * pty + rb, ptx + (c >> 1)]
                                R.append(da + sg * db)
                            if h == 0:
                                p0, p1 = R[0] - R[2], R[1] + R[2]
                            else:
                                p0, p1 = R[1] - R[0], R[0] - R[2]
                            V[0, wi, wc, lane, 4 * q2: 4 * q2 + 4] = p0
                            V[1, wi, wc, lane, 4 * q2: 4 * q2 + 4] = p1
                # consumers
                for wave in range(8):
                    wi = wave & 3
                    for jj in range(2):
                        a_op = U[cg, k, h, wave, jj].sum(0)                       # [lane][8]: A[m = lane % 32][kk = 8 (lane // 32) + e]
                        A = np.zeros((32, 16))
                        A[lanes % 32, :][:, :] = 0
                        for lane in range(64):
                            A[lane % 32, 8 * (lane // 32): 8 * (lane // 32) + 8] = a_op[lane]
                        for tb in range(2):
                            B = np.zeros((16, 32))                                # B[kk][n]: lane reads V[jj][wi][lane >> 5][32 tb + (lane & 31)]
                            for lane in range(64):
                                B[8 * (lane >> 5): 8 * (lane >> 5) + 8, lane & 31] = V[jj, wi, lane >> 5, 32 * tb + (lane & 31)]
                            acc[wave, 2 * h + jj, tb] += A @ B
        # epilogue: column half inside the wavefront, exchange [i][b][tile][cout], row half across the wavefronts
        Z = np.zeros((4, 2, 64, 64))
        for wave in range(8):
            wi, wc = wave & 3, wave >> 2
            for tb in range(2):
                z0 = acc[wave, 0, tb] + acc[wave, 1, tb] + acc[wave, 2, tb]
                z1 = acc[wave, 1, tb] - acc[wave, 2, tb] - acc[wave, 3, tb]
                # D[m][n]: cout = 32 wc + m, tile = 32 tb + n
                Z[wi, 0, 32 * tb: 32 * tb + 32, 32 * wc: 32 * wc + 32] = z0.T
                Z[wi, 1, 32 * tb: 32 * tb + 32, 32 * wc: 32 * wc + 32] = z1.T
        for oa in range(2):
            for ob in range(2):
                for tile in range(64):
                    ty, tx = tile // tbw, tile % tbw
                    s, ox = 2 * (ty0 + ty) + oa, 2 * (tx0 + tx) + ob
                    n, oy = s // pitch, s % pitch
                    if n < N and oy < H and ox < W:
                        o = Z[0, ob, tile] + Z[1, ob, tile] + Z[2, ob, tile] if oa == 0 else Z[1, ob, tile] - Z[2, ob, tile] - Z[3, ob, tile]
                        o = o + bias[64 * cg: 64 * cg + 64]
                        if residual is not None:
                            o = o + residual[n, oy, ox, 64 * cg: 64 * cg + 64]
                        y[n, oy, ox, 64 * cg: 64 * cg + 64] = np.maximum(o, 0) if relu else o
    return y
