"""Multi-agent feature fusion modules (SURVEY §8a rows G, H, H'), host side.

Class / function names and signatures follow opencood/models/fuse_modules/fusion_in_one.py
(``regroup`` :21-24, ``warp_feature`` :26-45, ``MaxFusion`` :47-89, ``AttFusion`` :91-136) and
opencood/models/sub_modules/torch_transformation_utils.py (``warp_affine_simple`` :322-331).  They own no
parameters; all arithmetic is the fused gfx950 kernel ``coalign_warp_fuse``.
"""
from __future__ import annotations

from typing import List, Sequence

import torch
import torch.nn as nn

from . import ops
from .encoder import host_ints


def regroup(x: torch.Tensor, record_len) -> List[torch.Tensor]:
    """Split the concatenated agent batch into per-frame views."""
    return list(torch.split(x, host_ints(record_len), dim=0))


def _ego_rows(affine: torch.Tensor, groups: Sequence[int]) -> torch.Tensor:
    """normalized_affine_matrix [B, L, L, 2, 3] -> theta [sum N, 2, 3]: row ``[b, 0, :N_b]`` of every frame
    (ego coordinates -> agent j; fusion_in_one.py:125-128)."""
    if affine.shape[0] == 1:
        return affine[0, 0, : groups[0]]
    return torch.cat([affine[b, 0, :n] for b, n in enumerate(groups)], dim=0)


def warp_affine_simple(src: torch.Tensor, M: torch.Tensor, dsize, mode="bilinear", padding_mode="zeros",
                       align_corners=False) -> torch.Tensor:
    """``F.grid_sample(src, F.affine_grid(M, ...).to(src))``, bilinear / zeros / align_corners=False
    (the extra keyword arguments are accepted and ignored exactly like the reference does)."""
    n = src.shape[0]
    return ops.warp_fuse(src, M, [1] * n if n <= 0 else _chunks(n), ops.FUSE_NONE, out_hw=(int(dsize[0]), int(dsize[1])))


def _chunks(n: int) -> List[int]:
    out = []
    while n > 0:
        out.append(min(8, n))
        n -= out[-1]
    return out


def warp_feature(x: torch.Tensor, record_len, pairwise_t_matrix: torch.Tensor) -> torch.Tensor:
    """Warp every agent of every frame into its ego frame; returns [sum N, C, H, W]."""
    groups = host_ints(record_len)
    return ops.warp_fuse(x, _ego_rows(pairwise_t_matrix, groups), groups, ops.FUSE_NONE)


def fuse_multiscale(xs: Sequence[torch.Tensor], record_len, affine: torch.Tensor, mode: int, rows=None):
    """All feature scales of a batch in ONE launch per frame when every map is channels-last (the route the split-bf16 backbone
    produces): -> list of fused maps [B, C_s, H_s, W_s] (channels-last), or None when the maps do not qualify (caller falls back
    to one ``coalign_warp_fuse`` launch per scale)."""
    if len(xs) > 3 or not all(ops.warp_fuse_nhwc_ok(x) for x in xs):
        return None
    groups = host_ints(record_len)
    if sum(groups) != xs[0].shape[0] or max(groups) > 8:
        return None
    outs, off = [], 0
    for b, n in enumerate(groups):
        theta = affine[b, 0, :n]
        r = None if rows is None else [int(v) for v in rows[off:off + n]]
        outs.append(ops.warp_fuse_nhwc([x[off:off + n] for x in xs], theta, mode, rows=r))
        off += n
    if len(outs) == 1:
        return outs[0]
    return [torch.cat([o[k] for o in outs], dim=0) for k in range(len(xs))]


class MaxFusion(nn.Module):
    def forward(self, x: torch.Tensor, record_len, pairwise_t_matrix: torch.Tensor, rows=None) -> torch.Tensor:
        """``rows`` (not in the reference): row of ``x`` holding logical agent i, for agent-sharded callers."""
        groups = host_ints(record_len)
        return ops.warp_fuse(x, _ego_rows(pairwise_t_matrix, groups), groups, ops.FUSE_MAX, rows=rows)


class AttFusion(nn.Module):
    def __init__(self, feature_dims: int):
        super().__init__()
        self.feature_dims = feature_dims

    def forward(self, xx: torch.Tensor, record_len, normalized_affine_matrix: torch.Tensor, rows=None) -> torch.Tensor:
        groups = host_ints(record_len)
        # ScaledDotProductAttention divides by sqrt(feat_dim) of the CONFIG (att_fuse.py:36-44), the kernel by sqrt(C) of the tensor:
        # the two agree for every shipped yaml; a config where they differ would silently diverge from the reference's checkpoints
        if self.feature_dims != xx.shape[1]:
            # not the case in any shipped yaml.  The kernel's scores are <X0, Xj> / sqrt(C); feeding s X with s = (C / feat_dim)^(1/4) turns them
            # into <X0, Xj> / sqrt(feat_dim) and scales the output by s, which is divided out again (the warp is linear): the reference's
            # result to float32 rounding (one extra multiply and divide per element), instead of raising
            s = (xx.shape[1] / float(self.feature_dims)) ** 0.25
            return ops.warp_fuse(xx * s, _ego_rows(normalized_affine_matrix, groups), groups, ops.FUSE_ATT, rows=rows) / s
        return ops.warp_fuse(xx, _ego_rows(normalized_affine_matrix, groups), groups, ops.FUSE_ATT, rows=rows)
