"""Pose algebra for the fusion step (host side, float64) -- SURVEY §8a row F.

The pose matrices are tiny (L x L 4x4 doubles per frame) and feed the warp kernel as six
doubles per agent, so they stay on the host in float64 exactly like the reference:

* ``x_to_world``                  <- opencood/utils/transformation_utils.py:263-306
* ``get_pairwise_transformation`` <- opencood/utils/transformation_utils.py:22-67
* ``normalize_pairwise_tfm``      <- opencood/utils/transformation_utils.py:69-91
* ``generate_noise``              <- opencood/utils/pose_utils.py:43-73
"""
from __future__ import annotations

from typing import Sequence

import numpy as np
import torch


def x_to_world(pose: Sequence[float]) -> np.ndarray:
    """``[x, y, z, roll, yaw, pitch]`` (degrees, CARLA convention) -> 4x4 T_world<-x (float64).

    Written as a product of elementary rotations  Rz(yaw) * Ry'(pitch) * Rx'(roll)  whose
    expansion is the closed form the reference spells out element by element."""
    x, y, z, roll, yaw, pitch = [float(v) for v in pose[:6]]
    cy, sy = np.cos(np.radians(yaw)), np.sin(np.radians(yaw))
    cr, sr = np.cos(np.radians(roll)), np.sin(np.radians(roll))
    cp, sp = np.cos(np.radians(pitch)), np.sin(np.radians(pitch))
    rz = np.array([[cy, -sy, 0.0], [sy, cy, 0.0], [0.0, 0.0, 1.0]])
    ry = np.array([[cp, 0.0, -sp], [0.0, 1.0, 0.0], [sp, 0.0, cp]])
    rx = np.array([[1.0, 0.0, 0.0], [0.0, cr, sr], [0.0, -sr, cr]])
    T = np.identity(4)
    T[:3, :3] = rz @ ry @ rx
    T[:3, 3] = (x, y, z)
    return T


def get_pairwise_transformation(lidar_poses: Sequence[Sequence[float]], max_cav: int,
                                proj_first: bool = False) -> np.ndarray:
    """``[L, L, 4, 4]`` float64; entry ``[i, j]`` is T_{j<-i} (agent-i coordinates -> agent-j).
    Unused slots stay identity.  With ``proj_first`` everything is identity."""
    out = np.tile(np.eye(4), (max_cav, max_cav, 1, 1))
    if proj_first:
        return out
    world = [x_to_world(p) for p in lidar_poses]
    for i, Ti in enumerate(world):
        for j, Tj in enumerate(world):
            if i != j:
                out[i, j] = np.linalg.solve(Tj, Ti)
    return out


def normalize_pairwise_tfm(pairwise_t_matrix: torch.Tensor, H: int, W: int, discrete_ratio: float,
                           downsample_rate: float = 1) -> torch.Tensor:
    """``[B, L, L, 4, 4]`` -> ``[B, L, L, 2, 3]`` affine in the normalised [-1, 1] coordinates that
    ``affine_grid`` uses.  Works on a copy (the caller's matrix is left untouched) and keeps the
    input dtype (float64 in the data pipeline)."""
    if pairwise_t_matrix.is_cuda and pairwise_t_matrix.dtype == torch.float64:
        # one launch instead of ~14 tiny element-wise kernels; same float64 operations in the same order (bit-identical)
        from . import ops
        return ops.normalize_pairwise(pairwise_t_matrix, H, W, downsample_rate * discrete_ratio * W, downsample_rate * discrete_ratio * H)
    rows = pairwise_t_matrix[..., :2, :]
    m = torch.stack((rows[..., 0], rows[..., 1], rows[..., 3]), dim=-1).clone()
    m[..., 0, 1] = m[..., 0, 1] * H / W
    m[..., 1, 0] = m[..., 1, 0] * W / H
    m[..., 0, 2] = m[..., 0, 2] / (downsample_rate * discrete_ratio * W) * 2
    m[..., 1, 2] = m[..., 1, 2] / (downsample_rate * discrete_ratio * H) * 2
    return m


def normalize_pairwise_np(pairwise_t_matrix: np.ndarray, H: int, W: int, discrete_ratio: float, downsample_rate: float = 1, out: np.ndarray = None) -> np.ndarray:
    """``normalize_pairwise_tfm`` on the HOST in numpy float64 (opencood/utils/transformation_utils.py:69-91): the same operations in the same order as the
    torch route and the device kernel, hence the same bits.  ``out`` ([..., 2, 3] float64, e.g. a view of a pinned staging buffer) is written in place."""
    m = np.asarray(pairwise_t_matrix, dtype=np.float64)
    if out is None:
        out = np.empty(m.shape[:-2] + (2, 3), dtype=np.float64)
    out[..., 0] = m[..., :2, 0]
    out[..., 1] = m[..., :2, 1]
    out[..., 2] = m[..., :2, 3]
    out[..., 0, 1] = out[..., 0, 1] * H / W
    out[..., 1, 0] = out[..., 1, 0] * W / H
    out[..., 0, 2] = out[..., 0, 2] / (downsample_rate * discrete_ratio * W) * 2
    out[..., 1, 2] = out[..., 1, 2] / (downsample_rate * discrete_ratio * H) * 2
    return out


def generate_noise_laplace(pos_b: float, rot_b: float, pos_mu: float = 0, rot_mu: float = 0, rng=np.random) -> np.ndarray:
    """Laplace localisation noise on (x, y, yaw) (opencood/utils/pose_utils.py:77-105): ``laplace(size=2)`` then ``laplace(size=1)``,
    the reference's draw order."""
    xy = rng.laplace(pos_mu, pos_b, size=2)
    yaw = rng.laplace(rot_mu, rot_b, size=1)
    return np.array([xy[0], xy[1], 0.0, 0.0, yaw[0], 0.0])


def generate_noise(pos_std: float, rot_std: float, pos_mean: float = 0, rot_mean: float = 0,
                   rng=np.random) -> np.ndarray:
    """Gaussian localisation noise on (x, y, yaw); draw order ``normal(size=2)`` then ``normal(size=1)``
    like the reference so a shared ``np.random.seed`` reproduces its sequence."""
    xy = rng.normal(pos_mean, pos_std, size=2)
    yaw = rng.normal(rot_mean, rot_std, size=1)
    return np.array([xy[0], xy[1], 0.0, 0.0, yaw[0], 0.0])
