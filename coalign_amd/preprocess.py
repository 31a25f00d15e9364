"""Points -> pillars pre-processing on the device.

``SpVoxelPreprocessor`` keeps the reference's constructor, ``preprocess(pcd)`` and ``collate_batch(batch)``
(opencood/data_utils/pre_processor/sp_voxel_preprocessor.py:18-174) -- there spconv's CPU voxel generator runs once per
cav inside the DataLoader workers; here one ``coalign_voxelize`` call turns the concatenated clouds of a whole frame into
the collated ``voxel_features / voxel_coords / voxel_num_points`` tensors the detector consumes.  ``mask_points_by_range`` and
``mask_ego_points`` (opencood/utils/pcd_utils.py:41-88) are folded into the same call as flags.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Union

import numpy as np
import torch

from . import ops


def _as_device_points(pcd, device) -> torch.Tensor:
    t = torch.from_numpy(np.ascontiguousarray(pcd, dtype=np.float32)) if isinstance(pcd, np.ndarray) else pcd
    return t.to(device=device, dtype=torch.float32)


class SpVoxelPreprocessor:
    def __init__(self, preprocess_params: dict, train: bool, device: Union[str, torch.device] = "cuda:0"):
        self.params = preprocess_params
        self.train = train
        self.device = torch.device(device)
        self.lidar_range = self.params["cav_lidar_range"]
        self.voxel_size = self.params["args"]["voxel_size"]
        self.max_points_per_voxel = self.params["args"]["max_points_per_voxel"]
        self.max_voxels = self.params["args"]["max_voxel_train" if train else "max_voxel_test"]
        grid = (np.array(self.lidar_range[3:6]) - np.array(self.lidar_range[0:3])) / np.array(self.voxel_size)
        self.grid_size = np.round(grid).astype(np.int64)

    # ------------------------------------------------------------------------------------------ whole frame, one launch
    def preprocess_clouds(self, clouds: Sequence, ego_filter: bool = False, filter_range: Optional[Sequence[float]] = None) -> Dict[str, torch.Tensor]:
        """The per-cav ``preprocess`` calls plus ``collate_batch`` of one frame in a single pass: ``clouds`` is a list of
        [n_i, 4] arrays / tensors (already shuffled or not -- the result follows their point order).  Returns the collated
        dictionary (coords [M, 4] = (cloud, z, y, x)) plus ``voxel_counts`` [n_clouds] (voxels per cloud, host ints)."""
        pts = [_as_device_points(c, self.device) for c in clouds]
        offsets = np.concatenate([[0], np.cumsum([p.shape[0] for p in pts])]).astype(np.int64)
        allpts = torch.cat(pts) if len(pts) > 1 else pts[0]
        voxels, coords, num, counts = ops.voxelize(allpts, offsets.tolist(), self.voxel_size, self.lidar_range, self.max_points_per_voxel,
                                                   self.max_voxels, ego_filter=ego_filter, filter_range=filter_range)
        host = counts.cpu().tolist()          # the one device -> host read: the voxel count sizes every later launch
        m = host[-1]
        return {"voxel_features": voxels[:m], "voxel_coords": coords[:m], "voxel_num_points": num[:m], "voxel_counts": host[:-1]}

    # ------------------------------------------------------------------------------------------ reference-shaped API
    def preprocess(self, pcd_np) -> Dict[str, torch.Tensor]:
        """One cloud -> ``voxel_features`` [M, P, 4], ``voxel_coords`` [M, 3] (z, y, x), ``voxel_num_points`` [M]
        (sp_voxel_preprocessor.py:62-85), as device tensors."""
        out = self.preprocess_clouds([pcd_np])
        return {"voxel_features": out["voxel_features"], "voxel_coords": out["voxel_coords"][:, 1:], "voxel_num_points": out["voxel_num_points"]}

    def collate_batch(self, batch: Union[List[dict], Dict[str, list]]) -> Dict[str, torch.Tensor]:
        """sp_voxel_preprocessor.py:87-174: concatenate per-cloud outputs, prefixing the coords with the cloud index."""
        if isinstance(batch, list):
            batch = {k: [b[k] for b in batch] for k in ("voxel_features", "voxel_coords", "voxel_num_points")}
        elif not isinstance(batch, dict):
            raise TypeError("batch has to be a list or a dictionary")
        as_t = lambda a: torch.from_numpy(a) if isinstance(a, np.ndarray) else a
        coords = [torch.nn.functional.pad(as_t(c), (1, 0), mode="constant", value=i) for i, c in enumerate(batch["voxel_coords"])]
        return {"voxel_features": torch.cat([as_t(v) for v in batch["voxel_features"]]),
                "voxel_coords": torch.cat(coords),
                "voxel_num_points": torch.cat([as_t(n) for n in batch["voxel_num_points"]])}


def build_preprocessor(preprocess_cfg: dict, train: bool, device: Union[str, torch.device] = "cuda:0") -> SpVoxelPreprocessor:
    """opencood/data_utils/pre_processor/__init__.py:18-33 for the one method on the hot path."""
    name = preprocess_cfg["core_method"]
    if name != "SpVoxelPreprocessor":
        raise NotImplementedError(f"{name}: only SpVoxelPreprocessor feeds the PointPillar hot path")
    return SpVoxelPreprocessor(preprocess_cfg, train, device)
