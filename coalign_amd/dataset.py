"""Batch-dict producer of the hot path: what ``IntermediateFusionDataset.__getitem__`` +
``collate_batch_test`` hand to the model and the post-processor at test time
(opencood/data_utils/datasets/intermediate_fusion_dataset.py:236-606), from *in-memory* per-cav records -- the disk readers
(yaml / pcd / OPV2V folder walking, basedataset) stay outside.

A scenario is the reference's ``base_data_dict``: an ordered mapping ``cav_id -> {'ego': bool, 'params': {'lidar_pose': [6],
'vehicles': {id: {'location', 'angle', 'extent'[, 'center']}}}, 'lidar_np': [n, 4] float32}`` with the ego first.  The batcher

* adds pose noise (``opencood/utils/pose_utils.py:10-74``), drops cavs beyond ``comm_range``, optionally refines the poses by box
  alignment from cached stage-1 results (``coalign_amd.box_align``, hook :301-328),
* builds ``pairwise_t_matrix`` (``transformation_utils.get_pairwise_transformation``), the ground-truth boxes in the clean ego frame
  (``base_postprocessor.generate_object_center`` -> ``box_utils.project_world_objects``), de-duplicated by object id and padded to
  ``max_num``,
* voxelises ALL cavs of the frame in one ``coalign_voxelize`` call (shuffle -> ego mask -> voxels, :96-117) -- the reference
  runs spconv once per cav on the CPU and concatenates in ``collate_batch`` --
* and returns the ``{'ego': {...}}`` batch with the keys the model / post-processor / evaluation read.  Training-only entries
  (``label_dict`` anchor targets, single-view supervision, camera inputs, knowledge-distillation lidar) are not produced.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np
import torch

from . import box_align
from .pose import generate_noise, generate_noise_laplace, get_pairwise_transformation, x_to_world
from .postprocess import build_postprocessor
from .preprocess import build_preprocessor


def x1_to_x2(x1: Sequence[float], x2: Sequence[float]) -> np.ndarray:
    """T_x2_x1 (transformation_utils.py:309-333)."""
    return np.dot(np.linalg.inv(x_to_world(x2)), x_to_world(x1))


def create_bbx(extent: Sequence[float]) -> np.ndarray:
    """8 corners of a box with the given half sizes in its own frame (box_utils.py:424-448)."""
    ex, ey, ez = extent
    return np.array([[ex, -ey, -ez], [ex, ey, -ez], [-ex, ey, -ez], [-ex, -ey, -ez],
                     [ex, -ey, ez], [ex, ey, ez], [-ex, ey, ez], [-ex, -ey, ez]], dtype=np.float64)


def project_world_objects(object_dict: dict, lidar_pose: Sequence[float], lidar_range: Sequence[float], order: str) -> "OrderedDict[int, np.ndarray]":
    """Objects annotated in the world frame -> (x, y, z, dims, yaw) [1, 7] in the given lidar frame; kept only when all 8
    corners are inside ``lidar_range`` (box_utils.py:451-513)."""
    out: "OrderedDict[int, np.ndarray]" = OrderedDict()
    lo, hi = np.asarray(lidar_range[0:3], dtype=np.float64), np.asarray(lidar_range[3:6], dtype=np.float64)
    for object_id, content in object_dict.items():
        centre = content.get("center", [0, 0, 0])
        pose = [content["location"][0] + centre[0], content["location"][1] + centre[1], content["location"][2] + centre[2],
                content["angle"][0], content["angle"][1], content["angle"][2]]
        corners = np.c_[create_bbx(content["extent"]), np.ones(8)]
        corners = np.dot(x1_to_x2(pose, lidar_pose), corners.T).T[None, :, :3]
        box = box_align.corner_to_center(corners, order)
        if np.all((corners >= lo) & (corners <= hi)):
            out[object_id] = box
    return out


def generate_object_center(cav_contents: Sequence[dict], reference_lidar_pose: Sequence[float], post_params: dict, train: bool):
    """base_postprocessor.generate_object_center (:201-253): (boxes [max_num, 7], mask [max_num], ids)."""
    objects: dict = {}
    for cav in cav_contents:
        objects.update(cav["params"]["vehicles"])
    filter_range = post_params["anchor_args"]["cav_lidar_range"] if train else post_params["gt_range"]
    kept = project_world_objects(objects, reference_lidar_pose, filter_range, post_params["order"])
    boxes = np.zeros((post_params["max_num"], 7))
    mask = np.zeros(post_params["max_num"])
    for i, box in enumerate(kept.values()):
        boxes[i] = box[0]
        mask[i] = 1
    return boxes, mask, list(kept.keys())


def add_noise_data_dict(data_dict: dict, noise_setting: dict, rng: Optional[np.random.RandomState] = None) -> dict:
    """pose_utils.add_noise_data_dict (:10-42): ``lidar_pose_clean`` = the annotated pose, ``lidar_pose`` += Gaussian (x, y, yaw)."""
    for cav in data_dict.values():
        cav["params"]["lidar_pose_clean"] = cav["params"]["lidar_pose"]
        if noise_setting["add_noise"]:
            a = noise_setting["args"]
            draw = generate_noise_laplace if a.get("laplace", False) is True else generate_noise      # the same key names serve both (pose_utils.py:19-34)
            cav["params"]["lidar_pose"] = cav["params"]["lidar_pose"] + draw(a["pos_std"], a["rot_std"], a["pos_mean"], a["rot_mean"], rng=rng if rng is not None else np.random)
    return data_dict


class IntermediateFusionBatcher:
    """``IntermediateFusionDataset`` reduced to what it does per frame at test time.  ``preprocessor`` defaults to the device
    voxeliser; any object with ``preprocess_clouds(list_of_clouds, ego_filter=...)`` can be injected (the CPU tests do)."""

    def __init__(self, hypes: dict, train: bool = False, device="cuda:0", preprocessor=None, stage1_result: Optional[dict] = None,
                 shuffle: Optional[Callable[[np.ndarray], np.ndarray]] = None):
        self.params = hypes
        self.train = train
        self.device = torch.device(device)
        self.max_cav = hypes["train_params"]["max_cav"]
        # proj_first: every cav's cloud is projected into the ego frame BEFORE it is voxelised and the pairwise matrices are identity
        # (intermediate_fusion_dataset.py:43-44, 104; transformation_utils.py:43-49)
        self.proj_first = bool(hypes.get("fusion", {}).get("args", {}).get("proj_first", False))
        self.post_processor = build_postprocessor(hypes["postprocess"], train)
        self.pre_processor = preprocessor if preprocessor is not None else build_preprocessor(hypes["preprocess"], train, device)
        self.anchor_box = self.post_processor.generate_anchor_box()
        self.anchor_box_torch = torch.from_numpy(self.anchor_box)
        self.box_align_args = hypes.get("box_align", {}).get("args") if "box_align" in hypes else None
        self.stage1_result = stage1_result
        self.shuffle = shuffle if shuffle is not None else (lambda pts: pts[np.random.permutation(pts.shape[0])])

    # ------------------------------------------------------------------------------------------ one frame
    def get_item(self, base_data_dict: "OrderedDict", idx=0, rng: Optional[np.random.RandomState] = None) -> dict:
        # The reference's retrieve_base_data hands every __getitem__ a FRESH dict (basedataset.py), so noise, box-aligned poses
        # and the communication-range cut never leak into the scenario.  Same here: per-cav records and their ``params`` are
        # copied (poses by value; point clouds and annotations are shared read-only), the caller's scenario is left untouched.
        fresh = OrderedDict()
        for cav_id, cav in base_data_dict.items():
            rec = dict(cav)
            rec["params"] = dict(cav["params"])
            rec["params"]["lidar_pose"] = list(cav["params"]["lidar_pose"])
            fresh[cav_id] = rec
        data = add_noise_data_dict(fresh, self.params["noise_setting"], rng)
        first = next(iter(data))
        if not data[first]["ego"]:
            raise ValueError("the first element of the scenario must be the ego")
        ego = data[first]
        ego_pose = ego["params"]["lidar_pose"]
        cav_ids, poses, poses_clean = [], [], []
        for cav_id, cav in list(data.items()):
            p = cav["params"]["lidar_pose"]
            if math.sqrt((p[0] - ego_pose[0]) ** 2 + (p[1] - ego_pose[1]) ** 2) > self.params["comm_range"]:
                data.pop(cav_id)                     # out of communication range
                continue
            cav_ids.append(cav_id); poses.append(p); poses_clean.append(cav["params"]["lidar_pose_clean"])

        # pose correction from the cached stage-1 detections (they cover every cav, in range or not)
        if self.box_align_args is not None and self.stage1_result is not None and str(idx) in self.stage1_result and self.stage1_result[str(idx)] is not None:
            s1 = self.stage1_result[str(idx)]
            where = [s1["cav_id_list"].index(c) for c in cav_ids]
            corners = [np.array(s1["pred_corner3d_np_list"][w], dtype=np.float64) for w in where]
            unc = [np.array(s1["uncertainty_np_list"][w], dtype=np.float64) for w in where]
            if sum(len(c) for c in corners) != 0:
                cur = np.array(poses)
                cur[:, [0, 1, 4]] = box_align.box_alignment_relative_sample_np(corners, cur, uncertainty_list=unc, device=self.device, **self.box_align_args)
                for i, cav_id in enumerate(cav_ids):
                    poses[i] = cur[i].tolist()
                    data[cav_id]["params"]["lidar_pose"] = cur[i].tolist()

        pairwise = get_pairwise_transformation([data[c]["params"]["lidar_pose"] for c in cav_ids], self.max_cav, self.proj_first)
        ego_clean = ego["params"]["lidar_pose_clean"]
        clouds, object_stack, object_ids = [], [], []
        for cav_id in cav_ids:
            cav = data[cav_id]
            pts = self.shuffle(cav["lidar_np"])                             # mask_ego_points rides in the voxeliser call ...
            if self.proj_first:
                # ... unless the cloud moves first: the reference masks the cav's own body in the SENSOR frame, then projects with the
                # float32 T_ego<-cav of the (noisy / corrected) poses (get_item_single_car :84-105, box_utils.py:893-921)
                pts = pts[~((pts[:, 0] >= -1.95) & (pts[:, 0] <= 2.95) & (pts[:, 1] >= -1.1) & (pts[:, 1] <= 1.1))].copy()
                T = torch.from_numpy(x1_to_x2(cav["params"]["lidar_pose"], data[first]["params"]["lidar_pose"])).float()
                homo = torch.nn.functional.pad(torch.from_numpy(np.ascontiguousarray(pts[:, :3])).float(), (0, 1), mode="constant", value=1)
                pts[:, :3] = torch.einsum("ik, jk->ij", homo, T)[:, :3].numpy()
            clouds.append(pts)
            boxes, mask, ids = generate_object_center([cav], ego_clean, self.params["postprocess"], self.train)
            object_stack.append(boxes[mask == 1])
            object_ids += ids
        unique = [object_ids.index(x) for x in set(object_ids)]                # repeated objects: first occurrence
        stack = np.vstack(object_stack)[unique]
        max_num = self.params["postprocess"]["max_num"]
        centre, mask = np.zeros((max_num, 7)), np.zeros(max_num)
        centre[: stack.shape[0]] = stack
        mask[: stack.shape[0]] = 1
        lidar = self.pre_processor.preprocess_clouds(clouds, ego_filter=not self.proj_first)
        return {"ego": {"object_bbx_center": centre, "object_bbx_mask": mask, "object_ids": [object_ids[i] for i in unique],
                        "anchor_box": self.anchor_box, "processed_lidar": lidar, "cav_num": len(cav_ids), "pairwise_t_matrix": pairwise,
                        "lidar_poses_clean": np.array(poses_clean).reshape(-1, 6), "lidar_poses": np.array(poses).reshape(-1, 6),
                        "sample_idx": idx, "cav_id_list": cav_ids}}

    # ------------------------------------------------------------------------------------------ batch of one (test time)
    def collate_batch_test(self, batch: List[dict]) -> dict:
        if len(batch) != 1:
            raise ValueError("batch size 1 is required during testing")
        ego = batch[0]["ego"]
        lidar = ego["processed_lidar"]
        eye = torch.from_numpy(np.identity(4)).float()
        return {"ego": {"object_bbx_center": torch.from_numpy(ego["object_bbx_center"][None]),
                        "object_bbx_mask": torch.from_numpy(ego["object_bbx_mask"][None]),
                        "processed_lidar": {k: lidar[k] for k in ("voxel_features", "voxel_coords", "voxel_num_points")},
                        "record_len": torch.from_numpy(np.array([ego["cav_num"]], dtype=int)),
                        "object_ids": ego["object_ids"],
                        "pairwise_t_matrix": torch.from_numpy(np.array([ego["pairwise_t_matrix"]])),
                        "lidar_pose_clean": torch.from_numpy(ego["lidar_poses_clean"]), "lidar_pose": torch.from_numpy(ego["lidar_poses"]),
                        "anchor_box": self.anchor_box_torch, "transformation_matrix": eye, "transformation_matrix_clean": eye.clone(),
                        "sample_idx": ego["sample_idx"], "cav_id_list": ego["cav_id_list"]}}

    def __call__(self, base_data_dict: "OrderedDict", idx=0, rng: Optional[np.random.RandomState] = None) -> dict:
        return self.collate_batch_test([self.get_item(base_data_dict, idx, rng)])

    def post_process(self, data_dict: dict, output_dict: dict):
        """(pred boxes, scores, ground-truth boxes) like the dataset's ``post_process`` (:585-606)."""
        boxes, scores = self.post_processor.post_process(data_dict, output_dict)
        return boxes, scores, self.post_processor.generate_gt_bbx(data_dict)
