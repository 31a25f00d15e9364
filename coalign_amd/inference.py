"""Per-frame inference drivers: the reference's
``inference_late_fusion`` / ``inference_early_fusion`` / ``inference_intermediate_fusion``
(opencood/tools/inference_utils.py:17-46, 122-173) with the dataset object reduced to what they use from it --
a post-processor.  The data layer (disk readers, collate) is outside the hot path (§8f next-4); ``gt_box_tensor`` is the
caller's, or -- when the batch carries the label keys the reference's collate emits (``object_bbx_center`` / ``_mask`` /
``object_ids`` / ``transformation_matrix_clean``) -- built by ``VoxelPostprocessor.generate_gt_bbx`` like
``dataset.post_process`` does (intermediate_fusion_dataset.py:585-606).
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Optional

import torch

from .postprocess import VoxelPostprocessor


def _ground_truth(batch_data: dict, post_processor: VoxelPostprocessor, supplied):
    if supplied is not None:
        return supplied
    if all("object_bbx_center" in cav for cav in batch_data.values()):
        return post_processor.generate_gt_bbx(batch_data)
    return None


def inference_late_fusion(batch_data: dict, model, post_processor: VoxelPostprocessor, gt_box_tensor: Optional[torch.Tensor] = None) -> dict:
    """One model call per cav (each on its own canvas), then ONE merged post-process: every agent's boxes are projected
    to the ego frame with its ``transformation_matrix`` and a single rotated NMS runs over the union
    (voxel_postprocessor.py:275-381).  ``batch_data``: {cav_id: {processed_lidar..., transformation_matrix, anchor_box}}."""
    output_dict = OrderedDict()
    with torch.no_grad():
        for cav_id, cav_content in batch_data.items():
            output_dict[cav_id] = model(cav_content)
        pred_box_tensor, pred_score = post_processor.post_process(batch_data, output_dict)
    return {"pred_box_tensor": pred_box_tensor, "pred_score": pred_score,
            "gt_box_tensor": _ground_truth(batch_data, post_processor, gt_box_tensor)}


def inference_early_fusion(batch_data: dict, model, post_processor: VoxelPostprocessor, gt_box_tensor: Optional[torch.Tensor] = None) -> dict:
    """``batch_data['ego']`` holds the whole (already fused / to-be-fused) frame; identity projection."""
    with torch.no_grad():
        output_dict = OrderedDict(ego=model(batch_data["ego"]))
        pred_box_tensor, pred_score = post_processor.post_process(batch_data, output_dict)
    return {"pred_box_tensor": pred_box_tensor, "pred_score": pred_score,
            "gt_box_tensor": _ground_truth(batch_data, post_processor, gt_box_tensor)}


def inference_intermediate_fusion(batch_data: dict, model, post_processor: VoxelPostprocessor, gt_box_tensor=None) -> dict:
    return inference_early_fusion(batch_data, model, post_processor, gt_box_tensor)
