"""Per-frame inference drivers: the reference's
``inference_late_fusion`` / ``inference_early_fusion`` / ``inference_intermediate_fusion``
(opencood/tools/inference_utils.py:17-46, 122-173) with the dataset object reduced to what they use from it --
a post-processor.  The data layer (disk readers, collate, ground-truth boxes) is outside the hot path (§8f next-4), so
``gt_box_tensor`` is returned only when the caller supplies it.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Optional

import torch

from .postprocess import VoxelPostprocessor


def inference_late_fusion(batch_data: dict, model, post_processor: VoxelPostprocessor, gt_box_tensor: Optional[torch.Tensor] = None) -> dict:
    """One model call per cav (each on its own canvas), then ONE merged post-process: every agent's boxes are projected
    to the ego frame with its ``transformation_matrix`` and a single rotated NMS runs over the union
    (voxel_postprocessor.py:275-381).  ``batch_data``: {cav_id: {processed_lidar..., transformation_matrix, anchor_box}}."""
    output_dict = OrderedDict()
    with torch.no_grad():
        for cav_id, cav_content in batch_data.items():
            output_dict[cav_id] = model(cav_content)
        pred_box_tensor, pred_score = post_processor.post_process(batch_data, output_dict)
    return {"pred_box_tensor": pred_box_tensor, "pred_score": pred_score, "gt_box_tensor": gt_box_tensor}


def inference_early_fusion(batch_data: dict, model, post_processor: VoxelPostprocessor, gt_box_tensor: Optional[torch.Tensor] = None) -> dict:
    """``batch_data['ego']`` holds the whole (already fused / to-be-fused) frame; identity projection."""
    with torch.no_grad():
        output_dict = OrderedDict(ego=model(batch_data["ego"]))
        pred_box_tensor, pred_score = post_processor.post_process(batch_data, output_dict)
    return {"pred_box_tensor": pred_box_tensor, "pred_score": pred_score, "gt_box_tensor": gt_box_tensor}


def inference_intermediate_fusion(batch_data: dict, model, post_processor: VoxelPostprocessor, gt_box_tensor=None) -> dict:
    return inference_early_fusion(batch_data, model, post_processor, gt_box_tensor)
