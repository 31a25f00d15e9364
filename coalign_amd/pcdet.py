"""OpenPCDet-semantics BEV / 3-D IoU and NMS: the functions callers import from
``opencood/pcdet_utils/iou3d_nms/iou3d_nms_utils.py`` (``boxes_iou_bev`` :47-63, ``boxes_iou3d_gpu`` :66-98, ``nms_gpu`` :255-271), whose CUDA
extension does not exist here.  Pair geometry (fp32 corner / edge intersection with the 1e-2 margin of the extension) runs in
``coalign_boxes_iou_bev`` / ``coalign_boxes_overlap_bev``; the greedy suppression of ``nms_gpu`` walks the thresholded matrix on the host
(the extension does the same walk over its bit mask on the CPU).
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops


def boxes_iou_bev(boxes_a: torch.Tensor, boxes_b: torch.Tensor) -> torch.Tensor:
    assert boxes_a.shape[1] == boxes_b.shape[1] == 7
    return ops.boxes_iou_bev(boxes_a, boxes_b)


def boxes_iou3d_gpu(boxes_a: torch.Tensor, boxes_b: torch.Tensor) -> torch.Tensor:
    """[N, 7] x [M, 7] (x, y, z, dx, dy, dz, heading) -> 3-D IoU [N, M]: BEV overlap area x height overlap over the union volume."""
    assert boxes_a.shape[1] == boxes_b.shape[1] == 7
    a_max, a_min = (boxes_a[:, 2] + boxes_a[:, 5] / 2).view(-1, 1), (boxes_a[:, 2] - boxes_a[:, 5] / 2).view(-1, 1)
    b_max, b_min = (boxes_b[:, 2] + boxes_b[:, 5] / 2).view(1, -1), (boxes_b[:, 2] - boxes_b[:, 5] / 2).view(1, -1)
    overlaps_bev = ops.boxes_overlap_bev(boxes_a, boxes_b)
    overlaps_h = torch.clamp(torch.min(a_max, b_max) - torch.max(a_min, b_min), min=0)
    overlaps_3d = overlaps_bev * overlaps_h
    vol_a = (boxes_a[:, 3] * boxes_a[:, 4] * boxes_a[:, 5]).view(-1, 1)
    vol_b = (boxes_b[:, 3] * boxes_b[:, 4] * boxes_b[:, 5]).view(1, -1)
    return overlaps_3d / torch.clamp(vol_a + vol_b - overlaps_3d, min=1e-6)


def nms_gpu(boxes: torch.Tensor, scores: torch.Tensor, thresh: float, pre_maxsize=None, **kwargs):
    """Sort by score (descending), keep a box unless an earlier kept box overlaps it with BEV IoU > thresh; returns
    (indices into the input, None) like the reference."""
    assert boxes.shape[1] == 7
    order = scores.sort(0, descending=True)[1]
    if pre_maxsize is not None:
        order = order[:pre_maxsize]
    b = boxes[order].contiguous()
    k = b.shape[0]
    if k == 0:
        return order, None
    suppress = (ops.boxes_iou_bev(b, b) > thresh).cpu().numpy()
    removed = np.zeros(k, dtype=bool)
    keep = []
    for i in range(k):
        if not removed[i]:
            keep.append(i)
            removed[i + 1:] |= suppress[i, i + 1:]
    return order[torch.as_tensor(keep, dtype=torch.long, device=order.device)].contiguous(), None
