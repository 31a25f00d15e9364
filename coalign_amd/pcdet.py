"""OpenPCDet-semantics BEV / 3-D IoU and NMS: the functions callers import from
``opencood/pcdet_utils/iou3d_nms/iou3d_nms_utils.py`` (``boxes_iou_bev`` :32-46, ``boxes_iou3d_gpu`` :147-181, ``nms_gpu`` :255-271,
``nms_normal_gpu`` :274-289), whose CUDA extension does not exist here.  Pair geometry (fp32 corner / edge intersection with the
1e-2 margin of the extension) runs in ``coalign_boxes_iou_bev`` / ``coalign_boxes_overlap_bev``; both NMS flavours run entirely on
the device in ``coalign_pcdet_nms``: a 64-wide suppression bitmask per 64 x 64 tile of the pair matrix and a single-wavefront greedy
walk over it (the extension builds the same mask on the GPU, copies it to the host and walks it there, iou3d_nms.cpp:90-137).
"""
from __future__ import annotations

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


def _nms(boxes: torch.Tensor, scores: torch.Tensor, thresh: float, pre_maxsize, normal: bool):
    assert boxes.shape[1] == 7
    # stable: equal scores keep their input order (torch's default sort leaves ties implementation-defined, like the reference's)
    order = scores.sort(dim=0, descending=True, stable=True)[1]
    if pre_maxsize is not None:
        order = order[:pre_maxsize]
    if order.numel() == 0:
        return order, None
    keep, cnt = ops.pcdet_nms(boxes[order].contiguous(), thresh, normal)
    return order[keep[: int(cnt.item())].long()].contiguous(), None


def nms_gpu(boxes: torch.Tensor, scores: torch.Tensor, thresh: float, pre_maxsize=None, **kwargs):
    """Rotated BEV boxes: sort by score (descending), keep a box unless an earlier kept box overlaps it with fp32 BEV IoU > thresh;
    returns (indices into the input, None) like the reference."""
    return _nms(boxes, scores, thresh, pre_maxsize, False)


def nms_normal_gpu(boxes: torch.Tensor, scores: torch.Tensor, thresh: float, **kwargs):
    """Heading ignored: axis-aligned (x, y, dx, dy) IoU; otherwise as ``nms_gpu`` (no ``pre_maxsize`` in the reference either)."""
    return _nms(boxes, scores, thresh, None, True)
