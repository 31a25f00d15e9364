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
    srt = boxes[order].contiguous()
    if srt.shape[0] > PCDET_NMS_DEVICE_MAX:
        return order[_nms_large(srt, thresh, normal)].contiguous(), None
    keep, cnt = ops.pcdet_nms(srt, thresh, normal)
    return order[keep[: int(cnt.item())].long()].contiguous(), None


PCDET_NMS_DEVICE_MAX = 16384          # coalign_pcdet_nms walks the bitmask on the device for up to this many boxes


def _nms_large(srt: torch.Tensor, thresh: float, normal: bool, rows_per_chunk: int = 2048) -> torch.Tensor:
    """More boxes than the device walk takes (the reference's nms_gpu has no size limit when ``pre_maxsize`` is None, iou3d_nms_utils.py:255-271):
    the extension's own scheme -- suppression matrix on the GPU, greedy walk on the host (iou3d_nms.cpp:90-137) -- in row chunks: IoU of
    ``rows_per_chunk`` sorted boxes against all, thresholded on the device, one bool block per chunk to the host."""
    import numpy as np
    n = srt.shape[0]
    removed = np.zeros(n, dtype=bool)
    keep = []
    for r0 in range(0, n, rows_per_chunk):
        rows = srt[r0: r0 + rows_per_chunk]
        if normal:
            a, b = rows, srt
            lo_x = torch.max((a[:, 0] - a[:, 3] / 2)[:, None], (b[:, 0] - b[:, 3] / 2)[None])
            hi_x = torch.min((a[:, 0] + a[:, 3] / 2)[:, None], (b[:, 0] + b[:, 3] / 2)[None])
            lo_y = torch.max((a[:, 1] - a[:, 4] / 2)[:, None], (b[:, 1] - b[:, 4] / 2)[None])
            hi_y = torch.min((a[:, 1] + a[:, 4] / 2)[:, None], (b[:, 1] + b[:, 4] / 2)[None])
            inter = torch.clamp(hi_x - lo_x, min=0) * torch.clamp(hi_y - lo_y, min=0)
            iou = inter / torch.clamp((a[:, 3] * a[:, 4])[:, None] + (b[:, 3] * b[:, 4])[None] - inter, min=1e-8)      # iou3d_nms_kernel.cu:313-325
        else:
            iou = ops.boxes_iou_bev(rows, srt)
        sup = (iou > thresh).cpu().numpy()
        for k in range(sup.shape[0]):
            i = r0 + k
            if removed[i]:
                continue
            keep.append(i)
            removed[i + 1:] |= sup[k, i + 1:]
    return torch.tensor(keep, dtype=torch.long, device=srt.device)


def nms_gpu(boxes: torch.Tensor, scores: torch.Tensor, thresh: float, pre_maxsize=None, **kwargs):
    """Rotated BEV boxes: sort by score (descending), keep a box unless an earlier kept box overlaps it with fp32 BEV IoU > thresh;
    returns (indices into the input, None) like the reference."""
    return _nms(boxes, scores, thresh, pre_maxsize, False)


def nms_normal_gpu(boxes: torch.Tensor, scores: torch.Tensor, thresh: float, **kwargs):
    """Heading ignored: axis-aligned (x, y, dx, dy) IoU; otherwise as ``nms_gpu`` (no ``pre_maxsize`` in the reference either)."""
    return _nms(boxes, scores, thresh, None, True)
