"""Detection evaluation: TP / FP matching with rotated IoU, global score sort, VOC-2010 AP.

Function names (including the reference's spelling ``caluclate_tp_fp``) and the ``result_stat`` dictionary layout follow
opencood/utils/eval_utils.py:19-169.  The IoU matrix -- one Shapely call per (detection, ground truth) pair in the
reference -- comes from the device (``coalign_iou_rotated_matrix``, same float64 clipping as the NMS); the greedy
assignment walks at most a few hundred detections per frame on the host.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from . import ops

IOU_THRESHOLDS = (0.3, 0.5, 0.7)


def new_result_stat(thresholds=IOU_THRESHOLDS) -> Dict[float, dict]:
    """The accumulator the reference's inference loop creates (opencood/tools/inference.py:106-108)."""
    return {t: {"tp": [], "fp": [], "gt": 0, "score": []} for t in thresholds}


def match_tp_fp(iou: np.ndarray, det_score: np.ndarray, iou_thresh: float) -> Tuple[List[int], List[int], np.ndarray]:
    """Greedy matching in descending score order: a detection is a TP if its best IoU against the still unmatched ground
    truths reaches the threshold (compared in float32 like the reference's float32 IoU array), and consumes that ground
    truth (first one on ties).  Equal scores: the reference's ``np.argsort(-score)`` is unstable; defined here as stable."""
    order = np.argsort(-det_score, kind="stable")
    alive = np.ones(iou.shape[1], dtype=bool)
    thr = np.float32(iou_thresh)
    tp, fp = [], []
    for d in order:
        if not alive.any():
            fp.append(1); tp.append(0)
            continue
        row = np.where(alive, iou[d], np.float32(-1.0))
        g = int(np.argmax(row))
        if row[g] < thr:
            fp.append(1); tp.append(0)
        else:
            fp.append(0); tp.append(1)
            alive[g] = False
    return tp, fp, det_score[order]


def caluclate_tp_fp(det_boxes: Optional[torch.Tensor], det_score: Optional[torch.Tensor], gt_boxes: torch.Tensor,
                    result_stat: Dict[float, dict], iou_thresh: float) -> None:
    """Accumulate one frame into ``result_stat[iou_thresh]`` (eval_utils.py:45-96).  ``det_boxes`` [N, 8, 3] or [N, 4, 2]
    device tensor or None, ``gt_boxes`` [G, 8, 3]."""
    gt = int(gt_boxes.shape[0])
    tp: List[int] = []
    fp: List[int] = []
    if det_boxes is not None:
        scores = det_score.detach().float().cpu().numpy()
        if det_boxes.shape[0] and gt:
            iou = ops.iou_rotated_matrix(det_boxes, gt_boxes.to(det_boxes.device)).cpu().numpy()
        else:
            iou = np.zeros((det_boxes.shape[0], gt), dtype=np.float32)
        tp, fp, sorted_scores = match_tp_fp(iou, scores, iou_thresh)
        result_stat[iou_thresh]["score"] += sorted_scores.tolist()
    result_stat[iou_thresh]["fp"] += fp
    result_stat[iou_thresh]["tp"] += tp
    result_stat[iou_thresh]["gt"] += gt


def voc_ap(rec: List[float], prec: List[float]):
    """VOC 2010 AP: precision envelope integrated over the recall steps (eval_utils.py:19-42)."""
    mrec = np.concatenate(([0.0], np.asarray(rec, dtype=np.float64), [1.0]))
    mpre = np.concatenate(([0.0], np.asarray(prec, dtype=np.float64), [0.0]))
    mpre = np.maximum.accumulate(mpre[::-1])[::-1]
    step = np.nonzero(mrec[1:] != mrec[:-1])[0] + 1
    ap = float(np.sum((mrec[step] - mrec[step - 1]) * mpre[step]))
    return ap, mrec.tolist(), mpre.tolist()


def calculate_ap(result_stat: Dict[float, dict], iou: float):
    """Global score sort over all frames, cumulative TP / FP, recall / precision, VOC AP (eval_utils.py:100-142)."""
    st = result_stat[iou]
    fp, tp, score = np.array(st["fp"]), np.array(st["tp"]), np.array(st["score"])
    assert len(fp) == len(tp) == len(score)
    order = np.argsort(-score, kind="stable")
    fp_c, tp_c = np.cumsum(fp[order]), np.cumsum(tp[order])
    gt_total = st["gt"]
    rec = (tp_c / gt_total).tolist() if len(tp_c) else []
    prec = (tp_c / np.maximum(fp_c + tp_c, 1)).tolist() if len(tp_c) else []
    return voc_ap(rec, prec)


def eval_final_results(result_stat: Dict[float, dict], save_path: Optional[str] = None, infer_info: Optional[str] = None):
    """AP at IoU 0.3 / 0.5 / 0.7; optionally dumps the reference's ``eval[_<info>].yaml`` (eval_utils.py:145-169)."""
    ap30, _, _ = calculate_ap(result_stat, 0.30)
    ap50, mrec50, mpre50 = calculate_ap(result_stat, 0.50)
    ap70, mrec70, mpre70 = calculate_ap(result_stat, 0.70)
    if save_path is not None:
        import os
        import yaml
        name = "eval.yaml" if infer_info is None else f"eval_{infer_info}.yaml"
        with open(os.path.join(save_path, name), "w") as fh:
            yaml.dump({"ap30": ap30, "ap_50": ap50, "ap_70": ap70, "mpre_50": mpre50, "mrec_50": mrec50,
                       "mpre_70": mpre70, "mrec_70": mrec70}, fh, default_flow_style=False)
    print("The Average Precision at IOU 0.3 is %.2f, The Average Precision at IOU 0.5 is %.2f, "
          "The Average Precision at IOU 0.7 is %.2f" % (ap30, ap50, ap70))
    return ap30, ap50, ap70
