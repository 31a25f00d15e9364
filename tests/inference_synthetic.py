#!/usr/bin/env python
"""Test infrastructure (lives under tests/ because it may call the CPU oracle).

Synthetic-ground-truth AP harness: the reference's inference loop
(opencood/tools/inference.py:100-160) on seeded synthetic frames -- model -> post-process -> TP/FP at IoU 0.3/0.5/0.7 ->
VOC AP -- once through the gfx950 path and, with ``--check-oracle``, once through the CPU oracle on the same inputs.

There are no trained checkpoints or datasets here, so the ground truth is *planted*: every frame's label set is a seeded
rigid jitter of a subset of the oracle-independent HIP detections plus a few unmatched objects.  The absolute AP therefore
says nothing about detection quality; what the harness shows is that both pipelines produce the same TP/FP sequence and
the same AP ("AP@0.7 vs ref" of BASELINE.json's metric) on identical inputs.

    python tests/inference_synthetic.py --config mini_coalign --frames 6 --agents 3 --pillars 150 --check-oracle
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from coalign_amd import evaluation as ev                     # noqa: E402
from coalign_amd.config import builtin_config                # noqa: E402
from coalign_amd.detector import build_model, to_device      # noqa: E402
from coalign_amd.inference import inference_intermediate_fusion   # noqa: E402
from coalign_amd.postprocess import build_postprocessor      # noqa: E402
from coalign_amd.synthetic import fill_parameters_, make_frame    # noqa: E402


def plant_ground_truth(pred: torch.Tensor, seed: int) -> torch.Tensor:
    """[G, 8, 3] corners: every second detection shifted rigidly by a seeded (dx, dy) of up to ~0.6 m, plus three
    objects nobody detects."""
    rs = np.random.RandomState(seed)
    base = pred[::2].cpu().numpy().astype(np.float32) if pred is not None and len(pred) else np.zeros((0, 8, 3), np.float32)
    shift = np.zeros((len(base), 1, 3), dtype=np.float32)
    shift[:, 0, :2] = rs.uniform(-0.6, 0.6, (len(base), 2))
    tmpl = np.array([[2.25, -1, -1], [2.25, 1, -1], [-2.25, 1, -1], [-2.25, -1, -1],
                     [2.25, -1, 0.5], [2.25, 1, 0.5], [-2.25, 1, 0.5], [-2.25, -1, 0.5]], dtype=np.float32)
    extra = tmpl[None] + np.concatenate([rs.uniform(-10, 10, (3, 1, 2)), np.zeros((3, 1, 1))], axis=2).astype(np.float32)
    return torch.from_numpy(np.concatenate([base + shift, extra]))


def run(config: str, frames: int, agents: int, pillars: int, check_oracle: bool, head_scale=(0.01, 0.05), oracle_heads: str = "oracle",
        cls_bias: float = -1.0) -> dict:
    """``oracle_heads`` = "device": the oracle's post-processing + evaluation run on the DEVICE model's head outputs (the model itself is compared
    with the oracle elsewhere): at full size ~1e5 logits lie within 1e-5 -- the two implementations' logit noise -- of the score threshold, so two
    independently computed candidate sets cannot be expected to coincide; decode, NMS, range filter, matching and AP can."""
    dev = torch.device("cuda:0")
    h = builtin_config(config)
    model = build_model(h)
    fill_parameters_(model, seed=0, cls_bias=cls_bias)
    with torch.no_grad():                  # small box deltas / moderate logits so that a usable number of boxes survives
        model.reg_head.weight.mul_(head_scale[0]); model.reg_head.bias.zero_(); model.cls_head.weight.mul_(head_scale[1])
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    model = model.to(dev).eval()
    post = build_postprocessor(h["postprocess"], False)
    anchors = torch.from_numpy(post.generate_anchor_box())
    stat_hip, stat_cpu = ev.new_result_stat(), ev.new_result_stat()
    n_boxes = n_cand = n_ties = 0
    all_frames = [make_frame(h, agents, pillars_per_agent=pillars, seed=100 + i, spread_xy=(4.0, 2.0), spread_yaw=45.0) for i in range(frames)]
    for i in range(frames):
        frame = all_frames[i]
        batch = {"ego": dict(to_device(frame, dev), transformation_matrix=torch.eye(4, device=dev), anchor_box=anchors.to(dev))}
        res = inference_intermediate_fusion(batch, model, post)
        gt = plant_ground_truth(res["pred_box_tensor"], 7000 + i)
        n_boxes += 0 if res["pred_box_tensor"] is None else len(res["pred_box_tensor"])
        for thr in ev.IOU_THRESHOLDS:
            ev.caluclate_tp_fp(res["pred_box_tensor"], res["pred_score"], gt.to(dev), stat_hip, thr)
        if check_oracle:
            from oracle import coalign_oracle as oracle
            with torch.no_grad():
                if oracle_heads == "device":
                    out = {k: v.cpu() for k, v in model(batch["ego"]).items()}
                else:
                    out = oracle.coalign_forward(sd, h["model"]["args"], frame)
            ob, osc, info = oracle.post_process([out], anchors, h["postprocess"])
            cs = np.asarray(info["cand_scores"])
            n_cand += len(cs)
            n_ties += len(cs) - len(np.unique(cs))
            for thr in ev.IOU_THRESHOLDS:
                oracle.caluclate_tp_fp(None if ob is None else ob.numpy(), None if osc is None else osc.numpy(), gt.numpy(), stat_cpu, thr)
    report = {"config": config, "frames": frames, "agents": agents, "detections": n_boxes,
              "hip": {f"ap{int(t * 100)}": ev.calculate_ap(stat_hip, t)[0] for t in ev.IOU_THRESHOLDS}}
    if check_oracle:
        from oracle import coalign_oracle as oracle
        report["oracle"] = {f"ap{int(t * 100)}": oracle.calculate_ap(stat_cpu, t)[0] for t in ev.IOU_THRESHOLDS}
        report["candidates"], report["tied_candidate_scores"] = n_cand, n_ties      # equal fp32 scores: their order is the sort algorithm's, also in the reference
        report["counts_identical"] = all(sum(stat_hip[t]["tp"]) == sum(stat_cpu[t]["tp"]) and sum(stat_hip[t]["fp"]) == sum(stat_cpu[t]["fp"])
                                         and stat_hip[t]["gt"] == stat_cpu[t]["gt"] for t in ev.IOU_THRESHOLDS)
        report["tp_fp_identical"] = all(stat_hip[t]["tp"] == stat_cpu[t]["tp"] and stat_hip[t]["fp"] == stat_cpu[t]["fp"]
                                        and stat_hip[t]["gt"] == stat_cpu[t]["gt"] for t in ev.IOU_THRESHOLDS)
    return report


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="mini_coalign")
    ap.add_argument("--frames", type=int, default=6)
    ap.add_argument("--agents", type=int, default=3)
    ap.add_argument("--pillars", type=int, default=150)
    ap.add_argument("--check-oracle", action="store_true")
    a = ap.parse_args()
    torch.set_num_threads(min(16, torch.get_num_threads()))
    print(json.dumps(run(a.config, a.frames, a.agents, a.pillars, a.check_oracle)))
