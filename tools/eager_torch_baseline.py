#!/usr/bin/env python
"""What the reference's own code path costs on this GPU: the per-frame detection path written the way opencood runs it --
stock eager PyTorch ops on ROCm (nn.Linear / BatchNorm1d / index scatter, Conv2d + BatchNorm2d + ReLU modules, F.affine_grid +
F.grid_sample, per-pixel attention with bmm + softmax, ~40 small tensor ops for the box decode) -- timed beside the gfx950
kernels of this repository on the same synthetic OPV2V frame, and compared numerically with them (a second, GPU-side parity
check against plain PyTorch).

The reference cannot travel to the GPU box; this file restates its op sequence (same files/lines as the host mirror cites).
Two things it cannot reproduce are replaced in the baseline's favour: rotated NMS (Shapely on the CPU in the reference, a few
ms per frame) goes through this repository's device NMS, and the range filter stays on the device.

    python tools/eager_torch_baseline.py [--agents 5] [--pillars 8000] [--steps 20]
"""
import argparse
import json
import math
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coalign_amd import backbone as bb_mod                       # noqa: E402
from coalign_amd.config import builtin_config                    # noqa: E402
from coalign_amd.detector import build_model, to_device          # noqa: E402
from coalign_amd.pose import normalize_pairwise_tfm              # noqa: E402
from coalign_amd.postprocess import build_postprocessor, nms_rotated   # noqa: E402
from coalign_amd.synthetic import fill_parameters_, make_frame   # noqa: E402


def eager_pillar_vfe(model, pl):
    """pillar_vfe.py:105-155 with torch ops."""
    vf, npts, coords = pl["voxel_features"], pl["voxel_num_points"], pl["voxel_coords"]
    vfe = model.pillar_vfe
    mean = vf[:, :, :3].sum(dim=1, keepdim=True) / npts.type_as(vf).view(-1, 1, 1)
    f_cluster = vf[:, :, :3] - mean
    f_center = torch.zeros_like(vf[:, :, :3])
    f_center[:, :, 0] = vf[:, :, 0] - (coords[:, 3].to(vf.dtype).unsqueeze(1) * vfe.voxel_x + vfe.x_offset)
    f_center[:, :, 1] = vf[:, :, 1] - (coords[:, 2].to(vf.dtype).unsqueeze(1) * vfe.voxel_y + vfe.y_offset)
    f_center[:, :, 2] = vf[:, :, 2] - (coords[:, 1].to(vf.dtype).unsqueeze(1) * vfe.voxel_z + vfe.z_offset)
    feats = torch.cat([vf, f_cluster, f_center], dim=-1)
    mask = (npts.unsqueeze(1) > torch.arange(vf.shape[1], device=vf.device).view(1, -1)).unsqueeze(-1).type_as(vf)
    feats = feats * mask
    pfn = vfe.pfn_layers[0]
    x = pfn.linear(feats)
    x = pfn.norm(x.permute(0, 2, 1)).permute(0, 2, 1)
    return torch.max(F.relu(x), dim=1)[0]


def eager_scatter(pf, coords, n, nx, ny):
    """point_pillar_scatter.py:15-72."""
    out = []
    for b in range(n):
        canvas = torch.zeros(pf.shape[1], nx * ny, dtype=pf.dtype, device=pf.device)
        sel = coords[:, 0] == b
        c = coords[sel]
        idx = (c[:, 1] + c[:, 2] * nx + c[:, 3]).long()
        canvas[:, idx] = pf[sel].t()
        out.append(canvas)
    return torch.stack(out).view(n, pf.shape[1], ny, nx)


def eager_warp(src, M, dsize):
    """torch_transformation_utils.py:322-331."""
    grid = F.affine_grid(M, [src.shape[0], src.shape[1], dsize[0], dsize[1]], align_corners=False).to(src)
    return F.grid_sample(src, grid, align_corners=False)


def eager_att_fusion(x, record_len, affine):
    """fusion_in_one.py:96-136."""
    _, C, H, W = x.shape
    out, start = [], 0
    for b, n in enumerate(record_len):
        xb = eager_warp(x[start: start + n], affine[b, 0, :n], (H, W))
        start += n
        q = xb.view(n, C, -1).permute(2, 0, 1)
        score = torch.bmm(q, q.transpose(1, 2)) / math.sqrt(C)
        ctx = torch.bmm(F.softmax(score, -1), q)
        out.append(ctx.permute(1, 2, 0).view(n, C, H, W)[0])
    return torch.stack(out)


def eager_forward(model, frame):
    """point_pillar_baseline_multiscale.py:93-135 with the stock modules (fast paths of this repository switched off)."""
    pl = frame["processed_lidar"]
    record_len = [int(v) for v in frame["record_len"]]
    n = sum(record_len)
    pf = eager_pillar_vfe(model, pl)
    nx, ny = model.scatter.nx, model.scatter.ny
    canvas = eager_scatter(pf, pl["voxel_coords"], n, nx, ny)
    affine = normalize_pairwise_tfm(frame["pairwise_t_matrix"], ny, nx, model.voxel_size[0])
    feats = model.backbone.get_multiscale_feature(canvas)
    fused = [eager_att_fusion(f, record_len, affine) for f in feats]
    x = model.backbone.decode_multiscale_feature(fused)
    if model.shrink_flag:
        x = model.shrink_conv(x)
    return {"cls_preds": model.cls_head(x), "reg_preds": model.reg_head(x), "dir_preds": model.dir_head(x)}


def eager_post_process(out, anchors, pp):
    """voxel_postprocessor.py:243-402 as tensor ops on the device (NMS: see the module docstring)."""
    prob = torch.sigmoid(out["cls_preds"].permute(0, 2, 3, 1)).reshape(1, -1)
    N = out["reg_preds"].shape[0]
    deltas = out["reg_preds"].permute(0, 2, 3, 1).contiguous().view(N, -1, 7)
    a = anchors.view(-1, 7).float().repeat(N, 1, 1)
    d = torch.sqrt(a[..., 4] ** 2 + a[..., 5] ** 2)
    boxes = torch.zeros_like(deltas)
    boxes[..., [0, 1]] = deltas[..., [0, 1]] * d.unsqueeze(-1) + a[..., [0, 1]]
    boxes[..., 2] = deltas[..., 2] * a[..., 3] + a[..., 2]
    boxes[..., [3, 4, 5]] = torch.exp(deltas[..., [3, 4, 5]]) * a[..., [3, 4, 5]]
    boxes[..., 6] = deltas[..., 6] + a[..., 6]
    mask = torch.gt(prob, pp["target_args"]["score_threshold"]).view(1, -1)
    b3 = torch.masked_select(boxes[0], mask.unsqueeze(2).repeat(1, 1, 7)[0]).view(-1, 7)
    scores = torch.masked_select(prob[0], mask[0])
    if len(b3) == 0:
        return None, None
    dm = out["dir_preds"].permute(0, 2, 3, 1).contiguous().reshape(1, -1, 2)[mask]
    labels = torch.max(dm, dim=-1)[1]
    off, period = pp["dir_args"]["dir_offset"], math.pi
    lp = lambda v, o, p: v - torch.floor(v / p + o) * p
    b3[..., 6] = lp(b3[..., 6] - off, 0, period) + off + period * labels.to(dm.dtype)
    b3[..., 6] = lp(b3[..., 6], 0.5, 2 * math.pi)
    bb = b3[:, [0, 1, 2, 5, 4, 3, 6]]
    tmpl = b3.new_tensor([[1, -1, -1], [1, 1, -1], [-1, 1, -1], [-1, -1, -1], [1, -1, 1], [1, 1, 1], [-1, 1, 1], [-1, -1, 1]]) / 2
    c3 = bb[:, None, 3:6].repeat(1, 8, 1) * tmpl[None]
    cs, sn = torch.cos(bb[:, 6]), torch.sin(bb[:, 6])
    z, o = torch.zeros_like(cs), torch.ones_like(cs)
    rot = torch.stack((cs, sn, z, -sn, cs, z, z, z, o), dim=1).view(-1, 3, 3)
    c3 = torch.matmul(c3, rot) + bb[:, None, 0:3]
    xl = c3[:, :, 0].max(1)[0] - c3[:, :, 0].min(1)[0]
    yl = c3[:, :, 1].max(1)[0] - c3[:, :, 1].min(1)[0]
    keep = (xl <= 6) & (yl <= 6) & (yl != 0) & (c3[:, :, 2].min(1)[0] >= -3) & (c3[:, :, 2].max(1)[0] <= 1)
    c3, scores = c3[keep], scores[keep]
    k = torch.from_numpy(nms_rotated(c3, scores, pp["nms_thresh"]).astype(np.int64)).to(c3.device)
    c3, scores = c3[k], scores[k]
    r = pp["gt_range"]
    inr = ((c3[:, :, 0] >= r[0]) & (c3[:, :, 0] <= r[3]) & (c3[:, :, 1] >= r[1]) & (c3[:, :, 1] <= r[4])).all(dim=1)
    return c3[inr], scores[inr]


def timed(fn, steps, warmup):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(steps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / steps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--agents", type=int, default=5)
    ap.add_argument("--pillars", type=int, default=8000)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    h = builtin_config("opv2v_coalign")
    model = build_model(h)
    fill_parameters_(model, seed=0, cls_bias=-2.0)
    with torch.no_grad():
        model.reg_head.weight.mul_(0.02); model.cls_head.weight.mul_(0.3)
    model = model.to(dev).eval()
    post = build_postprocessor(h["postprocess"], False)
    anchors = torch.from_numpy(post.generate_anchor_box()).to(dev)
    frame = to_device(make_frame(h, a.agents, pillars_per_agent=a.pillars, seed=303), dev)
    data = {"ego": {"transformation_matrix": torch.eye(4, device=dev), "anchor_box": anchors}}
    with torch.no_grad():
        bb_mod.FAST_INFERENCE = False
        out_eager = eager_forward(model, frame)
        boxes_e, scores_e = eager_post_process(out_eager, anchors, h["postprocess"])
        ms_eager_model = timed(lambda: eager_forward(model, frame), a.steps, a.warmup)
        ms_eager_all = timed(lambda: eager_post_process(eager_forward(model, frame), anchors, h["postprocess"]), a.steps, a.warmup)
        bb_mod.FAST_INFERENCE = True
        out_hip = model(frame)
        boxes_h, scores_h = post.post_process(data, {"ego": out_hip})
        ms_hip_model = timed(lambda: model(frame), a.steps, a.warmup)
        ms_hip_all = timed(lambda: post.post_process(data, {"ego": model(frame)}), a.steps, a.warmup)
    rel = {k: float((out_eager[k] - out_hip[k]).abs().max() / out_eager[k].abs().max()) for k in out_eager}
    rep = {"workload": f"opv2v_coalign {a.agents} agents x {a.pillars} pillars", "steps": a.steps,
           "eager_torch_ms": {"model": round(ms_eager_model, 3), "model_plus_postprocess": round(ms_eager_all, 3)},
           "this_repo_ms": {"model": round(ms_hip_model, 3), "model_plus_postprocess_serial": round(ms_hip_all, 3)},
           "speedup_model": round(ms_eager_model / ms_hip_model, 2), "speedup_all": round(ms_eager_all / ms_hip_all, 2),
           "max_rel_diff_head_outputs": rel,
           "boxes": {"eager": 0 if boxes_e is None else int(boxes_e.shape[0]), "hip": 0 if boxes_h is None else int(boxes_h.shape[0])}}
    if boxes_e is not None and boxes_h is not None and boxes_e.shape == boxes_h.shape:
        rep["boxes"]["max_abs_diff_m"] = float((boxes_e - boxes_h).abs().max())
    print(json.dumps(rep))


if __name__ == "__main__":
    main()
