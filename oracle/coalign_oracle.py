"""CPU oracle for the CoAlign per-frame detection hot path.

TEST INFRASTRUCTURE ONLY.  This module is a plain numpy / torch-CPU *restatement*
of the reference algorithm (yifanlu0227/CoAlign @ 2024_08_07).  It exists so that
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg can
check / time the HIP path against the reference's arithmetic.  Nothing under
``coalign_amd/`` imports it; the product path has no CPU fallback.

Pinning status (details in DESIGN.md §Oracle):
* rows A-L and O of SURVEY §8a (pillar VFE, scatter, pose normalisation, affine
  warp, attention / max fusion, backbone + heads, anchors, decode, corner
  geometry, filters) are PINNED: ``tests/golden/*.npz`` were produced by importing
  the reference itself in the build container (``tests/golden/make_golden.py``)
  and ``tests/test_oracle_golden.py`` checks every function below against them.
* row M (rotated NMS): the *control flow* (top-1000, descending order, strict
  ``>``, float32 IoU array) is pinned by running the reference's own
  ``nms_rotated`` with a polygon class backed by this oracle's clipping routine;
  the polygon *area arithmetic* lives in Shapely 2.0.0 -> GEOS, a third-party
  dependency that is neither vendored nor installed here ->
  **parity unpinned** for the IoU arithmetic.  It is restated as fp64 convex
  clipping and cross-checked against an independent second algorithm plus
  hand-computed known answers.

Every function cites the reference file:line it follows.  Paths are relative to
the reference repository root.
"""
from __future__ import annotations

import ctypes
import math
import os
import subprocess
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

_HERE = os.path.dirname(os.path.abspath(__file__))

# --------------------------------------------------------------------------------------
# Row A + B: PillarVFE  (opencood/models/sub_modules/pillar_vfe.py)
# --------------------------------------------------------------------------------------


def pillar_augment(voxel_features: torch.Tensor, voxel_num_points: torch.Tensor, coords: torch.Tensor,
                   voxel_size: Sequence[float], pc_range: Sequence[float]) -> torch.Tensor:
    """10-d point features, padded rows zeroed.  pillar_vfe.py:105-149.

    f = [x, y, z, i, xyz - mean_pillar, x - cx, y - cy, z - cz]; the mean divides the sum over
    all P slots by ``voxel_num_points`` (:118-120); pillar centres from integer coords
    (agent, z, y, x) (:123-131) with offsets ``voxel/2 + range_min`` (:84-89, python floats).
    Rows with index >= num_points are multiplied by 0 (:144-149).
    """
    vf = voxel_features if voxel_features.dtype == torch.float64 else voxel_features.float()      # (float64: the yardstick form of coalign_forward(dtype=...))
    npts = voxel_num_points.to(vf.dtype).view(-1, 1, 1)
    xyz = vf[:, :, :3]
    mean = xyz.sum(dim=1, keepdim=True) / npts
    f_cluster = xyz - mean
    vx, vy, vz = float(voxel_size[0]), float(voxel_size[1]), float(voxel_size[2])
    x_off = vx / 2 + pc_range[0]
    y_off = vy / 2 + pc_range[1]
    z_off = vz / 2 + pc_range[2]
    cf = coords.to(vf.dtype)
    f_center = torch.empty_like(xyz)
    f_center[:, :, 0] = vf[:, :, 0] - (cf[:, 3].unsqueeze(1) * vx + x_off)
    f_center[:, :, 1] = vf[:, :, 1] - (cf[:, 2].unsqueeze(1) * vy + y_off)
    f_center[:, :, 2] = vf[:, :, 2] - (cf[:, 1].unsqueeze(1) * vz + z_off)
    feats = torch.cat([vf, f_cluster, f_center], dim=-1)
    P = feats.shape[1]
    slot = torch.arange(P, dtype=torch.int32).view(1, P)
    mask = (voxel_num_points.int().view(-1, 1) > slot).unsqueeze(-1).to(vf.dtype)
    return feats * mask


def pfn_layer(feats10: torch.Tensor, weight: torch.Tensor, bn_weight: torch.Tensor, bn_bias: torch.Tensor,
              bn_mean: torch.Tensor, bn_var: torch.Tensor, eps: float = 1e-3) -> torch.Tensor:
    """Linear(10->64, no bias) -> BatchNorm1d(eval) -> ReLU -> max over points.  pillar_vfe.py:31-53.

    Padded (zeroed) rows still take part in the max: they contribute relu(BN(0)) (:46).
    """
    x = feats10 @ weight.t()                                    # [M, P, 64]
    x = (x - bn_mean) / torch.sqrt(bn_var + eps) * bn_weight + bn_bias
    x = torch.relu(x)
    return x.max(dim=1).values                                   # [M, 64]


def pillar_vfe(voxel_features, voxel_num_points, coords, sd: Dict[str, torch.Tensor], voxel_size, pc_range,
               prefix: str = "pillar_vfe.pfn_layers.0.") -> torch.Tensor:
    f = pillar_augment(voxel_features, voxel_num_points, coords, voxel_size, pc_range)
    return pfn_layer(f, sd[prefix + "linear.weight"], sd[prefix + "norm.weight"], sd[prefix + "norm.bias"],
                     sd[prefix + "norm.running_mean"], sd[prefix + "norm.running_var"], eps=1e-3)


# --------------------------------------------------------------------------------------
# Row C: PointPillarScatter  (opencood/models/sub_modules/point_pillar_scatter.py:15-72)
# --------------------------------------------------------------------------------------


def scatter(pillar_features: torch.Tensor, coords: torch.Tensor, n_agents: int, nx: int, ny: int) -> torch.Tensor:
    """canvas[b, :, idx] = feats with idx = z + y*nx + x (:54, nz == 1); zero elsewhere; -> [N, C, ny, nx]."""
    C = pillar_features.shape[1]
    canvas = torch.zeros(n_agents, C, ny * nx, dtype=pillar_features.dtype)
    c = coords.long()
    for b in range(n_agents):
        sel = c[:, 0] == b
        idx = c[sel, 1] + c[sel, 2] * nx + c[sel, 3]
        canvas[b][:, idx] = pillar_features[sel].t()
    return canvas.view(n_agents, C, ny, nx)


# --------------------------------------------------------------------------------------
# Row F: normalize_pairwise_tfm  (opencood/utils/transformation_utils.py:69-91)
# --------------------------------------------------------------------------------------


def normalize_pairwise_tfm(pairwise_t_matrix: torch.Tensor, H: int, W: int, discrete_ratio: float,
                           downsample_rate: float = 1) -> torch.Tensor:
    """[B,L,L,4,4] f64 -> [B,L,L,2,3] f64 normalised affine (rows {0,1}, cols {0,1,3}; see :83-89)."""
    m = pairwise_t_matrix[:, :, :, [0, 1], :][:, :, :, :, [0, 1, 3]].clone()
    m[..., 0, 1] = m[..., 0, 1] * H / W
    m[..., 1, 0] = m[..., 1, 0] * W / H
    m[..., 0, 2] = m[..., 0, 2] / (downsample_rate * discrete_ratio * W) * 2
    m[..., 1, 2] = m[..., 1, 2] / (downsample_rate * discrete_ratio * H) * 2
    return m


def x_to_world(pose: Sequence[float]) -> np.ndarray:
    """[x,y,z,roll,yaw,pitch] (degrees) -> 4x4 f64.  opencood/utils/transformation_utils.py:263-306."""
    x, y, z, roll, yaw, pitch = pose
    c_y, s_y = np.cos(np.radians(yaw)), np.sin(np.radians(yaw))
    c_r, s_r = np.cos(np.radians(roll)), np.sin(np.radians(roll))
    c_p, s_p = np.cos(np.radians(pitch)), np.sin(np.radians(pitch))
    m = np.identity(4)
    m[0, 3], m[1, 3], m[2, 3] = x, y, z
    m[0, 0] = c_p * c_y
    m[0, 1] = c_y * s_p * s_r - s_y * c_r
    m[0, 2] = -c_y * s_p * c_r - s_y * s_r
    m[1, 0] = s_y * c_p
    m[1, 1] = s_y * s_p * s_r + c_y * c_r
    m[1, 2] = -s_y * s_p * c_r + c_y * s_r
    m[2, 0] = s_p
    m[2, 1] = -c_p * s_r
    m[2, 2] = c_p * c_r
    return m


def pairwise_transformation(poses: Sequence[Sequence[float]], max_cav: int) -> np.ndarray:
    """[L,L,4,4] f64 with entry [i,j] = T_{j<-i} = solve(T_j, T_i).  transformation_utils.py:22-67."""
    out = np.tile(np.eye(4), (max_cav, max_cav, 1, 1))
    t = [x_to_world(p) for p in poses]
    for i in range(len(t)):
        for j in range(len(t)):
            if i != j:
                out[i, j] = np.linalg.solve(t[j], t[i])
    return out


# --------------------------------------------------------------------------------------
# Row G: warp_affine_simple  (opencood/models/sub_modules/torch_transformation_utils.py:322-331)
# explicit restatement of F.affine_grid (fp64) + F.grid_sample (bilinear, zeros, align_corners=False)
# --------------------------------------------------------------------------------------


def affine_grid_f64(theta: torch.Tensor, H: int, W: int) -> torch.Tensor:
    """theta [N,2,3] f64 -> sampling grid [N,H,W,2] **computed in f64, then cast to f32** (:328-330)."""
    xs = (2.0 * torch.arange(W, dtype=torch.float64) + 1.0) / W - 1.0
    ys = (2.0 * torch.arange(H, dtype=torch.float64) + 1.0) / H - 1.0
    yy, xx = torch.meshgrid(ys, xs, indexing="ij")
    th = theta.double()
    gx = th[:, 0, 0, None, None] * xx + th[:, 0, 1, None, None] * yy + th[:, 0, 2, None, None]
    gy = th[:, 1, 0, None, None] * xx + th[:, 1, 1, None, None] * yy + th[:, 1, 2, None, None]
    return torch.stack([gx, gy], dim=-1).float()


def grid_sample_bilinear_zeros(src: torch.Tensor, grid: torch.Tensor) -> torch.Tensor:
    """src [N,C,H,W] f32, grid [N,Ho,Wo,2] f32 -> [N,C,Ho,Wo].

    Un-normalise with the CPU kernel's form ``(g + 1) * (size / 2) - 0.5``; taps at floor / floor+1;
    weights nw=(1-tx)(1-ty) ...; out-of-range taps contribute 0.
    """
    N, C, H, W = src.shape
    grid = grid.to(src.dtype)                                    # (float32 in the reference; the float64 yardstick keeps the float32-ROUNDED sampling positions)
    gx, gy = grid[..., 0], grid[..., 1]
    ix = (gx + 1.0) * (W / 2.0) - 0.5
    iy = (gy + 1.0) * (H / 2.0) - 0.5
    x0 = torch.floor(ix)
    y0 = torch.floor(iy)
    tx = ix - x0
    ty = iy - y0
    x0 = x0.long()
    y0 = y0.long()
    out = torch.zeros(N, C, grid.shape[1], grid.shape[2], dtype=src.dtype)
    flat = src.reshape(N, C, H * W)
    taps = ((0, 0, (1 - ty) * (1 - tx)), (0, 1, (1 - ty) * tx), (1, 0, ty * (1 - tx)), (1, 1, ty * tx))
    for dy, dx, wgt in taps:
        xx = x0 + dx
        yy = y0 + dy
        ok = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
        idx = (yy.clamp(0, H - 1) * W + xx.clamp(0, W - 1)).view(N, 1, -1).expand(N, C, -1)
        v = torch.gather(flat, 2, idx).view(N, C, grid.shape[1], grid.shape[2])
        out = out + v * (wgt * ok.to(src.dtype)).unsqueeze(1)
    return out


def warp_affine_simple(src: torch.Tensor, theta: torch.Tensor, dsize: Tuple[int, int]) -> torch.Tensor:
    return grid_sample_bilinear_zeros(src, affine_grid_f64(theta, dsize[0], dsize[1]))


# --------------------------------------------------------------------------------------
# Row H / H': AttFusion, MaxFusion  (opencood/models/fuse_modules/fusion_in_one.py:47-136,
# ScaledDotProductAttention opencood/models/fuse_modules/att_fuse.py:43-47)
# --------------------------------------------------------------------------------------


def regroup(x: torch.Tensor, record_len: torch.Tensor) -> List[torch.Tensor]:
    """fusion_in_one.py:21-24."""
    ends = torch.cumsum(record_len, dim=0)
    return list(torch.tensor_split(x, ends[:-1].cpu()))


def att_fuse(x: torch.Tensor, record_len: torch.Tensor, norm_affine: torch.Tensor) -> torch.Tensor:
    """Per frame: warp all agents into ego (row ``[b, 0, :N]``), per-pixel softmax(X X^T / sqrt(C)) X,
    keep agent 0.  fusion_in_one.py:96-136."""
    _, C, H, W = x.shape
    outs = []
    for b, xb in enumerate(regroup(x, record_len)):
        n = int(record_len[b])
        theta = norm_affine[b, 0, :n]
        w = warp_affine_simple(xb, theta, (H, W))                 # [n, C, H, W]
        q = w.view(n, C, -1).permute(2, 0, 1)                     # [HW, n, C]
        score = torch.bmm(q, q.transpose(1, 2)) / np.sqrt(C)
        attn = F.softmax(score, -1)
        ctx = torch.bmm(attn, q)                                  # [HW, n, C]
        outs.append(ctx.permute(1, 2, 0).reshape(n, C, H, W)[0])
    return torch.stack(outs)


def max_fuse(x: torch.Tensor, record_len: torch.Tensor, norm_affine: torch.Tensor) -> torch.Tensor:
    """fusion_in_one.py:51-89: same warp, elementwise max over agents."""
    _, C, H, W = x.shape
    outs = []
    for b, xb in enumerate(regroup(x, record_len)):
        n = int(record_len[b])
        w = warp_affine_simple(xb, norm_affine[b, 0, :n], (H, W))
        outs.append(w.max(dim=0).values)
    return torch.stack(outs)


# --------------------------------------------------------------------------------------
# Rows D, E, I: dense CNN stages, functional, driven by a reference-named state_dict
# --------------------------------------------------------------------------------------


def _bn2d(x, sd, p, eps):
    return F.batch_norm(x, sd[p + "running_mean"], sd[p + "running_var"], sd[p + "weight"], sd[p + "bias"],
                        training=False, eps=eps)


def resnet_stages(x: torch.Tensor, sd, layer_nums, layer_strides, prefix="backbone.resnet.") -> List[torch.Tensor]:
    """resblock.py:53-69 (BasicBlock) and :212-221; BN eps 1e-5 (torch default, :38-39)."""
    feats = []
    for i, (n, s) in enumerate(zip(layer_nums, layer_strides)):
        for j in range(n):
            p = f"{prefix}layer{i}.{j}."
            stride = s if j == 0 else 1
            y = F.conv2d(x, sd[p + "conv1.weight"], None, stride=stride, padding=1)
            y = torch.relu(_bn2d(y, sd, p + "bn1.", 1e-5))
            y = F.conv2d(y, sd[p + "conv2.weight"], None, stride=1, padding=1)
            y = _bn2d(y, sd, p + "bn2.", 1e-5)
            if (p + "downsample.0.weight") in sd:
                idt = F.conv2d(x, sd[p + "downsample.0.weight"], None, stride=stride)
                idt = _bn2d(idt, sd, p + "downsample.1.", 1e-5)
            else:
                idt = x
            x = torch.relu(y + idt)
        feats.append(x)
    return feats


def plain_stages(x: torch.Tensor, sd, layer_nums, layer_strides, prefix="backbone.blocks.") -> List[torch.Tensor]:
    """base_bev_backbone.py:39-58: ZeroPad2d(1)+conv(s)+BN(1e-3)+ReLU then n x [conv+BN+ReLU]."""
    feats = []
    for i, (n, s) in enumerate(zip(layer_nums, layer_strides)):
        p = f"{prefix}{i}."
        x = F.conv2d(F.pad(x, (1, 1, 1, 1)), sd[p + "1.weight"], None, stride=s)
        x = torch.relu(_bn2d(x, sd, p + "2.", 1e-3))
        for k in range(n):
            ci = 4 + 3 * k
            x = F.conv2d(x, sd[p + f"{ci}.weight"], None, padding=1)
            x = torch.relu(_bn2d(x, sd, p + f"{ci + 1}.", 1e-3))
        feats.append(x)
    return feats


def deblocks_concat(feats: Sequence[torch.Tensor], sd, upsample_strides, prefix="backbone.deblocks.") -> torch.Tensor:
    """base_bev_backbone_resnet.py:121-138: ConvTranspose(k=s, stride=s)+BN(1e-3)+ReLU per scale, concat."""
    ups = []
    for i, (f, s) in enumerate(zip(feats, upsample_strides)):
        p = f"{prefix}{i}."
        y = F.conv_transpose2d(f, sd[p + "0.weight"], None, stride=s)
        ups.append(torch.relu(_bn2d(y, sd, p + "1.", 1e-3)))
    return torch.cat(ups, dim=1)


def shrink_and_heads(x: torch.Tensor, sd, use_dir=True) -> Dict[str, torch.Tensor]:
    """downsample_conv.py:47-50 (one DoubleConv) + heads point_pillar_baseline_multiscale.py:126-133."""
    if "shrink_conv.layers.0.double_conv.0.weight" in sd:
        p = "shrink_conv.layers.0.double_conv."
        x = torch.relu(F.conv2d(x, sd[p + "0.weight"], sd[p + "0.bias"], padding=1))
        x = torch.relu(F.conv2d(x, sd[p + "2.weight"], sd[p + "2.bias"], padding=1))
    out = {"cls_preds": F.conv2d(x, sd["cls_head.weight"], sd["cls_head.bias"]),
           "reg_preds": F.conv2d(x, sd["reg_head.weight"], sd["reg_head.bias"])}
    if "unc_head.weight" in sd:                     # stage-1 models, point_pillar_uncertainty.py:36-37,71
        out["unc_preds"] = F.conv2d(x, sd["unc_head.weight"], sd["unc_head.bias"])
    if use_dir and "dir_head.weight" in sd:
        out["dir_preds"] = F.conv2d(x, sd["dir_head.weight"], sd["dir_head.bias"])
    return out


def naive_compressor(x: torch.Tensor, sd, prefix: str = "naive_compressor.") -> torch.Tensor:
    """NaiveCompressor.forward, opencood/models/sub_modules/naive_compress.py:5-31: encoder = conv3x3 (64 -> 64 / r, bias) + BN(eps 1e-3)
    + ReLU; decoder = two more such triples back to 64 channels (Sequential indices 0-1, 0-1, 3-4)."""
    for conv, bn in (("encoder.0", "encoder.1"), ("decoder.0", "decoder.1"), ("decoder.3", "decoder.4")):
        x = F.conv2d(x, sd[prefix + conv + ".weight"], sd[prefix + conv + ".bias"], stride=1, padding=1)
        x = torch.relu(_bn2d(x, sd, prefix + bn + ".", 1e-3))
    return x


def coalign_forward(sd, margs: dict, batch: dict, return_intermediate: bool = False, dtype: torch.dtype = torch.float32):
    """PointPillarBaselineMultiscale.forward, opencood/models/point_pillar_baseline_multiscale.py:93-135.

    ``dtype=torch.float64`` (round 6): the SAME function of the same float32 inputs and float32 parameters evaluated in float64 -- the yardstick every float32
    arithmetic (the reference's own op-by-op fp32, this build's convolution modes) is measured against.  What stays as the reference defines it: the sampling grid
    is built in float64 and ROUNDED to float32 (torch_transformation_utils.py:328-330), i.e. the same sampling positions; everything else is exact to 2^-53."""
    pl = batch["processed_lidar"]
    rl = batch["record_len"]
    nx, ny, _ = [int(v) for v in margs["point_pillar_scatter"]["grid_size"]]
    if dtype != torch.float32:
        sd = {k: (v.to(dtype) if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in sd.items()}
    pf = pillar_vfe(pl["voxel_features"].to(dtype), pl["voxel_num_points"], pl["voxel_coords"], sd,
                    margs["voxel_size"], margs["lidar_range"])
    n_total = int(rl.sum())
    canvas = scatter(pf, pl["voxel_coords"], n_total, nx, ny)
    H0, W0 = canvas.shape[2:]
    aff = normalize_pairwise_tfm(batch["pairwise_t_matrix"], H0, W0, margs["voxel_size"][0])
    bb = margs["base_bev_backbone"]
    if "compression" in margs:                                    # point_pillar_baseline_multiscale.py:113-114
        canvas = naive_compressor(canvas, sd)
    if bb.get("resnet", True):
        feats = resnet_stages(canvas, sd, bb["layer_nums"], bb["layer_strides"])
    else:
        feats = plain_stages(canvas, sd, bb["layer_nums"], bb["layer_strides"])
    fuse = att_fuse if margs["fusion_method"] == "att" else max_fuse
    fused = [fuse(f, rl, aff) for f in feats]
    x = deblocks_concat(fused, sd, bb["upsample_strides"])
    out = shrink_and_heads(x, sd, use_dir="dir_args" in margs)
    if return_intermediate:
        out = dict(out, pillar_features=pf, spatial_features=canvas, affine=aff, feats=feats, fused=fused)
    return out


def pointpillar_forward(sd, margs: dict, batch: dict) -> Dict[str, torch.Tensor]:
    """PointPillar.forward (single agent / late fusion), opencood/models/point_pillar.py:52-80."""
    pl = batch["processed_lidar"]
    nx, ny, _ = [int(v) for v in margs["point_pillar_scatter"]["grid_size"]]
    pf = pillar_vfe(pl["voxel_features"], pl["voxel_num_points"], pl["voxel_coords"], sd,
                    margs["voxel_size"], margs["lidar_range"])
    n = int(pl["voxel_coords"][:, 0].max()) + 1
    canvas = scatter(pf, pl["voxel_coords"], n, nx, ny)
    bb = margs["base_bev_backbone"]
    if bb.get("resnet", False):
        feats = resnet_stages(canvas, sd, bb["layer_nums"], bb["layer_strides"])
    else:
        feats = plain_stages(canvas, sd, bb["layer_nums"], bb["layer_strides"])
    x = deblocks_concat(feats, sd, bb["upsample_strides"])
    return shrink_and_heads(x, sd, use_dir="dir_args" in margs)


# --------------------------------------------------------------------------------------
# Row J: anchors  (opencood/data_utils/post_processor/voxel_postprocessor.py:30-81)
# --------------------------------------------------------------------------------------


def generate_anchor_box(anchor_args: dict, order: str = "hwl") -> np.ndarray:
    W, H = anchor_args["W"], anchor_args["H"]
    yaws = [math.radians(r) for r in anchor_args["r"]]
    na = len(yaws)
    vw, vh = anchor_args["vw"], anchor_args["vh"]
    rng = anchor_args["cav_lidar_range"]
    fs = anchor_args.get("feature_stride", 2)
    x = np.linspace(rng[0] + vw, rng[3] - vw, W // fs)
    y = np.linspace(rng[1] + vh, rng[4] - vh, H // fs)
    cx, cy = np.meshgrid(x, y)
    cx = np.tile(cx[..., None], na)
    cy = np.tile(cy[..., None], na)
    cz = np.ones_like(cx) * -1.0
    w = np.ones_like(cx) * anchor_args["w"]
    l = np.ones_like(cx) * anchor_args["l"]
    h = np.ones_like(cx) * anchor_args["h"]
    r = np.ones_like(cx)
    for i, a in enumerate(yaws):
        r[..., i] = a
    if order == "hwl":
        return np.stack([cx, cy, cz, h, w, l, r], axis=-1)
    if order == "lhw":
        return np.stack([cx, cy, cz, l, h, w, r], axis=-1)
    raise ValueError(order)


# --------------------------------------------------------------------------------------
# Rows K, L: decode + geometry + filters
# --------------------------------------------------------------------------------------


def delta_to_boxes3d(deltas: torch.Tensor, anchors: torch.Tensor) -> torch.Tensor:
    """voxel_postprocessor.py:405-450.  deltas [N,7A,H,W]; anchors [H,W,A,7] (f64 ok) -> [N,H*W*A,7]."""
    N = deltas.shape[0]
    d = deltas.permute(0, 2, 3, 1).contiguous().view(N, -1, 7)
    a = anchors.view(-1, 7).float()
    diag = torch.sqrt(a[:, 4] ** 2 + a[:, 5] ** 2)
    out = torch.zeros_like(d)
    out[..., 0] = d[..., 0] * diag + a[:, 0]
    out[..., 1] = d[..., 1] * diag + a[:, 1]
    out[..., 2] = d[..., 2] * a[:, 3] + a[:, 2]
    out[..., 3:6] = torch.exp(d[..., 3:6]) * a[:, 3:6]
    out[..., 6] = d[..., 6] + a[:, 6]
    return out


def limit_period(val: torch.Tensor, offset: float = 0.5, period: float = 2 * np.pi) -> torch.Tensor:
    """opencood/utils/common_utils.py:70-79."""
    return val - torch.floor(val / period + offset) * period


def boxes_to_corners_3d(boxes3d: torch.Tensor, order: str) -> torch.Tensor:
    """opencood/utils/box_utils.py:152-204 (+ rotate_points_along_z common_utils.py:105-127)."""
    b = boxes3d[:, [0, 1, 2, 5, 4, 3, 6]] if order == "hwl" else boxes3d
    template = b.new_tensor([[1, -1, -1], [1, 1, -1], [-1, 1, -1], [-1, -1, -1],
                             [1, -1, 1], [1, 1, 1], [-1, 1, 1], [-1, -1, 1]]) / 2
    corners = b[:, None, 3:6].repeat(1, 8, 1) * template[None]
    cosa, sina = torch.cos(b[:, 6]), torch.sin(b[:, 6])
    zeros, ones = torch.zeros_like(cosa), torch.ones_like(cosa)
    rot = torch.stack((cosa, sina, zeros, -sina, cosa, zeros, zeros, zeros, ones), dim=1).view(-1, 3, 3).float()
    corners = torch.matmul(corners, rot)
    return corners + b[:, None, 0:3]


def project_box3d(box3d: torch.Tensor, T: torch.Tensor) -> torch.Tensor:
    """box_utils.py:278-316: homogeneous 4x4 applied to the 8 corners."""
    c = box3d.transpose(1, 2)
    c = torch.cat((c, torch.ones(c.shape[0], 1, 8, dtype=c.dtype)), dim=1)
    return torch.matmul(T, c)[:, :3, :].transpose(1, 2)


def remove_large_pred_bbx(c: torch.Tensor) -> torch.Tensor:
    """box_utils.py:840-869 incl. its quirk: the "z" extent is taken from the **y** column and used as a
    truthy mask (:862-867)."""
    x_len = c[:, :, 0].max(dim=1)[0] - c[:, :, 0].min(dim=1)[0]
    y_len = c[:, :, 1].max(dim=1)[0] - c[:, :, 1].min(dim=1)[0]
    z_len = c[:, :, 1].max(dim=1)[0] - c[:, :, 1].min(dim=1)[0]
    idx = torch.logical_and(x_len <= 6, y_len <= 6)
    return torch.logical_and(idx, z_len)


def remove_bbx_abnormal_z(c: torch.Tensor) -> torch.Tensor:
    """box_utils.py:872-890."""
    return torch.logical_and(c[:, :, 2].min(dim=1)[0] >= -3, c[:, :, 2].max(dim=1)[0] <= 1)


def mask_boxes_outside_range(corners: np.ndarray, limit_range: Sequence[float], min_num_corners: int = 8):
    """box_utils.py:384-421 for [K,8,3] input."""
    lim = np.asarray(limit_range)
    m = ((corners >= lim[0:3]) & (corners <= lim[3:6])).all(axis=2)
    return m.sum(axis=1) >= min_num_corners


def decode_candidates(cls_preds, reg_preds, dir_preds, anchors, score_thr: float, order: str,
                      dir_offset: float = 0.7853, num_bins: int = 2, T: Optional[torch.Tensor] = None):
    """voxel_postprocessor.py:291-355 for one agent: returns (flat indices that pass the score
    threshold in (h, w, anchor) order, boxes3d [K,7], scores [K], projected corners [K,8,3])."""
    prob = torch.sigmoid(cls_preds.permute(0, 2, 3, 1)).reshape(1, -1)
    boxes = delta_to_boxes3d(reg_preds, anchors)
    mask = torch.gt(prob, score_thr).view(-1)
    idx = torch.nonzero(mask).view(-1)
    boxes3d = boxes[0][mask].clone()
    scores = prob[0][mask]
    if dir_preds is not None and len(boxes3d):
        dm = dir_preds.permute(0, 2, 3, 1).contiguous().reshape(1, -1, num_bins)[0][mask]
        labels = torch.max(dm, dim=-1)[1]
        period = 2 * np.pi / num_bins
        rot = limit_period(boxes3d[..., 6] - dir_offset, 0, period)
        boxes3d[..., 6] = rot + dir_offset + period * labels.to(dm.dtype)
        boxes3d[..., 6] = limit_period(boxes3d[..., 6], 0.5, 2 * np.pi)
    if len(boxes3d) == 0:
        return idx, boxes3d, scores, torch.zeros(0, 8, 3)
    corners = boxes_to_corners_3d(boxes3d, order)
    if T is None:
        T = torch.eye(4)
    return idx, boxes3d, scores, project_box3d(corners, T.float())


def post_process(agents: Sequence[dict], anchors: torch.Tensor, pp_cfg: dict):
    """VoxelPostprocessor.post_process, voxel_postprocessor.py:243-402.

    ``agents``: one dict per cav in iteration order with keys cls_preds, reg_preds, [dir_preds],
    [transformation_matrix].  Returns (pred_box3d [K',8,3] f32 | None, scores | None, info dict)."""
    thr = pp_cfg["target_args"]["score_threshold"]
    order = pp_cfg["order"]
    da = pp_cfg.get("dir_args", {})
    all_c, all_s, all_i = [], [], []
    for ag in agents:
        idx, _, scores, corners = decode_candidates(
            ag["cls_preds"], ag["reg_preds"], ag.get("dir_preds"), anchors, thr, order,
            da.get("dir_offset", 0.7853), da.get("num_bins", 2), ag.get("transformation_matrix"))
        if len(scores):
            all_c.append(corners)
            all_s.append(scores)
            all_i.append(idx)
    if not all_c:
        return None, None, {}
    corners = torch.vstack(all_c)
    scores = torch.cat(all_s)
    keep1 = torch.logical_and(remove_large_pred_bbx(corners), remove_bbx_abnormal_z(corners))
    c1, s1 = corners[keep1], scores[keep1]
    keep = nms_rotated(c1.numpy(), s1.numpy(), pp_cfg["nms_thresh"])
    c2, s2 = c1[keep], s1[keep]
    m = mask_boxes_outside_range(c2.numpy(), pp_cfg["gt_range"])
    info = dict(cand_index=torch.cat(all_i), cand_corners=corners, cand_scores=scores, keep_filter=keep1,
                keep_nms=keep, keep_range=m)
    return c2[torch.from_numpy(m)], s2[torch.from_numpy(m)], info


# --------------------------------------------------------------------------------------
# Row M: rotated NMS  (opencood/utils/box_utils.py:693-738; IoU opencood/utils/common_utils.py:196-236)
# --------------------------------------------------------------------------------------

_NMS_LIB = None


def build_c(force: bool = False) -> str:
    """Compile oracle/rotated_nms.c + oracle/voxelize.c -> oracle/_build/librotated_nms.so with gcc."""
    out_dir = os.path.join(_HERE, "_build")
    so = os.path.join(out_dir, "librotated_nms.so")
    srcs = [os.path.join(_HERE, "rotated_nms.c"), os.path.join(_HERE, "voxelize.c")]
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in srcs):
        os.makedirs(out_dir, exist_ok=True)
        subprocess.check_call(["gcc", "-O2", "-std=c99", "-ffp-contract=off", "-shared", "-fPIC", *srcs, "-o", so, "-lm"])
    return so


def _nms_lib():
    global _NMS_LIB
    if _NMS_LIB is None:
        lib = ctypes.CDLL(build_c())
        lib.oracle_quad_iou.restype = ctypes.c_double
        lib.oracle_quad_iou.argtypes = [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
        lib.oracle_quad_intersection_area.restype = ctypes.c_double
        lib.oracle_quad_intersection_area.argtypes = [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
        lib.oracle_nms_rotated.restype = ctypes.c_int
        lib.oracle_nms_rotated.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                           ctypes.c_float, ctypes.c_int, ctypes.c_void_p]
        lib.oracle_pcdet_overlap.restype = ctypes.c_float
        lib.oracle_pcdet_overlap.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        lib.oracle_pcdet_nms.restype = ctypes.c_int
        lib.oracle_pcdet_nms.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_void_p]
        lib.oracle_points_to_voxel.restype = ctypes.c_int
        lib.oracle_points_to_voxel.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                               ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                               ctypes.c_void_p]
        _NMS_LIB = lib
    return _NMS_LIB


def quad_iou(a_xy: np.ndarray, b_xy: np.ndarray) -> float:
    """fp64 IoU of two convex quads given as [4,2] arrays (C implementation)."""
    a = np.ascontiguousarray(a_xy, dtype=np.float64)
    b = np.ascontiguousarray(b_xy, dtype=np.float64)
    return _nms_lib().oracle_quad_iou(a.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                                      b.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))


def quad_intersection_area(a_xy: np.ndarray, b_xy: np.ndarray) -> float:
    a = np.ascontiguousarray(a_xy, dtype=np.float64)
    b = np.ascontiguousarray(b_xy, dtype=np.float64)
    return _nms_lib().oracle_quad_intersection_area(a.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                                                    b.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))


def score_order(scores: np.ndarray, top: int = 1000) -> np.ndarray:
    """``scores.argsort()[::-1][:top]`` (box_utils.py:719-721).  numpy's default sort is unstable, so the
    order among *equal* scores is implementation-defined in the reference; this oracle (and the HIP
    path) define it as stable-ascending-then-reversed, i.e. ties in descending index order."""
    return np.argsort(scores, kind="stable")[::-1][:top]


def nms_rotated(boxes: np.ndarray, scores: np.ndarray, threshold: float, top: int = 1000) -> np.ndarray:
    """boxes [K,8,3] or [K,4,2] f32, scores [K] f32 -> int32 indices in pick order (C implementation)."""
    K = boxes.shape[0]
    if K == 0:
        return np.array([], dtype=np.int32)
    b = np.ascontiguousarray(boxes, dtype=np.float32)
    s = np.ascontiguousarray(scores, dtype=np.float32)
    stride = b.shape[1] * b.shape[2]
    keep = np.empty(min(K, top), dtype=np.int32)
    n = _nms_lib().oracle_nms_rotated(b.ctypes.data, b.shape[2], s.ctypes.data, K, ctypes.c_float(threshold),
                                      top, keep.ctypes.data) if stride else 0
    return keep[:n].copy()


def nms_rotated_numpy(boxes: np.ndarray, scores: np.ndarray, threshold: float, top: int = 1000) -> np.ndarray:
    """Pure-python restatement of box_utils.py:693-738 using :func:`quad_iou_python` (small cases only)."""
    if boxes.shape[0] == 0:
        return np.array([], dtype=np.int32)
    polys = [np.asarray(b[:4, :2], dtype=np.float64) for b in boxes]
    ixs = score_order(scores, top)
    pick = []
    while len(ixs) > 0:
        i = ixs[0]
        pick.append(i)
        iou = np.array([quad_iou_python(polys[i], polys[j]) for j in ixs[1:]], dtype=np.float32)
        remove = np.where(iou > threshold)[0] + 1
        ixs = np.delete(ixs, remove)
        ixs = np.delete(ixs, 0)
    return np.array(pick, dtype=np.int32)


def _shoelace(p: np.ndarray) -> float:
    x, y = p[:, 0], p[:, 1]
    return 0.5 * float(np.sum(x * np.roll(y, -1) - np.roll(x, -1) * y))


def quad_iou_python(a: np.ndarray, b: np.ndarray) -> float:
    """Independent second algorithm for cross-checking the C clipper: collect (i) vertices of each quad
    inside the other and (ii) all edge-edge intersection points, order them by angle around their centroid,
    shoelace.  fp64 throughout.  (Same idea as the OpenPCDet box_overlap but without its 1e-2 margins.)"""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    area_a, area_b = abs(_shoelace(a)), abs(_shoelace(b))

    def inside(p, q):
        s = np.sign(_shoelace(q)) or 1.0
        for k in range(4):
            e = q[(k + 1) % 4] - q[k]
            if s * (e[0] * (p[1] - q[k][1]) - e[1] * (p[0] - q[k][0])) < 0:
                return False
        return True

    pts = [p for p in a if inside(p, b)] + [p for p in b if inside(p, a)]
    for i in range(4):
        p1, p2 = a[i], a[(i + 1) % 4]
        for j in range(4):
            q1, q2 = b[j], b[(j + 1) % 4]
            r, s = p2 - p1, q2 - q1
            den = r[0] * s[1] - r[1] * s[0]
            if den == 0:
                continue
            t = ((q1[0] - p1[0]) * s[1] - (q1[1] - p1[1]) * s[0]) / den
            u = ((q1[0] - p1[0]) * r[1] - (q1[1] - p1[1]) * r[0]) / den
            if 0 <= t <= 1 and 0 <= u <= 1:
                pts.append(p1 + t * r)
    if len(pts) < 3:
        inter = 0.0
    else:
        P = np.array(pts)
        c = P.mean(axis=0)
        ang = np.arctan2(P[:, 1] - c[1], P[:, 0] - c[0])
        inter = abs(_shoelace(P[np.argsort(ang)]))
    union = area_a + area_b - inter
    return inter / union if union != 0 else float("nan")


# --------------------------------------------------------------------------------------
# Row N: OpenPCDet fp32 BEV IoU / NMS  (opencood/pcdet_utils/iou3d_nms/src/iou3d_cpu.cpp:128-252,
# iou3d_nms_kernel.cu:104-311, iou3d_nms.cpp:90-136) -- C implementation in rotated_nms.c
# --------------------------------------------------------------------------------------


def pcdet_overlap(box_a: np.ndarray, box_b: np.ndarray) -> float:
    a = np.ascontiguousarray(box_a, dtype=np.float32)
    b = np.ascontiguousarray(box_b, dtype=np.float32)
    return float(_nms_lib().oracle_pcdet_overlap(a.ctypes.data, b.ctypes.data))


def pcdet_nms(boxes7: np.ndarray, scores: np.ndarray, thr: float, pre_max: Optional[int] = None) -> np.ndarray:
    """iou3d_nms_utils.nms_gpu (:255-271): sort by score desc, bitmask NMS on fp32 BEV IoU, returns indices
    into the original array."""
    order = np.argsort(-scores, kind="stable")
    if pre_max is not None:
        order = order[:pre_max]
    b = np.ascontiguousarray(boxes7[order], dtype=np.float32)
    keep = np.empty(len(order), dtype=np.int32)
    n = _nms_lib().oracle_pcdet_nms(b.ctypes.data, len(order), ctypes.c_float(thr), keep.ctypes.data)
    return order[keep[:n]]


def pcdet_nms_normal(boxes7: np.ndarray, scores: np.ndarray, thr: float) -> np.ndarray:
    """iou3d_nms_utils.nms_normal_gpu (:274-289): heading ignored, axis-aligned IoU (iou3d_nms_kernel.cu:313-325)."""
    order = np.argsort(-scores, kind="stable")
    b = np.ascontiguousarray(boxes7[order], dtype=np.float32)
    keep = np.empty(len(order), dtype=np.int32)
    lib = _nms_lib()
    lib.oracle_pcdet_nms_normal.restype = ctypes.c_int
    lib.oracle_pcdet_nms_normal.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_void_p]
    n = lib.oracle_pcdet_nms_normal(b.ctypes.data, len(order), ctypes.c_float(thr), keep.ctypes.data)
    return order[keep[:n]]


def pcdet_min_margin(boxes7: np.ndarray, thr: float, normal: bool = False) -> float:
    """min |IoU - thr| over the overlapping pairs (test helper, see rotated_nms.c)."""
    b = np.ascontiguousarray(boxes7, dtype=np.float32)
    lib = _nms_lib()
    lib.oracle_pcdet_min_margin.restype = ctypes.c_float
    lib.oracle_pcdet_min_margin.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_int]
    return float(lib.oracle_pcdet_min_margin(b.ctypes.data, len(b), ctypes.c_float(thr), int(normal)))


# --------------------------------------------------------------------------------------
# Evaluation: opencood/utils/eval_utils.py:19-169
# --------------------------------------------------------------------------------------


def iou_matrix(det: np.ndarray, gt: np.ndarray) -> np.ndarray:
    """float32 IoU of every (detection, ground truth) pair, corners 0..3 (x, y) as polygons (common_utils.py:196-236)."""
    out = np.zeros((len(det), len(gt)), dtype=np.float32)
    for i, a in enumerate(det):
        for j, b in enumerate(gt):
            out[i, j] = np.float32(quad_iou(a[:4, :2], b[:4, :2]))
    return out


def caluclate_tp_fp(det_boxes, det_score, gt_boxes, result_stat, iou_thresh):
    """eval_utils.py:45-96: detections in descending score order; TP if the best IoU against the remaining ground truths
    is >= the threshold, the matched ground truth is removed."""
    fp, tp = [], []
    gt = gt_boxes.shape[0]
    if det_boxes is not None:
        det_boxes, det_score, gt_np = np.asarray(det_boxes), np.asarray(det_score), np.asarray(gt_boxes)
        order = np.argsort(-det_score, kind="stable")
        det_score = det_score[order]
        remaining = list(range(gt))
        for i in range(len(order)):
            ious = np.array([quad_iou(det_boxes[order[i]][:4, :2], gt_np[g][:4, :2]) for g in remaining], dtype=np.float32)
            if len(remaining) == 0 or np.max(ious) < iou_thresh:
                fp.append(1); tp.append(0)
                continue
            fp.append(0); tp.append(1)
            remaining.pop(int(np.argmax(ious)))
        result_stat[iou_thresh]["score"] += det_score.tolist()
    result_stat[iou_thresh]["fp"] += fp
    result_stat[iou_thresh]["tp"] += tp
    result_stat[iou_thresh]["gt"] += gt


def voc_ap(rec, prec):
    """eval_utils.py:19-42."""
    mrec = [0.0] + list(rec) + [1.0]
    mpre = [0.0] + list(prec) + [0.0]
    for i in range(len(mpre) - 2, -1, -1):
        mpre[i] = max(mpre[i], mpre[i + 1])
    ap = 0.0
    for i in range(1, len(mrec)):
        if mrec[i] != mrec[i - 1]:
            ap += (mrec[i] - mrec[i - 1]) * mpre[i]
    return ap, mrec, mpre


def calculate_ap(result_stat, iou):
    """eval_utils.py:100-142."""
    st = result_stat[iou]
    fp, tp, score = np.array(st["fp"]), np.array(st["tp"]), np.array(st["score"])
    order = np.argsort(-score, kind="stable")
    fp, tp = np.cumsum(fp[order]), np.cumsum(tp[order])
    rec = [float(t) / st["gt"] for t in tp]
    prec = [float(t) / (f + t) for t, f in zip(tp, fp)]
    return voc_ap(rec, prec)


def generate_gt_bbx(data_dict: dict, order: str, gt_range) -> torch.Tensor:
    """opencood/data_utils/post_processor/base_postprocessor.py:46-106: per agent, masked object centres -> corners ->
    projected with the clean transform; duplicates removed by object id (first occurrence, ids visited in ``set`` order);
    kept only if all 8 corners lie inside ``gt_range`` including z (box_utils.py:384-421, compared in float64 after numpy
    promotes the float32 corners against the Python-float range)."""
    boxes, ids = [], []
    for cav in data_dict.values():
        centre = cav["object_bbx_center"][cav["object_bbx_mask"] == 1]
        boxes.append(project_box3d(boxes_to_corners_3d(centre, order).float(), cav["transformation_matrix_clean"]))
        ids += list(cav["object_ids"])
    boxes = torch.vstack(boxes)
    boxes = boxes[[ids.index(x) for x in set(ids)]].numpy()
    lo, hi = np.asarray(gt_range[0:3]), np.asarray(gt_range[3:6])
    inside = ((boxes >= lo) & (boxes <= hi)).all(axis=2).sum(axis=1) >= 8
    return torch.from_numpy(boxes[inside])


# ---------------------------------------------------------------------------------------------------------------------
# points -> pillars.  spconv is a third-party dependency absent from /root/reference and from this
# image: PARITY UNPINNED for the voxel generator itself (see oracle/voxelize.c); the two point filters are the reference's
# own numpy code and are pinned by tests/golden/points.npz.

def mask_points_by_range(points: np.ndarray, limit_range) -> np.ndarray:
    """opencood/utils/pcd_utils.py:41-66: strict inequalities on x, y, z; float32 points against Python floats compare in
    float32 (NumPy weak-scalar promotion)."""
    lo, hi = np.asarray(limit_range[0:3], dtype=np.float32), np.asarray(limit_range[3:6], dtype=np.float32)
    m = np.ones(len(points), dtype=bool)
    for j in range(3):
        m &= (points[:, j] > lo[j]) & (points[:, j] < hi[j])
    return points[m]


def mask_ego_points(points: np.ndarray) -> np.ndarray:
    """pcd_utils.py:69-88: drop the returns from the ego vehicle's own body (closed box in x, y)."""
    f = np.float32
    m = (points[:, 0] >= f(-1.95)) & (points[:, 0] <= f(2.95)) & (points[:, 1] >= f(-1.1)) & (points[:, 1] <= f(1.1))
    return points[~m]


def voxel_grid_size(voxel_size, lidar_range) -> np.ndarray:
    """sp_voxel_preprocessor.py:40-42 / spconv: round((max - min) / voxel) in float32, (x, y, z)."""
    r, v = np.asarray(lidar_range, dtype=np.float32), np.asarray(voxel_size, dtype=np.float32)
    return np.round((r[3:6] - r[0:3]) / v).astype(np.int32)


def points_to_voxel(points: np.ndarray, voxel_size, lidar_range, max_points: int, max_voxels: int):
    """C restatement of spconv's sequential generator -> (voxels [M, max_points, 4], coords [M, 3] zyx int32, num [M] int32)."""
    pts = np.ascontiguousarray(points, dtype=np.float32)
    assert pts.ndim == 2 and pts.shape[1] == 4
    vs, rg = np.asarray(voxel_size, dtype=np.float32), np.asarray(lidar_range, dtype=np.float32)
    grid = voxel_grid_size(voxel_size, lidar_range)
    cap = min(len(pts), max_voxels)
    voxels = np.zeros((cap, max_points, 4), dtype=np.float32)
    coors = np.zeros((cap, 3), dtype=np.int32)
    num = np.zeros(cap, dtype=np.int32)
    scratch = np.empty(int(grid[0]) * int(grid[1]) * int(grid[2]), dtype=np.int32)
    m = _nms_lib().oracle_points_to_voxel(pts.ctypes.data, len(pts), vs.ctypes.data, rg.ctypes.data, grid.ctypes.data, max_points,
                                          max_voxels, voxels.ctypes.data, coors.ctypes.data, num.ctypes.data, scratch.ctypes.data)
    return voxels[:m], coors[:m], num[:m]


def points_to_voxel_python(points: np.ndarray, voxel_size, lidar_range, max_points: int, max_voxels: int):
    """The same loop in plain Python (small inputs only) -- cross-checks the C build."""
    vs, rg = np.asarray(voxel_size, dtype=np.float32), np.asarray(lidar_range, dtype=np.float32)
    grid = voxel_grid_size(voxel_size, lidar_range)
    table, voxels, coors, num = {}, [], [], []
    for p in np.asarray(points, dtype=np.float32):
        c = np.floor((p[:3] - rg[:3]) / vs)
        if not np.all((c >= 0) & (c < grid)):
            continue
        key = (int(c[2]), int(c[1]), int(c[0]))
        v = table.get(key)
        if v is None:
            if len(voxels) >= max_voxels:
                continue
            v = table[key] = len(voxels)
            voxels.append(np.zeros((max_points, 4), dtype=np.float32)); coors.append(key); num.append(0)
        if num[v] < max_points:
            voxels[v][num[v]] = p
            num[v] += 1
    if not voxels:
        return np.zeros((0, max_points, 4), np.float32), np.zeros((0, 3), np.int32), np.zeros(0, np.int32)
    return np.stack(voxels), np.asarray(coors, dtype=np.int32), np.asarray(num, dtype=np.int32)


def collate_voxels(per_cloud):
    """SpVoxelPreprocessor.collate_batch_list (sp_voxel_preprocessor.py:107-147): concatenate and prefix the cloud index."""
    feats = np.concatenate([v for v, _, _ in per_cloud])
    num = np.concatenate([n for _, _, n in per_cloud])
    coords = np.concatenate([np.pad(c, ((0, 0), (1, 0)), mode="constant", constant_values=i) for i, (_, c, _) in enumerate(per_cloud)])
    return feats, coords, num


# ---------------------------------------------------------------------------------------------------------------------
# Pose correction by box alignment: agent-object pose graph built from every agent's stage-1 boxes,
# solved with Levenberg-Marquardt.  The solver in the reference is g2o (python binding `g2o`, C++; absent from
# /root/reference and from this image, no version pinned): PARITY UNPINNED for the optimiser -- restated below from
# g2o's published OptimizationAlgorithmLevenberg / EdgeSE2 / EdgeSE2PointXY, and cross-checked in the tests against an
# independent solver (scipy.optimize.least_squares) on the same residuals.  Everything in front of and behind the solver
# (box clustering, landmark choice, information matrices, hard-case rules, degree/radian handling) is the reference's own
# Python and IS pinned: tests/golden/box_align.npz comes from the reference's function running with a recording g2o stand-in.

def pose_to_tfm(pose) -> np.ndarray:
    """opencood/utils/transformation_utils.py:93-160.  numpy input is cast to **float32** first (common_utils.py:82-85)."""
    p = torch.from_numpy(np.asarray(pose)).float()
    n = p.shape[0]
    tfm = torch.eye(4).view(1, 4, 4).repeat(n, 1, 1)
    if p.shape[1] == 3:
        yaw = torch.deg2rad(p[:, 2])
        tfm[:, 0, 0], tfm[:, 0, 1], tfm[:, 1, 0], tfm[:, 1, 1] = torch.cos(yaw), -torch.sin(yaw), torch.sin(yaw), torch.cos(yaw)
        tfm[:, 0, 3], tfm[:, 1, 3] = p[:, 0], p[:, 1]
        return tfm.numpy()
    cy, sy = torch.cos(torch.deg2rad(p[:, 4])), torch.sin(torch.deg2rad(p[:, 4]))
    cr, sr = torch.cos(torch.deg2rad(p[:, 3])), torch.sin(torch.deg2rad(p[:, 3]))
    cp, sp = torch.cos(torch.deg2rad(p[:, 5])), torch.sin(torch.deg2rad(p[:, 5]))
    tfm[:, 0, 3], tfm[:, 1, 3], tfm[:, 2, 3] = p[:, 0], p[:, 1], p[:, 2]
    tfm[:, 0, 0], tfm[:, 0, 1], tfm[:, 0, 2] = cp * cy, cy * sp * sr - sy * cr, -cy * sp * cr - sy * sr
    tfm[:, 1, 0], tfm[:, 1, 1], tfm[:, 1, 2] = sy * cp, sy * sp * sr + cy * cr, -sy * sp * cr + cy * sr
    tfm[:, 2, 0], tfm[:, 2, 1], tfm[:, 2, 2] = sp, -cp * sr, cp * cr
    return tfm.numpy()


def corner_to_center(corner3d: np.ndarray, order: str = "lwh") -> np.ndarray:
    """opencood/utils/box_utils.py:25-85: centre = mean of corners 0, 3, 5, 6; edge lengths and heading averaged over 4 edges."""
    c = np.asarray(corner3d)
    xyz = np.mean(c[:, [0, 3, 5, 6], :], axis=1)
    h = abs(np.mean(c[:, 4:, 2] - c[:, :4, 2], axis=1, keepdims=True))
    d = lambda i, j: np.sqrt(np.sum((c[:, i, [0, 1]] - c[:, j, [0, 1]]) ** 2, axis=1, keepdims=True))
    l = (d(0, 3) + d(2, 1) + d(4, 7) + d(5, 6)) / 4
    w = (d(0, 1) + d(2, 3) + d(4, 5) + d(6, 7)) / 4
    a = lambda i, j: np.arctan2(c[:, i, 1] - c[:, j, 1], c[:, i, 0] - c[:, j, 0])
    theta = (a(1, 2) + a(0, 3) + a(5, 6) + a(4, 7))[:, np.newaxis] / 4
    parts = [xyz, l, w, h, theta] if order == "lwh" else [xyz, h, w, l, theta]
    return np.concatenate(parts, axis=1).reshape(c.shape[0], 7)


def project_box3d_np(box3d: np.ndarray, T: np.ndarray) -> np.ndarray:
    """box_utils.project_box3d on numpy input: both operands are cast to float32 (common_utils.py:82-85)."""
    return project_box3d(torch.from_numpy(np.asarray(box3d)).float(), torch.from_numpy(np.asarray(T)).float()).numpy()


def all_pair_l2(A: np.ndarray, B: np.ndarray) -> np.ndarray:
    """box_align_v2.py:79-96, in the dtype of its inputs (float32 on this path)."""
    two_ab = 2 * A @ B.T
    with np.errstate(invalid="ignore"):          # rounding can push a zero distance slightly negative -> NaN, like the reference
        return np.sqrt(np.sum(A * A, 1, keepdims=True).repeat(two_ab.shape[1], axis=1)
                       + np.sum(B * B, 1, keepdims=True).T.repeat(two_ab.shape[0], axis=0) - two_ab)


def build_pose_graph(pred_corners_list, noisy_lidar_pose, uncertainty_list=None, landmark_SE2=True, adaptive_landmark=False,
                     normalize_uncertainty=False, abandon_hard_cases=False, drop_hard_boxes=False, drop_unsure_edge=False,
                     use_uncertainty=True, thres=1.5, yaw_var_thres=0.2):
    """box_align_v2.py:150-372 up to (not including) the optimiser call.  Returns None when the hard-case rules say "keep the
    noisy poses", else a dict: vertices [V, 3] float64 (agents first: x, y, yaw in radians; landmarks: x, y, yaw or x, y, 0),
    kinds [V] (0 fixed SE2, 1 free SE2, 2 free XY), edges = (agent [E], landmark [E], measurement [E, 3], information [E, 3]),
    clusters (list of box-index lists, seed first)."""
    if not use_uncertainty:
        uncertainty_list = None
    noisy_lidar_pose = np.asarray(noisy_lidar_pose)
    N = noisy_lidar_pose.shape[0]
    tfm = pose_to_tfm(noisy_lidar_pose)
    nonempty = [i for i, c in enumerate(pred_corners_list) if len(c) != 0]
    world_corners = [project_box3d_np(pred_corners_list[i], tfm[i]) for i in nonempty]
    box_local = np.concatenate([corner_to_center(c, "lwh") for c in pred_corners_list if len(c) != 0], axis=0)
    box_world = [corner_to_center(c, "lwh") for c in world_corners]
    center_world = np.concatenate([b[:, :3] for b in box_world], axis=0)
    yaw_world = np.concatenate([b[:, 6] for b in box_world], axis=0)
    pred_len = [len(c) for c in pred_corners_list]
    box_to_agent = [i for i in range(N) for _ in range(pred_len[i])]
    certainty = None
    if uncertainty_list is not None:
        certainty = np.exp(-np.concatenate([u for u in uncertainty_list if len(u) != 0], axis=0))
        certainty[:, :2] /= 1.6 ** 2 + 3.9 ** 2                      # anchor diagonal squared (:187-202)
        if normalize_uncertainty:
            certainty = np.sqrt(certainty)
    dist = all_pair_l2(center_world, center_world)
    cum = 0
    for i in range(N):
        dist[cum: cum + pred_len[i], cum: cum + pred_len[i]] = 10000  # boxes of one agent never pair up
        cum += pred_len[i]
    remain = set(range(cum))
    clusters, landmarks, varies = [], [], []
    for box_idx in range(cum):
        if box_idx not in remain:
            continue
        near = (dist[box_idx] < thres).nonzero()[0].tolist()
        if len(near) == 0:
            continue                                              # stays in `remain` (it has no neighbours anyway)
        # the reference's growth loop re-reads the SEED's row (:247), so a cluster is the seed plus its free neighbours
        members = [box_idx] + [j for j in near if j in remain]
        if len(members) == 1:
            remain.remove(box_idx)
            continue
        yaw_var = np.var([yaw_world[j] for j in members])
        if landmark_SE2 and not (adaptive_landmark and yaw_var > yaw_var_thres):
            lm = center_world[box_idx].copy()
            lm[2] = yaw_world[box_idx]
        else:
            lm = center_world[box_idx][:2]
            if landmark_SE2:
                for j in members:
                    certainty[j] *= 2
        clusters.append(members)
        landmarks.append(lm)
        varies.append(bool(yaw_var > yaw_var_thres))
        for j in members:
            remain.remove(j)
    L = len(clusters)
    if abandon_hard_cases and (L <= 3 or sum(varies) >= 0.5 * L):
        return None
    vertices = np.zeros((N + L, 3), dtype=np.float64)
    kinds = np.ones(N + L, dtype=np.int32)
    vertices[:N] = noisy_lidar_pose[:, [0, 1, 4]]
    vertices[:N, 2] = np.deg2rad(vertices[:N, 2])
    kinds[0] = 0
    e_agent, e_lm, e_meas, e_info = [], [], [], []
    for k, (members, lm) in enumerate(zip(clusters, landmarks)):
        se2 = lm.shape[0] == 3
        vertices[N + k, : lm.shape[0]] = lm
        kinds[N + k] = 1 if se2 else 2
        if drop_hard_boxes and varies[k]:
            continue
        for j in members:
            info = np.ones(3)
            if certainty is not None:
                if drop_unsure_edge and sum(certainty[j]) < 100:
                    continue
                info[: 3 if se2 else 2] = certainty[j][: 3 if se2 else 2]
            meas = box_local[j][[0, 1, 6]].astype(np.float64)
            if not se2:
                meas[2], info[2] = 0.0, 0.0
            e_agent.append(box_to_agent[j]); e_lm.append(N + k); e_meas.append(meas); e_info.append(info)
    edges = (np.asarray(e_agent, dtype=np.int32), np.asarray(e_lm, dtype=np.int32),
             np.asarray(e_meas, dtype=np.float64).reshape(-1, 3), np.asarray(e_info, dtype=np.float64).reshape(-1, 3))
    return {"vertices": vertices, "kinds": kinds, "edges": edges, "clusters": clusters, "n_agents": N}


def _normalize_theta(t):
    return np.arctan2(np.sin(t), np.cos(t))


def pose_graph_residuals(vertices, kinds, edges):
    """g2o EdgeSE2::computeError (e = toVector(M^-1 * (X1^-1 * X2))) / EdgeSE2PointXY (e = X1^-1 * l - m), stacked [E, 3]
    (third component 0 for point landmarks)."""
    ea, el, meas, _ = edges
    a, l = vertices[ea], vertices[el]
    c1, s1 = np.cos(a[:, 2]), np.sin(a[:, 2])
    dx, dy = l[:, 0] - a[:, 0], l[:, 1] - a[:, 1]
    rx, ry = c1 * dx + s1 * dy, -s1 * dx + c1 * dy                  # R1^T (t2 - t1)
    se2 = kinds[el] == 1
    cm, sm = np.where(se2, np.cos(meas[:, 2]), 1.0), np.where(se2, np.sin(meas[:, 2]), 0.0)
    ux, uy = rx - meas[:, 0], ry - meas[:, 1]
    e = np.zeros((len(ea), 3))
    e[:, 0], e[:, 1] = cm * ux + sm * uy, -sm * ux + cm * uy
    e[:, 2] = np.where(se2, _normalize_theta(l[:, 2] - a[:, 2] - meas[:, 2]), 0.0)
    return e


def pose_graph_chi2(vertices, kinds, edges):
    e = pose_graph_residuals(vertices, kinds, edges)
    return float(np.sum(e * e * edges[3]))


def pose_graph_lm(vertices, kinds, edges, max_iterations: int = 1000):
    """g2o's SparseOptimizer::optimize with OptimizationAlgorithmLevenberg over a dense solve, restated:
    lambda0 = 1e-5 * max diag(H); each iteration tries up to 10 damped steps  (H + lambda I) dx = b,  applies
    X <- X * dx for SE2 vertices (t += R dt, theta += dtheta, normalised) and l += dl for points,
    rho = (chi2 - chi2_new) / (dx . (lambda dx + b) + 1e-3); accepted steps scale lambda by
    clamp(1 - (2 rho - 1)^3, 1/3, 2/3), rejected ones multiply it by ni = 2, 4, 8...; stop on rho == 0, ten rejections in
    a row or max_iterations.  Vertices without edges are left untouched (g2o's active set).  Returns (vertices, stats)."""
    x = np.array(vertices, dtype=np.float64)
    ea, el, meas, info = edges
    E, V = len(ea), len(x)
    active = np.zeros(V, dtype=bool)
    active[ea] = True; active[el] = True
    free = np.nonzero(active & (kinds != 0))[0]
    if E == 0 or len(free) == 0:
        return x, {"iterations": 0, "chi2": 0.0, "chi2_initial": 0.0}
    dim = np.where(kinds[free] == 2, 2, 3)
    start = np.concatenate([[0], np.cumsum(dim)])
    col = {int(v): int(s) for v, s in zip(free, start[:-1])}
    n = int(start[-1])

    def linearize(x):
        H, b = np.zeros((n, n)), np.zeros(n)
        e = pose_graph_residuals(x, kinds, edges)
        for k in range(E):
            a, l = x[ea[k]], x[el[k]]
            se2 = kinds[el[k]] == 1
            c1, s1 = np.cos(a[2]), np.sin(a[2])
            cm, sm = (np.cos(meas[k, 2]), np.sin(meas[k, 2])) if se2 else (1.0, 0.0)
            RmT = np.array([[cm, sm], [-sm, cm]])
            dRT1 = np.array([[-s1, c1], [-c1, -s1]])
            d = l[:2] - a[:2]
            A = np.zeros((3, 3)); B = np.zeros((3, 3))
            A[:2, :2] = -RmT
            A[:2, 2] = RmT @ dRT1 @ d
            R1T = np.array([[c1, s1], [-s1, c1]])
            if se2:
                c2, s2 = np.cos(l[2]), np.sin(l[2])
                B[:2, :2] = RmT @ R1T @ np.array([[c2, -s2], [s2, c2]])
                A[2, 2], B[2, 2] = -1.0, 1.0
            else:
                B[:2, :2] = R1T
            W = np.diag(info[k])
            blocks = [(ea[k], A, 3), (el[k], B, 3 if se2 else 2)]
            for vi, Ji, di in blocks:
                if kinds[vi] == 0:
                    continue
                ci = col[int(vi)]
                b[ci: ci + di] -= (Ji.T @ W @ e[k])[:di]
                for vj, Jj, dj in blocks:
                    if kinds[vj] == 0:
                        continue
                    cj = col[int(vj)]
                    H[ci: ci + di, cj: cj + dj] += (Ji.T @ W @ Jj)[:di, :dj]
        return H, b, float(np.sum(e * e * info))

    def apply(x, dx):
        y = x.copy()
        for v, s, d in zip(free, start[:-1], dim):
            if d == 3:
                c, s_ = np.cos(y[v, 2]), np.sin(y[v, 2])
                y[v, 0] += c * dx[s] - s_ * dx[s + 1]
                y[v, 1] += s_ * dx[s] + c * dx[s + 1]
                y[v, 2] = _normalize_theta(y[v, 2] + dx[s + 2])
            else:
                y[v, :2] += dx[s: s + 2]
        return y

    lam, it = 0.0, 0
    chi0 = None
    for it in range(max_iterations):
        H, b, chi = linearize(x)
        if it == 0:
            lam, chi0 = 1e-5 * float(np.max(np.diag(H))), chi
        ni, rho, qmax = 2.0, 0.0, 0
        while True:
            try:
                dx = np.linalg.solve(H + lam * np.eye(n), b)
                y = apply(x, dx)
                new = pose_graph_chi2(y, kinds, edges)
                ok = np.all(np.isfinite(dx))
            except np.linalg.LinAlgError:
                ok, new, dx, y = False, np.inf, np.zeros(n), x
            rho = (chi - new) / ((float(dx @ (lam * dx + b)) + 1e-3) if ok else 1.0)
            if rho > 0 and np.isfinite(new) and ok:
                alpha = min(1.0 - (2 * rho - 1) ** 3, 2.0 / 3.0)
                lam *= max(1.0 / 3.0, alpha)
                ni, x, chi = 2.0, y, new
            else:
                lam *= ni
                ni *= 2
                if not np.isfinite(lam):
                    break
            qmax += 1
            if not (rho < 0 and qmax < 10):
                break
        if qmax == 10 or rho == 0 or not np.isfinite(lam):
            break
    return x, {"iterations": it + 1, "chi2": pose_graph_chi2(x, kinds, edges), "chi2_initial": chi0}


def box_alignment_relative_sample_np(pred_corners_list, noisy_lidar_pose, uncertainty_list=None, max_iterations=1000, **flags):
    """box_align_v2.py:101-396 -> refined [N, 3] (x, y, yaw in degrees)."""
    noisy_lidar_pose = np.asarray(noisy_lidar_pose)
    g = build_pose_graph(pred_corners_list, noisy_lidar_pose, uncertainty_list, **flags)
    if g is None:
        return noisy_lidar_pose[:, [0, 1, 4]]
    x, _ = pose_graph_lm(g["vertices"], g["kinds"], g["edges"], max_iterations)
    refined = x[: g["n_agents"]].copy()
    refined[:, 2] = np.rad2deg(refined[:, 2])
    return refined


def post_process_stage1(out: Dict[str, torch.Tensor], anchors: torch.Tensor, pp_cfg: dict):
    """UncertaintyVoxelPostprocessor.post_process_stage1 (uncertainty_voxel_postprocessor.py:30-112): per agent -- score
    threshold, decode + direction fix, corners in the agent's own frame (no projection, no size / z / range filters), rotated
    NMS -> lists of corners [K_i, 8, 3], boxes [K_i, 7], uncertainty [K_i, dim]; (None, None, None) without any candidate."""
    thr, order, da = pp_cfg["target_args"]["score_threshold"], pp_cfg["order"], pp_cfg.get("dir_args", {})
    cls, reg, unc, dirp = out["cls_preds"], out["reg_preds"], out["unc_preds"], out.get("dir_preds")
    A = cls.shape[1]
    udim = unc.shape[1] // A
    corners_l, boxes_l, unc_l, total = [], [], [], 0
    for i in range(cls.shape[0]):
        idx, boxes3d, scores, corners = decode_candidates(cls[i: i + 1], reg[i: i + 1], None if dirp is None else dirp[i: i + 1], anchors, thr,
                                                          order, da.get("dir_offset", 0.7853), da.get("num_bins", 2), None)
        total += len(idx)
        u = unc[i].permute(1, 2, 0).reshape(-1, udim)[idx]
        keep = torch.from_numpy(nms_rotated(corners, scores, pp_cfg["nms_thresh"]).astype(np.int64)) if len(idx) else torch.zeros(0, dtype=torch.long)
        corners_l.append(corners[keep]); boxes_l.append(boxes3d[keep]); unc_l.append(u[keep])
    if total == 0:
        return None, None, None
    return corners_l, boxes_l, unc_l
