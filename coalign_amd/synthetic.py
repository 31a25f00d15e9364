"""Synthetic OPV2V-shaped frames (SURVEY §8d "Synthetic inputs").

There is no dataset in the build or bench environment, so every test and ``bench.py`` feed the hot
path with frames generated here.  The output is the batch dict the reference's collate functions
produce for ``model(batch_data['ego'])`` (intermediate_fusion_dataset.py:441-575 keys only):

    processed_lidar: voxel_features [sum M, 32, 4] f32, voxel_coords [sum M, 4] i32 (agent, z, y, x),
                     voxel_num_points [sum M] i32
    record_len [B] int64, pairwise_t_matrix [B, L, L, 4, 4] float64
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from .pose import generate_noise, get_pairwise_transformation


def make_pillars(rng: np.random.RandomState, agent: int, n_pillars: int, nx: int, ny: int,
                 voxel_size: Sequence[float], pc_range: Sequence[float], max_points: int = 32,
                 num_points_mode: str = "geometric"):
    """One agent's pillars: ``n_pillars`` distinct cells, 1..max_points points each (LiDAR-like skew:
    ``min(P, 1 + Geometric(0.12))`` or uniform), points jittered around the cell centre, zero padded."""
    n_pillars = min(n_pillars, nx * ny)
    cells = rng.choice(nx * ny, size=n_pillars, replace=False)
    cy, cx = np.divmod(cells, nx)
    if num_points_mode == "geometric":
        npts = np.minimum(max_points, rng.geometric(0.12, size=n_pillars)).astype(np.int32)
    else:
        npts = rng.randint(1, max_points + 1, size=n_pillars).astype(np.int32)
    vx, vy = voxel_size[0], voxel_size[1]
    ctr_x = (cx + 0.5) * vx + pc_range[0]
    ctr_y = (cy + 0.5) * vy + pc_range[1]
    pts = np.zeros((n_pillars, max_points, 4), dtype=np.float32)
    pts[:, :, 0] = ctr_x[:, None] + rng.uniform(-0.5 * vx, 0.5 * vx, size=(n_pillars, max_points))
    pts[:, :, 1] = ctr_y[:, None] + rng.uniform(-0.5 * vy, 0.5 * vy, size=(n_pillars, max_points))
    pts[:, :, 2] = rng.uniform(pc_range[2], pc_range[5], size=(n_pillars, max_points))
    pts[:, :, 3] = rng.uniform(0.0, 1.0, size=(n_pillars, max_points))
    pts *= (np.arange(max_points)[None, :] < npts[:, None])[:, :, None]
    coords = np.stack([np.full(n_pillars, agent), np.zeros(n_pillars, dtype=np.int64), cy, cx], axis=1).astype(np.int32)
    return pts, coords, npts


def make_poses(rng: np.random.RandomState, n_agents: int, noise: Optional[Sequence[float]] = None,
               spread_xy=(20.0, 10.0), spread_yaw=30.0, infra_agent: bool = False) -> List[np.ndarray]:
    """Ego at the origin; others within +-spread (metres / degrees).  ``infra_agent`` places agent 1 like a
    DAIR-V2X road-side unit (~30 m ahead, facing back).  Optional Gaussian noise (pos m, rot deg) is
    added to every agent's pose with the reference's draw order."""
    poses = [np.zeros(6)]
    for a in range(1, n_agents):
        if infra_agent and a == 1:
            poses.append(np.array([30.0, 5.0, 0.0, 0.0, 170.0, 0.0]))
        else:
            poses.append(np.array([rng.uniform(-spread_xy[0], spread_xy[0]), rng.uniform(-spread_xy[1], spread_xy[1]),
                                   0.0, 0.0, rng.uniform(-spread_yaw, spread_yaw), 0.0]))
    if noise is not None:
        poses = [p + generate_noise(noise[0], noise[1], rng=rng) for p in poses]
    return poses


def make_frame(hypes: dict, n_agents: Sequence[int] | int, pillars_per_agent: int = 8000, seed: int = 303,
               num_points_mode: str = "geometric", noise: Optional[Sequence[float]] = None,
               infra_agent: bool = False, spread_xy=(20.0, 10.0), spread_yaw=30.0) -> Dict:
    """Batch dict for ``model.forward``.  ``n_agents`` may be a list (one entry per frame in the batch)."""
    rng = np.random.RandomState(seed)
    record = [n_agents] if isinstance(n_agents, int) else list(n_agents)
    margs = hypes["model"]["args"]
    nx, ny, _ = [int(v) for v in margs["point_pillar_scatter"]["grid_size"]]
    L = int(hypes.get("train_params", {}).get("max_cav", 5))
    L = max(L, max(record))
    feats, coords, npts, pair = [], [], [], []
    agent = 0
    for n in record:
        for _ in range(n):
            p, c, k = make_pillars(rng, agent, pillars_per_agent, nx, ny, margs["voxel_size"], margs["lidar_range"],
                                   num_points_mode=num_points_mode)
            feats.append(p); coords.append(c); npts.append(k)
            agent += 1
        poses = make_poses(rng, n, noise=noise, infra_agent=infra_agent, spread_xy=spread_xy, spread_yaw=spread_yaw)
        pair.append(get_pairwise_transformation(poses, L))
    return {
        "processed_lidar": {
            "voxel_features": torch.from_numpy(np.concatenate(feats)),
            "voxel_coords": torch.from_numpy(np.concatenate(coords)),
            "voxel_num_points": torch.from_numpy(np.concatenate(npts)),
        },
        "record_len": torch.tensor(record, dtype=torch.int64),
        "pairwise_t_matrix": torch.from_numpy(np.stack(pair)),
    }


def fill_parameters_(module: torch.nn.Module, seed: int = 0, cls_bias: float = -2.0) -> None:
    """Deterministic, name-keyed test weights (independent of module construction order, so two
    differently written implementations with the same state_dict names get identical values).
    BN running stats are randomised (mean ~ N(0, .1), var ~ U(.5, 1.5)) so BN folding is exercised."""
    import zlib
    sd = module.state_dict()
    with torch.no_grad():
        for name in sorted(sd.keys()):
            t = sd[name]
            if not t.is_floating_point():
                continue
            g = torch.Generator().manual_seed((zlib.crc32(name.encode()) + seed) & 0x7FFFFFFF)
            shape = tuple(t.shape)
            if name.endswith("running_var"):
                v = torch.rand(shape, generator=g) + 0.5
            elif name.endswith("running_mean"):
                v = torch.randn(shape, generator=g) * 0.1
            elif name.endswith("bias"):
                v = torch.randn(shape, generator=g) * 0.1
                if name == "cls_head.bias":
                    v = v + cls_bias
            elif t.dim() == 1:                       # BN / norm scale
                v = torch.rand(shape, generator=g) + 0.5
            else:                                    # conv / linear / deconv weight
                fan_in = t[0].numel() if t.dim() > 1 else t.numel()
                if "deblocks" in name and t.dim() == 4:   # ConvTranspose2d weight is [Cin, Cout, k, k]
                    fan_in = t.shape[0]
                v = torch.randn(shape, generator=g) * (1.5 / fan_in) ** 0.5
            t.copy_(v.to(t.dtype))


def fill_parameters_trained_like_(model: torch.nn.Module, seed: int = 0, cls_bias: float = -2.0, log2_lo: float = -8.0, log2_hi: float = 4.0,
                                  dead_fraction: float = 0.3, tiny_gamma_fraction: float = 0.05) -> Dict[str, float]:
    """A second deterministic parameter set that has the statistics of a TRAINED checkpoint where ``fill_parameters_`` has those of an initialisation
    (what really gets loaded: opencood/tools/train_utils.py:29-74):

    * convolution / deconvolution weights are Student-t (nu = 3, heavy tailed) instead of Gaussian, same variance;
    * every BatchNorm-folded convolution has its own scale: the map behind BN layer ``l`` lives at 2^c_l, c_l a clipped random walk with steps drawn
      uniformly from [``log2_lo``, ``log2_hi``] (folded weight scale 2^(c_l - c_(l-1)): 2^-8 ... 2^4 around the initialisation's), the running statistics
      track the data as they do after training (running_mean ~ 2^c_in, running_var ~ 4^c_in);
    * ``tiny_gamma_fraction`` of every layer's BN weights are 1e-3 of their neighbours (channels training has switched off by shrinking gamma);
    * ``dead_fraction`` of the channels carry a BN bias three scales below zero: dead behind the ReLU;
    * the shrink header's biased convolutions are rescaled the same way.

    The heads are left to ``calibrate_heads_``.  Returns {BN / conv name: log2 scale of the map behind it} (what the tests print)."""
    from .backbone import BasicBlock, DoubleConv
    import zlib
    fill_parameters_(model, seed=seed, cls_bias=cls_bias)

    def gen(name):
        return torch.Generator().manual_seed((zlib.crc32(("trained:" + name).encode()) + seed) & 0x7FFFFFFF)

    def student_t_(w: torch.Tensor, name: str):
        g = gen(name)
        z = torch.randn(w.shape, generator=g)
        chi = (torch.randn((3,) + tuple(w.shape), generator=g) ** 2).sum(0) / 3.0
        t = z / chi.sqrt() / 3.0 ** 0.5                                   # Student-t(3), unit variance
        w.copy_((t * w.float().std()).to(w.dtype))

    names = {m: n for n, m in model.named_modules()}
    scales: Dict[str, float] = {}

    def step(c_in: float, name: str) -> float:
        g = gen("step:" + name)
        t = log2_lo + (log2_hi - log2_lo) * float(torch.rand((), generator=g))
        return min(max(c_in + t, log2_lo), log2_hi)

    def set_bn(bn, conv_w, c_in: float, c_out: float):
        name = names[bn]
        g = gen(name)
        ch = bn.weight.shape[0]
        student_t_(conv_w, name + ".conv")
        bn.running_mean.copy_(torch.randn(ch, generator=g) * 0.1 * 2.0 ** c_in)
        bn.running_var.copy_((torch.rand(ch, generator=g) + 0.5) * 4.0 ** c_in)
        gamma = (torch.rand(ch, generator=g) + 0.5) * 2.0 ** c_out
        tiny = torch.rand(ch, generator=g) < tiny_gamma_fraction
        gamma = torch.where(tiny, gamma * 1e-3, gamma)
        beta = torch.randn(ch, generator=g) * 0.1 * 2.0 ** c_out
        dead = torch.rand(ch, generator=g) < dead_fraction
        beta = torch.where(dead, -3.0 * gamma.abs() - 0.1 * 2.0 ** c_out, beta)
        bn.weight.copy_(gamma)
        bn.bias.copy_(beta)
        scales[name] = c_out

    with torch.no_grad():
        c = 0.0
        stage_c: List[float] = []
        last_stage = None
        for mod in model.modules():
            if isinstance(mod, BasicBlock):
                stage = names[mod].rsplit(".", 1)[0]
                if last_stage is not None and stage != last_stage:
                    stage_c.append(c)
                last_stage = stage
                c1 = step(c, names[mod.bn1])
                set_bn(mod.bn1, mod.conv1.weight, c, c1)
                c2 = step(c1, names[mod.bn2])
                set_bn(mod.bn2, mod.conv2.weight, c1, c2)
                skip = c
                if mod.downsample is not None:
                    skip = step(c, names[mod.downsample[1]])
                    set_bn(mod.downsample[1], mod.downsample[0].weight, c, skip)
                c = max(c2, skip)
        if last_stage is not None:
            stage_c.append(c)
        backbone = getattr(model, "backbone", None)
        if backbone is not None and hasattr(backbone, "blocks"):            # the plain conv-BN-ReLU stacks of BaseBEVBackbone
            c, stage_c = 0.0, []
            for blk in backbone.blocks:
                mods = list(blk)
                for k, m in enumerate(mods):
                    if isinstance(m, torch.nn.BatchNorm2d):
                        c2 = step(c, names[m])
                        set_bn(m, mods[k - 1].weight, c, c2)
                        c = c2
                stage_c.append(c)
        c_cat = c
        if backbone is not None and len(getattr(backbone, "deblocks", [])):
            outs = []
            for i, blk in enumerate(backbone.deblocks):
                c_in = stage_c[i] if i < len(stage_c) else c
                c_out = step(c_in, names[blk[1]])
                set_bn(blk[1], blk[0].weight, c_in, c_out)
                outs.append(c_out)
            c_cat = max(outs)
        for mod in model.modules():
            if isinstance(mod, DoubleConv):
                for conv in (mod.double_conv[0], mod.double_conv[2]):
                    name = names[conv]
                    c_out = step(c_cat, name)
                    student_t_(conv.weight, name)
                    conv.weight.mul_(2.0 ** (c_out - c_cat))
                    g = gen(name + ".bias")
                    b = torch.randn(conv.bias.shape, generator=g) * 0.1 * 2.0 ** c_out
                    dead = torch.rand(conv.bias.shape, generator=g) < dead_fraction
                    conv.bias.copy_(torch.where(dead, -3.0 * 2.0 ** c_out * torch.ones_like(b), b))
                    scales[name] = c_out
                    c_cat = c_out
    return scales


def calibrate_heads_(model: torch.nn.Module, batch: Dict, score_threshold: float, target: int = 600) -> None:
    """Make a random-init detector's heads behave like a trained one's on ``batch`` ("head biases shifted so K ~ 300-1000
    candidates"): box deltas of std 0.1 (decoded boxes stay car sized and pass the size / z sanity filters, neighbouring anchors
    overlap -> NMS has work), classification logits of unit variance (scores spread over (0, 1): no pile-up of candidates at a
    saturated score of exactly 1.0, whose order would be decided by tie-breaking alone) and a bias that puts ~``target`` anchors
    above the score threshold.  Deterministic given (weights, batch)."""
    import math
    with torch.no_grad():
        out = model(batch)
        model.reg_head.weight *= 0.1 / float(out["reg_preds"].std())
        model.reg_head.bias.zero_()
        model.cls_head.weight *= 1.0 / float(out["cls_preds"].std())
        model.cls_head.bias.zero_()
        logits = model(batch)["cls_preds"].flatten()
        k = min(target, logits.numel() - 1)
        v = torch.topk(logits, k + 1).values[-1]
        model.cls_head.bias += (math.log(score_threshold / (1 - score_threshold)) - float(v))


def make_point_cloud(seed: int, beams: int = 64, azimuth_steps: int = 1800, sensor_height: float = 1.9, max_range: float = 120.0,
                     n_boxes: int = 30, dropout: float = 0.3) -> np.ndarray:
    """A spinning-lidar sweep [N, 4] float32 (x, y, z, intensity) in the sensor frame, N ~ 60-70 k like one OPV2V cav
    (64 beams x 0.2 deg): flat ground, a street canyon of two walls, and axis-aligned vehicle-sized boxes -- so that, like
    real data, pillars near the sensor and on vertical surfaces hold far more than 32 points while most hold a handful.
    The point order is the sensor's firing order (azimuth major), i.e. spatially coherent, not shuffled."""
    rs = np.random.RandomState(seed)
    pitch = np.radians(np.linspace(-24.8, 2.0, beams))
    az = np.radians(np.arange(azimuth_steps) * (360.0 / azimuth_steps))
    dx, dy, dz = (np.cos(pitch)[None] * np.cos(az)[:, None]), (np.cos(pitch)[None] * np.sin(az)[:, None]), np.broadcast_to(np.sin(pitch)[None], (azimuth_steps, beams))
    t = np.full(dx.shape, np.inf)
    with np.errstate(divide="ignore", invalid="ignore"):
        tg = np.where(dz < 0, -sensor_height / dz, np.inf)                       # ground
        t = np.minimum(t, tg)
        for wall_y in (rs.uniform(12, 30), -rs.uniform(12, 30)):                 # building fronts, 6 m high, with gaps
            tw = wall_y / dy
            zw, xw = tw * dz, tw * dx
            solid = np.sin(xw / rs.uniform(6, 14) + rs.uniform(0, 6)) > -0.2
            t = np.minimum(t, np.where((tw > 0) & (zw < 6.0 - sensor_height) & solid, tw, np.inf))
        for _ in range(n_boxes):                                                 # vehicles: slab test against an AABB
            cx, cy = rs.uniform(-90, 90), rs.uniform(-8, 8)
            lo = np.array([cx - 2.3, cy - 1.0, -sensor_height])
            hi = np.array([cx + 2.3, cy + 1.0, -sensor_height + 1.6])
            if lo[0] < 3 and hi[0] > -3 and lo[1] < 1.5 and hi[1] > -1.5:
                continue                                                          # not on top of the sensor
            t1, t2 = lo[0] / dx, hi[0] / dx
            tn, tf = np.minimum(t1, t2), np.maximum(t1, t2)
            t1, t2 = lo[1] / dy, hi[1] / dy
            tn, tf = np.maximum(tn, np.minimum(t1, t2)), np.minimum(tf, np.maximum(t1, t2))
            t1, t2 = lo[2] / dz, hi[2] / dz
            tn, tf = np.maximum(tn, np.minimum(t1, t2)), np.minimum(tf, np.maximum(t1, t2))
            t = np.minimum(t, np.where((tn <= tf) & (tn > 0), tn, np.inf))
    hit = np.isfinite(t) & (t < max_range) & (rs.uniform(0, 1, t.shape) > dropout)
    t = np.where(hit, t, 0.0) + rs.normal(0, 0.02, t.shape)
    pts = np.stack([t * dx, t * dy, t * dz, rs.uniform(0, 1, t.shape)], axis=-1)[hit]
    return np.ascontiguousarray(pts, dtype=np.float32)


def make_points_frame(hypes: dict, n_agents: int, seed: int = 303, noise: Optional[Sequence[float]] = None, **cloud_kw) -> Dict:
    """A frame as the reference's loader holds it BEFORE the voxeliser: one raw sweep per cav (``make_point_cloud``, each in its own sensor
    frame, ~65 k points) + poses -> {"clouds": [ndarray [n_i, 4]], "record_len": [n_agents], "pairwise_t_matrix": float64 [1, L, L, 4, 4]} --
    the input of ``FramePipeline.submit_points``."""
    rng = np.random.RandomState(seed)
    L = max(int(hypes.get("train_params", {}).get("max_cav", 5)), n_agents)
    clouds = [make_point_cloud(seed * 16 + i, **cloud_kw) for i in range(n_agents)]
    poses = make_poses(rng, n_agents, noise=noise)
    return {"clouds": clouds, "record_len": [n_agents], "pairwise_t_matrix": torch.from_numpy(get_pairwise_transformation(poses, L)[None])}
