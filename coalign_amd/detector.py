"""Detector classes of the hot path -- the plugin boundary.

``PointPillarBaselineMultiscale`` / ``CoAlign`` / ``PointPillar`` keep the reference's class names, constructor
(``args`` = ``hypes['model']['args']``), ``forward(data_dict) -> {'cls_preds', 'reg_preds'[, 'dir_preds']}`` and
``state_dict`` key names (opencood/models/point_pillar_baseline_multiscale.py:17-135,
opencood/models/point_pillar_coalign.py:9-10, opencood/models/point_pillar.py:16-80), so reference yamls and
checkpoints drive them unchanged.  Pillar encode + scatter and warp + fusion run in the gfx950 kernels; the
dense convolutions run on MIOpen through PyTorch-ROCm.
"""
from __future__ import annotations

import torch
import torch.nn as nn

import torch.nn.functional as F

from . import ops
from .backbone import BaseBEVBackbone, DownsampleConv, NaiveCompressor, ResNetBEVBackbone, _cache_of, _fast_ok
from .encoder import PillarVFE, PointPillarScatter, host_ints
from .fusion import AttFusion, MaxFusion, fuse_multiscale
from .pose import normalize_pairwise_tfm


import os as _os

# Round 4: PillarVFE + PointPillarScatter as one launch with a sparse canvas (csrc/pillar_sparse.hip) feeding the first ResNet block directly.
# "0": the dense persistent canvas of rounds 2-3 (measurement aid; module attribute, read at every call).
SPARSE_CANVAS = _os.environ.get("COALIGN_SPARSE_CANVAS", "1") != "0"


def _run_heads(model: nn.Module, x: torch.Tensor) -> dict:
    """cls / reg / dir 1x1 heads.  Inference fast path: one convolution with the concatenated head weights (+ one fused
    bias pass) instead of three convolutions and three bias kernels; the outputs are channel slices of one tensor."""
    heads = [("cls_preds", model.cls_head), ("reg_preds", model.reg_head)]
    if getattr(model, "unc_head", None) is not None:
        heads.append(("unc_preds", model.unc_head))
    if model.use_dir:
        heads.append(("dir_preds", model.dir_head))
    if isinstance(x, ops.SplitMap) and not heads_take_split_map(model, x):
        x = x.dense()
    if not isinstance(x, ops.SplitMap) and not _fast_ok(model, x):
        return {k: h(x) for k, h in heads}

    def build():
        return (torch.cat([h.weight for _, h in heads]).contiguous(), torch.cat([h.bias for _, h in heads]).contiguous())
    w, b = _cache_of(model.cls_head).get([t for _, h in heads for t in (h.weight, h.bias)], build)
    from . import backbone as _bb
    if isinstance(x, ops.SplitMap):
        # round 6: the shrink header's last convolution handed its map over as a SplitMap: the merged heads read it with one 16-byte load per lane and operand
        img = _bb._pw_cache_of(model.reg_head).get([t for _, h in heads for t in (h.weight, h.bias)], lambda: ops.pack_heads_sp_weight(w))      # (a cache slot of its own: reg_head's fold cache holds the pointwise image)
        y = ops.heads_sp(x, img, b, w.shape[0])
    elif _bb.CONV_EMU_TERMS in (3, 16) and _bb.POINTWISE_EMU and w.shape[1] % 16 == 0 and w.shape[1] <= 512:
        # round 4: the merged 1x1 heads on the hand-written pointwise kernel (split-bf16 matrix cores; GEMM rows padded to 32), reading the shrink
        # header's map in whatever layout it has and writing the NCHW maps the decode kernel reads -- no rocBLAS / bias pass in the frame
        pk = _cache_of(model.reg_head).get([t for _, h in heads for t in (h.weight, h.bias)], lambda: _bb.PointwisePack(w, False))
        y = ops.pointwise_conv(x, pk.get(), b, w.shape[0], relu=False)
    else:
        y = ops.bias_act_(F.conv2d(x, w, None), b, None, False)
    out, c0 = {}, 0
    for k, h in heads:
        out[k] = y[:, c0:c0 + h.out_channels]
        c0 += h.out_channels
    return out


def heads_take_split_map(model: nn.Module, x=None) -> bool:
    """The merged heads can read the shrink header's map as a SplitMap (``ops.heads_sp``): <= 32 head channels in all, Cin % 16 == 0, eval mode, the switch on."""
    from . import backbone as _bb
    if not _bb.HEADS_SPLIT_IN or model.training or not _bb.split_maps_active():
        return False
    heads = [model.cls_head, model.reg_head] + ([model.unc_head] if getattr(model, "unc_head", None) is not None else []) + ([model.dir_head] if model.use_dir else [])
    rows = sum(h.out_channels for h in heads)
    return rows <= 32 and model.cls_head.in_channels % 16 == 0 and all(tuple(h.kernel_size) == (1, 1) for h in heads) and model.cls_head.weight.is_cuda


def _single_agent_batch(data_dict: dict) -> dict:
    """``batch_dict`` of the single-agent models (point_pillar.py:52-60) plus the optional keys of this build's callers: ``record_len`` when the batch carries one
    (saves PillarVFE its host read of the largest agent index), the device voxeliser's streaming keys and FramePipeline's frame record -- PillarVFE raises
    ``FrameRecordUnsupported`` for a record on a route that reads the arrays directly, so a record can never be dropped silently."""
    pl = data_dict["processed_lidar"]
    batch_dict = {"voxel_features": pl["voxel_features"], "voxel_coords": pl["voxel_coords"], "voxel_num_points": pl["voxel_num_points"]}
    if data_dict.get("record_len") is not None:
        batch_dict["record_len"] = host_ints(data_dict["record_len"])
    for k in ("voxel_count_dev", "voxel_cells_unique", "want_pillar_features", "pillar_frame"):
        if k in pl:
            batch_dict[k] = pl[k]
    return batch_dict


def x_is_cuda(maps) -> bool:
    return len(maps) > 0 and all(getattr(m, "is_cuda", False) for m in maps)


class PointPillarBaselineMultiscale(nn.Module):
    def __init__(self, args: dict):
        super().__init__()
        self.pillar_vfe = PillarVFE(args["pillar_vfe"], num_point_features=4, voxel_size=args["voxel_size"],
                                    point_cloud_range=args["lidar_range"])
        self.scatter = PointPillarScatter(args["point_pillar_scatter"])
        bb = args["base_bev_backbone"]
        self.backbone = ResNetBEVBackbone(bb, 64) if bb.get("resnet", True) else BaseBEVBackbone(bb, 64)
        self.voxel_size = args["voxel_size"]
        self.fusion_net = nn.ModuleList()
        for i in range(len(bb["layer_nums"])):
            if args["fusion_method"] == "max":
                self.fusion_net.append(MaxFusion())
            elif args["fusion_method"] == "att":
                self.fusion_net.append(AttFusion(args["att"]["feat_dim"][i]))
            else:
                raise NotImplementedError(f"fusion_method '{args['fusion_method']}' is outside the CoAlign hot path (att | max)")
        self.out_channel = sum(bb["num_upsample_filter"])
        self.shrink_flag = "shrink_header" in args
        if self.shrink_flag:
            self.shrink_conv = DownsampleConv(args["shrink_header"])
            self.out_channel = args["shrink_header"]["dim"][-1]
        self.compression = "compression" in args
        if self.compression:
            self.naive_compressor = NaiveCompressor(64, args["compression"])
        self.cls_head = nn.Conv2d(self.out_channel, args["anchor_number"], kernel_size=1)
        self.reg_head = nn.Conv2d(self.out_channel, 7 * args["anchor_number"], kernel_size=1)
        self.use_dir = "dir_args" in args
        if self.use_dir:
            self.dir_head = nn.Conv2d(self.out_channel, args["dir_args"]["num_bins"] * args["anchor_number"], kernel_size=1)
        if args.get("backbone_fix"):
            self.backbone_fix()

    def backbone_fix(self):
        frozen = [self.pillar_vfe, self.scatter, self.backbone, self.cls_head, self.reg_head]
        if self.compression:
            frozen.append(self.naive_compressor)
        if self.shrink_flag:
            frozen.append(self.shrink_conv)
        for m in frozen:
            for p in m.parameters():
                p.requires_grad = False

    # -- stages, exposed separately so the sharded runner can place them on different ranks ----------------
    accepts_normalized_affine = True      # encode() takes data_dict['normalized_affine_matrix'] in place of normalising pairwise_t_matrix itself (FramePipeline)
    accepts_pillar_frame = True           # encode() hands processed_lidar['pillar_frame'] (ops.PillarFrameRecord) to PillarVFE: FramePipeline may read frames in place

    def encode(self, data_dict: dict):
        """Per-agent part: pillars -> canvas -> multiscale features.  Returns (feature list, normalised affine)."""
        pl = data_dict["processed_lidar"]
        record_len = host_ints(data_dict["record_len"])
        batch_dict = {"voxel_features": pl["voxel_features"], "voxel_coords": pl["voxel_coords"],
                      "voxel_num_points": pl["voxel_num_points"], "record_len": record_len}
        for k in ("voxel_count_dev", "voxel_cells_unique", "want_pillar_features", "pillar_frame"):      # the device voxeliser's streaming form (PillarVFE.forward); FramePipeline's frame record
            if k in pl:
                batch_dict[k] = pl[k]
        # round 4: the encoder hands a SparseCanvas (one launch, no dense canvas) to a backbone whose first block reads it
        resnet = getattr(self.backbone, "resnet", None)
        first = resnet.layer0[0] if resnet is not None and hasattr(resnet, "layer0") and hasattr(resnet.layer0[0], "takes_sparse_canvas") else None
        keep_sparse = self.pillar_vfe.sparse_canvas
        self.pillar_vfe.sparse_canvas = bool(SPARSE_CANVAS and not self.compression and not self.training and first is not None and first.takes_sparse_canvas())
        try:
            batch_dict = self.scatter(self.pillar_vfe(batch_dict))
        finally:
            self.pillar_vfe.sparse_canvas = keep_sparse
        spatial_features = batch_dict["spatial_features"]
        H0, W0 = spatial_features.shape[2:]
        affine = data_dict.get("normalized_affine_matrix")      # FramePipeline: normalised on the host from the dataset's host copy of the matrix (same float64 steps)
        if affine is None:
            affine = normalize_pairwise_tfm(data_dict["pairwise_t_matrix"], H0, W0, self.voxel_size[0])
        if self.compression:
            spatial_features = self.naive_compressor(spatial_features)
        return self.backbone.get_multiscale_feature(spatial_features), affine

    def _fuse_scales(self, feature_list, record_len, affine, rows=None):
        """The per-scale fusion launches are independent: on the GPU the coarser scales (few workgroups, latency bound)
        run on side streams next to the finest one instead of queueing behind it."""
        n = len(feature_list)
        x0 = feature_list[0]
        if x0.is_cuda and not self.training:
            kinds = {type(f) for f in self.fusion_net}
            # (an AttFusion whose configured feat_dim differs from its map's channels -- no shipped yaml -- takes the per-scale route below,
            #  where the module rescales its input so that the scores are divided by sqrt(feat_dim) like the reference's)
            plain_att = kinds == {AttFusion} and all(f.feature_dims == x.shape[1] for f, x in zip(self.fusion_net, feature_list))
            if plain_att or kinds == {MaxFusion}:
                fused = fuse_multiscale(feature_list, record_len, affine, ops.FUSE_ATT if kinds == {AttFusion} else ops.FUSE_MAX, rows)
                if fused is not None:
                    return fused
        if n == 1 or not x0.is_cuda or self.training:
            return [f(x, record_len, affine, rows=rows) for f, x in zip(self.fusion_net, feature_list)]
        main = torch.cuda.current_stream(x0.device)
        side = self.__dict__.get("_fusion_streams")
        if side is None or len(side) != n - 1 or side[0].device != x0.device:
            side = self.__dict__["_fusion_streams"] = [torch.cuda.Stream(device=x0.device) for _ in range(n - 1)]
        fused = [None] * n
        for i in range(1, n):
            s = side[i - 1]
            s.wait_stream(main)
            with torch.cuda.stream(s):
                fused[i] = self.fusion_net[i](feature_list[i], record_len, affine, rows=rows)
            feature_list[i].record_stream(s)
            affine.record_stream(s)
        fused[0] = self.fusion_net[0](feature_list[0], record_len, affine, rows=rows)
        for i in range(1, n):
            main.wait_stream(side[i - 1])
            fused[i].record_stream(main)
        return fused

    def fuse_and_head(self, feature_list, record_len, affine, rows=None) -> dict:
        """Ego part: per-scale warp + fusion, deblocks, shrink header, heads.  ``rows``: see ``AttFusion.forward``."""
        fused = self._fuse_scales(feature_list, record_len, affine, rows)
        # round 5: a shrink header on the SplitMap route gets the concatenated map from the up-sampling heads already split
        want_split = bool(self.shrink_flag and x_is_cuda(fused) and self.shrink_conv.takes_split_maps())
        x = self.backbone.decode_multiscale_feature(fused, out_split=True) if want_split else self.backbone.decode_multiscale_feature(fused)
        if self.shrink_flag:
            x = self.shrink_conv(x, out_split=isinstance(x, ops.SplitMap) and heads_take_split_map(self))
        elif isinstance(x, ops.SplitMap):
            x = x.dense()
        return _run_heads(self, x)

    def forward(self, data_dict: dict) -> dict:
        record_len = host_ints(data_dict["record_len"])
        feats, affine = self.encode(dict(data_dict, record_len=record_len))
        return self.fuse_and_head(feats, record_len, affine)


class CoAlign(PointPillarBaselineMultiscale):
    pass


class PointPillar(nn.Module):
    """Single-agent PointPillar (late-fusion config)."""

    def __init__(self, args: dict):
        super().__init__()
        self.pillar_vfe = PillarVFE(args["pillar_vfe"], num_point_features=4, voxel_size=args["voxel_size"],
                                    point_cloud_range=args["lidar_range"])
        self.scatter = PointPillarScatter(args["point_pillar_scatter"])
        bb = args["base_bev_backbone"]
        self.backbone = ResNetBEVBackbone(bb, 64) if bb.get("resnet", False) else BaseBEVBackbone(bb, 64)
        self.out_channel = sum(bb["num_upsample_filter"])
        self.shrink_flag = "shrink_header" in args
        if self.shrink_flag:
            self.shrink_conv = DownsampleConv(args["shrink_header"])
            self.out_channel = args["shrink_header"]["dim"][-1]
        self.cls_head = nn.Conv2d(self.out_channel, args["anchor_number"], kernel_size=1)
        self.reg_head = nn.Conv2d(self.out_channel, 7 * args["anchor_number"], kernel_size=1)
        self.use_dir = "dir_args" in args
        if self.use_dir:
            self.dir_head = nn.Conv2d(self.out_channel, args["dir_args"]["num_bins"] * args["anchor_number"], kernel_size=1)

    def forward(self, data_dict: dict) -> dict:
        batch_dict = self.backbone(self.scatter(self.pillar_vfe(_single_agent_batch(data_dict))))
        x = batch_dict["spatial_features_2d"]
        if self.shrink_flag:
            x = self.shrink_conv(x)
        return _run_heads(self, x)


class PointPillarUncertainty(nn.Module):
    """Stage-1 single-agent detector of CoAlign's box alignment (opencood/models/point_pillar_uncertainty.py:15-84): PointPillar
    with a fourth 1x1 head ``unc_head`` predicting ``uncertainty_dim`` log-variances (x, y[, yaw]) per anchor."""

    def __init__(self, args: dict):
        super().__init__()
        self.pillar_vfe = PillarVFE(args["pillar_vfe"], num_point_features=4, voxel_size=args["voxel_size"],
                                    point_cloud_range=args["lidar_range"])
        self.scatter = PointPillarScatter(args["point_pillar_scatter"])
        self.backbone = BaseBEVBackbone(args["base_bev_backbone"], 64)
        self.uncertainty_dim = args["uncertainty_dim"]
        width = 128 * 3                                          # hard-coded in the reference (:26-37)
        self.cls_head = nn.Conv2d(width, args["anchor_number"], kernel_size=1)
        self.reg_head = nn.Conv2d(width, 7 * args["anchor_number"], kernel_size=1)
        self.unc_head = nn.Conv2d(width, self.uncertainty_dim * args["anchor_number"], kernel_size=1)
        self.use_dir = "dir_args" in args
        if self.use_dir:
            self.dir_head = nn.Conv2d(width, args["dir_args"]["num_bins"] * args["anchor_number"], kernel_size=1)

    def forward(self, data_dict: dict) -> dict:
        batch_dict = self.backbone(self.scatter(self.pillar_vfe(_single_agent_batch(data_dict))))
        return _run_heads(self, batch_dict["spatial_features_2d"])


MODEL_REGISTRY = {
    "point_pillar_baseline_multiscale": PointPillarBaselineMultiscale,
    "point_pillar_coalign": CoAlign,
    "point_pillar": PointPillar,
    "point_pillar_uncertainty": PointPillarUncertainty,
}


def build_model(hypes: dict) -> nn.Module:
    """``train_utils.create_model`` for the hot-path model families (opencood/tools/train_utils.py:113-146):
    ``hypes['model']['core_method']`` names the model, ``hypes['model']['args']`` is its constructor argument."""
    name = hypes["model"]["core_method"]
    if name not in MODEL_REGISTRY:
        raise KeyError(f"model '{name}' is outside the CoAlign hot path (available: {sorted(MODEL_REGISTRY)})")
    return MODEL_REGISTRY[name](hypes["model"]["args"])


def load_saved_model(saved_path: str, model: nn.Module):
    """``train_utils.load_saved_model`` (opencood/tools/train_utils.py:29-74): find the checkpoint of a training folder and load it into ``model``.

    * ``net_epoch_bestval_at<E>.pth`` present (exactly one is allowed) -> that file, returns ``(E, model)``;
    * else the highest ``net_epoch<E>.pth`` among ``*epoch*.pth`` -> that file, returns ``(E, model)``;
    * else nothing is loaded and ``(0, model)`` comes back.

    Like the reference: tensors are loaded onto the CPU (``map_location='cpu'``) and copied into whatever device the model lives on, ``strict=False`` (keys the
    model does not have are ignored, keys the file lacks keep their initial values), a missing folder is an ``AssertionError``.  Unlike the reference the epoch is
    parsed with a regular expression, never ``eval``-ed, and the file is read with ``weights_only=True`` (a ``state_dict`` holds tensors only)."""
    import glob
    import os
    import re
    assert os.path.exists(saved_path), "{} not found".format(saved_path)

    def load(path):
        try:
            sd = torch.load(path, map_location="cpu", weights_only=True)
        except TypeError:                       # (a torch without the keyword)
            sd = torch.load(path, map_location="cpu")
        model.load_state_dict(sd, strict=False)

    best = glob.glob(os.path.join(saved_path, "net_epoch_bestval_at*.pth"))
    if best:
        assert len(best) == 1
        m = re.fullmatch(r"net_epoch_bestval_at(\d+)\.pth", os.path.basename(best[0]))
        if m is None:
            raise ValueError(f"cannot read the epoch out of '{os.path.basename(best[0])}'")
        epoch = int(m.group(1))
        print("resuming best validation model at epoch %d" % epoch)
        load(best[0])
        return epoch, model
    epochs = []
    for f in glob.glob(os.path.join(saved_path, "*epoch*.pth")):
        m = re.search(r"epoch(\d+)\.pth", os.path.basename(f))
        if m is not None:
            epochs.append(int(m.group(1)))
    initial_epoch = max(epochs) if epochs else 0
    if initial_epoch > 0:
        print("resuming by loading epoch %d" % initial_epoch)
        load(os.path.join(saved_path, "net_epoch%d.pth" % initial_epoch))
    return initial_epoch, model


def to_device(inputs, device):
    """Recursive ``.to(device)`` over lists / dicts; non-tensors pass through (opencood/tools/train_utils.py:249-258)."""
    if isinstance(inputs, list):
        return [to_device(x, device) for x in inputs]
    if isinstance(inputs, dict):
        return {k: to_device(v, device) for k, v in inputs.items()}
    if isinstance(inputs, (int, float, str)) or inputs is None or not hasattr(inputs, "to"):
        return inputs
    return inputs.to(device)
