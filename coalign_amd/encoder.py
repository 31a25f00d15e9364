"""Pillar encoder + BEV scatter modules, host side.

Same class names, constructor arguments, ``batch_dict`` keys and ``state_dict`` names as
opencood/models/sub_modules/pillar_vfe.py (``PFNLayer`` :10-53, ``PillarVFE`` :56-155) and
opencood/models/sub_modules/point_pillar_scatter.py (``PointPillarScatter`` :5-72); the arithmetic runs in the
fused gfx950 kernel ``coalign_pillar_vfe_scatter`` (one pass produces both the pillar features and the dense
canvas, so ``PointPillarScatter`` normally just hands over the canvas ``PillarVFE`` already produced).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops


def host_ints(t) -> list:
    """record_len-like value -> python ints.  A CUDA tensor costs one host sync (the reference pays the same in
    ``regroup``'s ``.cpu()``, fusion_in_one.py:23); pass a CPU tensor / list to avoid it."""
    if torch.is_tensor(t):
        return [int(v) for v in t.tolist()]
    return [int(v) for v in t]


class PFNLayer(nn.Module):
    """Parameter container: Linear(in, out, bias = not use_norm) [+ BatchNorm1d(out, eps=1e-3, momentum=0.01)]."""

    def __init__(self, in_channels: int, out_channels: int, use_norm: bool = True, last_layer: bool = False):
        super().__init__()
        self.last_vfe = last_layer
        self.use_norm = use_norm
        if not last_layer:
            out_channels = out_channels // 2
        self.linear = nn.Linear(in_channels, out_channels, bias=not use_norm)
        if use_norm:
            self.norm = nn.BatchNorm1d(out_channels, eps=1e-3, momentum=0.01)
        self.part = 50000


class PillarVFE(nn.Module):
    def __init__(self, model_cfg: dict, num_point_features: int, voxel_size, point_cloud_range):
        super().__init__()
        self.model_cfg = model_cfg
        self.use_norm = model_cfg["use_norm"]
        self.with_distance = model_cfg["with_distance"]
        self.use_absolute_xyz = model_cfg["use_absolute_xyz"]
        if num_point_features != 4:
            raise NotImplementedError("the gfx950 pillar kernel reads (x, y, z, intensity) points")
        cin = num_point_features + (6 if self.use_absolute_xyz else 3) + (1 if self.with_distance else 0)
        self.num_filters = list(model_cfg["num_filters"])
        assert len(self.num_filters) > 0
        dims = [cin] + self.num_filters
        self.pfn_layers = nn.ModuleList(
            PFNLayer(dims[i], dims[i + 1], self.use_norm, last_layer=(i >= len(dims) - 2)) for i in range(len(dims) - 1))
        self.voxel_size = [float(v) for v in voxel_size]
        self.point_cloud_range = [float(v) for v in point_cloud_range]
        self.voxel_x, self.voxel_y, self.voxel_z = self.voxel_size
        self.x_offset = self.voxel_x / 2 + self.point_cloud_range[0]
        self.y_offset = self.voxel_y / 2 + self.point_cloud_range[1]
        self.z_offset = self.voxel_z / 2 + self.point_cloud_range[2]
        # Opt-in (FramePipeline turns it on): keep one canvas per HIP stream alive and clear only the rows the previous frame wrote instead
        # of zero-filling 36 MB per agent every frame.  The returned ``spatial_features`` is then overwritten by the next forward on the
        # same stream -- fine for a detector that runs its backbone right away, wrong for a caller that holds canvases across frames.
        self.persistent_canvas = False
        # Opt-in (the detector's fused forward turns it on): hand the backbone a SparseCanvas (feature rows + cell stamps, csrc/pillar_sparse.hip) instead
        # of a dense tensor.  ``spatial_features`` is then a ``ops.SparseCanvas`` (``.shape`` as the tensor's, ``.dense()`` materialises the reference's tensor).
        self.sparse_canvas = False
        self.nx = int(round((self.point_cloud_range[3] - self.point_cloud_range[0]) / self.voxel_x))
        self.ny = int(round((self.point_cloud_range[4] - self.point_cloud_range[1]) / self.voxel_y))

    def get_output_feature_dim(self) -> int:
        return self.num_filters[-1]

    def forward(self, batch_dict: dict) -> dict:
        if len(self.pfn_layers) != 1:
            raise NotImplementedError("stacked PFN layers are outside the CoAlign hot path (configs use num_filters: [64])")
        if self.training:
            raise NotImplementedError("the gfx950 pillar kernel implements eval-mode BatchNorm (inference hot path)")
        vf, npts, coords = batch_dict["voxel_features"], batch_dict["voxel_num_points"], batch_dict["voxel_coords"]
        if "record_len" in batch_dict:
            n_agents = sum(host_ints(batch_dict["record_len"]))
        else:  # single-agent / late-fusion model: same rule (and same sync) as point_pillar_scatter.py:41
            n_agents = int(coords[:, 0].max().item()) + 1 if coords.shape[0] else 1
        pfn = self.pfn_layers[0]
        bn = (pfn.norm.weight, pfn.norm.bias, pfn.norm.running_mean, pfn.norm.running_var) if self.use_norm else None
        from . import backbone                       # the canvas layout follows the convolution route that will read it
        channels_last = backbone.NHWC_STAGE_OUTPUTS and backbone.emu_active() and backbone.FAST_INFERENCE
        count_dev = batch_dict.get("voxel_count_dev")
        if self.sparse_canvas and channels_last and not self.with_distance and vf.shape[1] <= 32 and self.num_filters[-1] <= 64:
            # round 4 fast path (the detector switches it on when its first ResNet block consumes it): ONE launch, feature rows + cell stamps, no dense canvas
            eps = pfn.norm.eps if self.use_norm else 0.0
            cache = self.__dict__.get("_folded_cache")
            if cache is None:
                cache = self.__dict__["_folded_cache"] = backbone._FoldCache()
            # the folded channel parameters: once per weight set (refreshed when a tensor of the PFN layer is replaced or written in place)
            folded = cache.get(pfn, lambda: ops.pillar_fold_params(pfn.linear.weight, pfn.linear.bias, bn, eps, self.use_absolute_xyz))
            sc = ops.pillar_encode_sparse(vf, npts, coords, pfn.linear.weight, pfn.linear.bias, bn, eps, self.use_absolute_xyz,
                                          self.voxel_size, self.point_cloud_range[:3], n_agents, self.ny, self.nx,
                                          canvas_cache=self.__dict__.setdefault("_canvas_cache", {}), count_dev=count_dev, folded=folded,
                                          frame=batch_dict.get("pillar_frame"))
            batch_dict["pillar_features"] = sc.feats
            batch_dict["_sparse_canvas"] = sc
            return batch_dict
        if batch_dict.get("pillar_frame") is not None:
            # (any other route would read the tensors in batch_dict, which a frame record's owner passes for their SHAPES only)
            raise ops.FrameRecordUnsupported("a pillar frame record is consumed by the sparse-canvas route only (eval, channels-last fp16/bf16 convolutions, "
                                             "P <= 32, C <= 64, no distance feature)")
        if count_dev is not None:
            # the producer (the device voxeliser) left the pillar count on the device: capacity-sized arrays, no host read of the count,
            # always the persistent channels-last canvas (FramePipeline.submit_points; include/coalign_amd.h coalign_pillar_encode_stream)
            if not channels_last:
                raise ops.hip.CoalignHipError("voxel_count_dev needs the channels-last canvas route (default convolution mode)")
            feats, canvas = ops.pillar_encode_stream(
                vf, npts, coords, count_dev, pfn.linear.weight, pfn.linear.bias, bn, pfn.norm.eps if self.use_norm else 0.0,
                self.use_absolute_xyz, self.with_distance, self.voxel_size, self.point_cloud_range[:3], n_agents, self.ny, self.nx,
                canvas_cache=self.__dict__.setdefault("_canvas_cache", {}), unique_cells=bool(batch_dict.get("voxel_cells_unique", True)),
                want_features=bool(batch_dict.get("want_pillar_features", False)))
            batch_dict["pillar_features"] = feats
            batch_dict["_fused_canvas"] = (feats, canvas)
            return batch_dict
        feats, canvas = ops.pillar_vfe_scatter(
            vf, npts, coords, pfn.linear.weight, pfn.linear.bias, bn, pfn.norm.eps if self.use_norm else 0.0,
            self.use_absolute_xyz, self.with_distance, self.voxel_size, self.point_cloud_range[:3], n_agents, self.ny, self.nx,
            channels_last=channels_last, canvas_cache=self.__dict__.setdefault("_canvas_cache", {}) if self.persistent_canvas else None)
        batch_dict["pillar_features"] = feats
        batch_dict["_fused_canvas"] = (feats, canvas)
        return batch_dict


class PointPillarScatter(nn.Module):
    def __init__(self, model_cfg: dict):
        super().__init__()
        self.model_cfg = model_cfg
        self.num_bev_features = model_cfg["num_features"]
        self.nx, self.ny, self.nz = [int(v) for v in model_cfg["grid_size"]]
        assert self.nz == 1

    def forward(self, batch_dict: dict) -> dict:
        feats, coords = batch_dict["pillar_features"], batch_dict["voxel_coords"]
        sc = batch_dict.pop("_sparse_canvas", None)
        if sc is not None and sc.feats is feats and sc.shape[1:] == (self.num_bev_features, self.ny, self.nx):
            batch_dict["spatial_features"] = sc              # (the scatter already happened: 8 bytes per pillar)
            return batch_dict
        fused = batch_dict.pop("_fused_canvas", None)
        if fused is not None and fused[0] is feats and tuple(fused[1].shape[1:]) == (self.num_bev_features, self.ny, self.nx):
            canvas = fused[1]
        else:  # pillar features produced elsewhere: scatter-only kernel
            if "record_len" in batch_dict:
                n_agents = sum(host_ints(batch_dict["record_len"]))
            else:
                n_agents = int(coords[:, 0].max().item()) + 1 if coords.shape[0] else 1
            canvas = ops.scatter_to_bev(feats, coords, n_agents, self.ny, self.nx)
        batch_dict["spatial_features"] = canvas
        return batch_dict
