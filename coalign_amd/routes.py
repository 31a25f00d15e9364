"""Which kernel serves which layer of a config -- decided from the config alone, no GPU needed.

The detector picks its kernels per layer at run time (``backbone.conv3x3_fused``, ``_pointwise_ok``, ``fusion.fuse_multiscale``,
``launch_rows`` in csrc/pillar_scatter.hip); a layer whose shape a hand-written kernel does not take falls back to MIOpen / the
per-scale NCHW fusion kernel / the fp32 VALU encoder -- correct, slower, and until round 3 silent.  ``plan(hypes)`` restates those
rules on the model built from a hypes dictionary and lists the route of every layer, so that a yaml that would leave the fast path
shows up in a CPU test (tests/test_host_cpu.py walks the reference's ``hypes_yaml/**/pointpillar*.yaml`` with it) instead of in a
profile.

    python -m coalign_amd.routes <hypes.yaml> [...]          # prints the plan(s) as JSON
"""
from __future__ import annotations

import json
import sys
from typing import Dict

import torch.nn as nn

from . import backbone as bb
from .detector import MODEL_REGISTRY, build_model

EMU, F32, MIOPEN, ROCBLAS, POINTWISE = "conv3x3_emu (split 16-bit matrix cores)", "conv3x3 (fp32 matrix cores) / MIOpen by shape", "MIOpen", "rocBLAS (1x1 heads)", "pointwise"
WINO = "conv3x3_wino (Winograd F(2x2,3x3), split-bf16 matrix cores)"
SP = "conv3x3_sp (SplitMap input: operands by LDS-DMA, fp16 x 2)"
DEFAULT_TERMS = 16      # backbone.CONV_EMU_TERMS: the 2-way fp16 split since round 4


def _conv3x3_route(conv: nn.Conv2d, terms: int) -> str:
    """backbone.conv3x3_fused / packable: Cout % 64 == 0, Cin % 8 == 0, stride 1 or 2 -> the hand-written kernels."""
    s = conv.stride[0]
    if tuple(conv.kernel_size) != (3, 3) or conv.stride[0] != conv.stride[1]:
        return MIOPEN
    ok = conv.out_channels % 64 == 0 and conv.in_channels % 8 == 0 and s in (1, 2)
    if not ok:
        return MIOPEN + f" (unpackable: Cout {conv.out_channels} % 64 or Cin {conv.in_channels} % 8 or stride {s})"
    if terms in (2, 3, 16):
        if terms == 16 and s == 1 and conv.in_channels % 16:
            return F32 + " (the fp16 split serves the tap-major and the strided images)"
        if terms == 3 and s == 1 and bb.CONV_WINOGRAD and conv.in_channels % 16 == 0:
            return WINO + " when its input is channels-last (COALIGN_WINOGRAD=1)"
        return EMU + (", tap-major image" if s == 1 and conv.in_channels % 16 == 0 else ", tap-pair image") + {2: ", bf16 x 2", 3: ", bf16 x 3", 16: ", fp16 x 2"}[terms]
    return F32 if s == 1 else MIOPEN + " (native mode, strided)"


def _pointwise_route(cin: int, terms: int) -> str:
    """backbone.PointwisePack.get: split-bf16 image when the 3x3 layers use the 3-way split and Cin % 16 == 0, else the fp32 kernel."""
    return POINTWISE + (" (split-bf16 matrix cores)" if terms in (3, 16) and cin % 16 == 0 and bb.POINTWISE_EMU else " (fp32 matrix cores)")


def plan(hypes: dict, terms: int = DEFAULT_TERMS) -> Dict[str, object]:
    """-> {"model", "layers": {module name: route}, "pillar", "fusion", "fallbacks": [names of 3x3 / pointwise layers NOT on a hand-written
    kernel], "outside_hot_path": reason or None}."""
    name = hypes["model"]["core_method"]
    if name not in MODEL_REGISTRY:
        return {"model": name, "outside_hot_path": f"model family '{name}' is not part of the CoAlign hot path", "layers": {}, "fallbacks": []}
    model = build_model(hypes)
    layers: Dict[str, str] = {}
    fallbacks = []

    def note(n, route, is_fallback):
        layers[n] = route
        if is_fallback:
            fallbacks.append(n)

    for n, m in model.named_modules():
        if isinstance(m, bb.NaiveCompressor):
            for cn, c in m.named_modules():
                if isinstance(c, nn.Conv2d):
                    layers[f"{n}.{cn}"] = MIOPEN + " (compressor: SURVEY 8a row D keeps it on the library)"
        elif isinstance(m, nn.Conv2d) and "naive_compressor" not in n:
            if tuple(m.kernel_size) == (3, 3):
                r = _conv3x3_route(m, terms)
                note(n, r, r.startswith(MIOPEN))
            elif tuple(m.kernel_size) == (1, 1) and n.endswith("_head"):
                # detector._run_heads: the merged 1x1 heads on the pointwise kernel (GEMM rows padded to 32) when the split-bf16 image exists
                ok = terms in (3, 16) and bb.POINTWISE_EMU and m.in_channels % 16 == 0 and m.in_channels <= 512
                note(n, _pointwise_route(m.in_channels, terms) + ", merged 1x1 heads" if ok else ROCBLAS, not ok)
            elif tuple(m.kernel_size) == (1, 1) and ".downsample." in n:      # BasicBlock skip: pointwise kernel when stride 2, Cin even and <= 256
                ok = m.stride[0] == 2 and m.in_channels % 2 == 0 and m.in_channels <= 256
                note(n, _pointwise_route(m.in_channels, terms) if ok else MIOPEN + " (skip convolution outside the pointwise kernel's shapes)", not ok)
            else:
                note(n, MIOPEN, True)
    # round 5: inside a ResNet stage and in the shrink header the fp16 mode hands SplitMaps from 3x3 layer to 3x3 layer (backbone.BasicBlock.takes_split_maps,
    # DoubleConv.forward): the stride-1 layers read them with coalign_conv3x3_sp, the layer in front of a chain writes the first one
    backbone = getattr(model, "backbone", None)
    heads_ok = backbone is not None and len(getattr(backbone, "deblocks", [])) > 0          # the up-sampling heads are on the pointwise kernel (rule of _pointwise_ok)
    if heads_ok:
        for i in range(backbone.num_levels):
            op = backbone.deblocks[i][0]
            heads_ok = heads_ok and isinstance(op, nn.ConvTranspose2d) and op.kernel_size == op.stride and op.stride[0] == op.stride[1] and op.stride[0] in (1, 2, 4)
            heads_ok = heads_ok and op.in_channels <= 256 and op.in_channels % 2 == 0 and (op.out_channels * op.stride[0] ** 2) % 32 == 0
    # second half of round 5: the heads write the concatenated map as ONE SplitMap when the shrink header's first DoubleConv takes it (detector.fuse_and_head)
    first_shrink = getattr(model, "shrink_conv", None)
    first_shrink = first_shrink.layers[0] if getattr(model, "shrink_flag", False) and first_shrink is not None and len(first_shrink.layers) else None
    heads_split = False
    if terms == 16 and bb.SPLIT_MAPS and bb.NHWC_STAGE_OUTPUTS and bb.CONV_EMU_TAP_MAJOR and bb.POINTWISE_EMU:
        for n, m in model.named_modules():
            if isinstance(m, bb.BasicBlock):
                m.eval()
                if m.takes_split_maps() and f"{n}.conv1" in layers and f"{n}.conv2" in layers:
                    layers[f"{n}.conv1"] = SP if m.stride == 1 else layers[f"{n}.conv1"] + ", SplitMap out"
                    layers[f"{n}.conv2"] = SP
            elif isinstance(m, bb.DoubleConv):
                c1, c2 = m.double_conv[0], m.double_conv[2]
                ok = all(tuple(c.kernel_size) == (3, 3) and c.stride == (1, 1) and c.out_channels % 64 == 0 and c.in_channels % 16 == 0 for c in (c1, c2))
                if ok and f"{n}.double_conv.0" in layers:
                    from_heads = (m is first_shrink and bb.HEAD_SPLIT_MAPS and heads_ok and len(backbone.deblocks) == backbone.num_levels
                                  and all(backbone.deblocks[i][0].in_channels % 16 == 0 and backbone.deblocks[i][0].out_channels % 16 == 0 for i in range(backbone.num_levels)))
                    heads_split = heads_split or from_heads
                    layers[f"{n}.double_conv.0"] = SP if from_heads else layers[f"{n}.double_conv.0"] + ", SplitMap out"
                    layers[f"{n}.double_conv.2"] = SP
    if backbone is not None and len(getattr(backbone, "deblocks", [])):
        ok = heads_ok
        for i in range(len(backbone.deblocks)):
            note(f"backbone.deblocks.{i}", _pointwise_route(backbone.deblocks[i][0].in_channels, terms) + (", writes its slice of the concatenated SplitMap" if heads_split else ", writes its slice of the concatenation")
                 if ok and i < backbone.num_levels else MIOPEN + " + bias_act", not (ok and i < backbone.num_levels))
    vfe = getattr(model, "pillar_vfe", None)
    pillar = None
    if vfe is not None:
        P = int(hypes.get("preprocess", {}).get("args", {}).get("max_points_per_voxel", 32))
        C = vfe.num_filters[-1]
        if len(vfe.pfn_layers) != 1:
            pillar = "unsupported: stacked PFN layers"
        elif vfe.with_distance or P > 32 or C > 64:
            pillar = "fp32 VALU encoder (distance feature / P > 32 / C > 64)"
        else:
            rn = getattr(getattr(model, "backbone", None), "resnet", None)
            first = rn.layer0[0] if rn is not None and hasattr(rn, "layer0") else None
            sparse = (terms in (3, 16) and first is not None and first.stride == 2 and first.downsample is not None and first.conv1.out_channels % 64 == 0 and
                      first.conv1.in_channels % 16 == 0 and first.downsample[0].out_channels % 32 == 0 and "compression" not in hypes["model"]["args"])
            pillar = ("matrix-core encoder (one fp16 matrix instruction per pillar and 32 channels on a 22-bit operand split), ONE launch, sparse canvas read by the first ResNet block" if sparse else
                      "matrix-core encoder (linearised PFN, split-bf16), persistent dense canvas" if terms in (2, 3, 16) else "matrix-core encoder, NCHW strip writer")
    fusion = None
    if hasattr(model, "fusion_net"):
        dims = [int(d) for d in hypes["model"]["args"]["base_bev_backbone"]["num_filters"]]
        if len(model.fusion_net) != len(dims):
            dims = dims[-len(model.fusion_net):]
        feat = [getattr(f, "feature_dims", None) for f in model.fusion_net]
        if terms in (2, 3, 16) and len(dims) <= 3 and all(d in (64, 128, 256) for d in dims) and all(fd in (None, d) for fd, d in zip(feat, dims)):
            fusion = "warp_fuse_nhwc: all scales in one launch (channels-last)"
        else:
            fusion = "warp_fuse: one launch per scale (NCHW, LDS-staged patches)"
            fallbacks.append("fusion")
    return {"model": name, "outside_hot_path": None, "layers": layers, "pillar": pillar, "fusion": fusion, "fallbacks": fallbacks}


def summary(p: dict) -> dict:
    """Counts per route + the fallback list: what the walk test records per yaml."""
    counts: Dict[str, int] = {}
    for r in p["layers"].values():
        key = r.split(" (")[0].split(",")[0]
        counts[key] = counts.get(key, 0) + 1
    return {"model": p["model"], "outside_hot_path": p["outside_hot_path"], "routes": counts, "pillar": p.get("pillar"), "fusion": p.get("fusion"),
            "fallbacks": p["fallbacks"]}


if __name__ == "__main__":
    from .config import load_yaml
    out = {}
    for path in sys.argv[1:]:
        try:
            out[path] = summary(plan(load_yaml(path)))
        except Exception as e:      # noqa: BLE001  (other model families' yamls need parsers / keys outside the hot path)
            out[path] = {"outside_hot_path": f"{type(e).__name__}: {str(e)[:120]}"}
    print(json.dumps(out, indent=1))
