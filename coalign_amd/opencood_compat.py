"""Make the hot-path classes resolvable under the reference's ``opencood.*`` import paths (SURVEY §8b "Discovery").

``train_utils.create_model`` imports ``opencood.models.<core_method>`` and picks the class whose lower-cased name equals
``core_method`` without underscores (opencood/tools/train_utils.py:127-146).  ``install()`` registers module objects
under exactly those dotted names, backed by this package:

* if no ``opencood`` package is importable, synthetic modules are created, so unchanged yamls
  (``core_method: point_pillar_baseline_multiscale``) and scripts written against the reference API work as is;
* if a real ``opencood`` checkout is importable, ``install(override=True)`` replaces just the hot-path symbols in it
  (the drop-in a maintainer would ship).
"""
from __future__ import annotations

import importlib
import sys
import types
from typing import Dict

from . import backbone, box_align, config, detector, encoder, evaluation, fusion, pcdet, pose, postprocess, preprocess

# dotted module name -> {attribute: object}
_EXPORTS: Dict[str, Dict[str, object]] = {
    "opencood.models.point_pillar_baseline_multiscale": {"PointPillarBaselineMultiscale": detector.PointPillarBaselineMultiscale},
    "opencood.models.point_pillar_coalign": {"CoAlign": detector.CoAlign},
    "opencood.models.point_pillar": {"PointPillar": detector.PointPillar},
    "opencood.models.sub_modules.pillar_vfe": {"PillarVFE": encoder.PillarVFE, "PFNLayer": encoder.PFNLayer},
    "opencood.models.sub_modules.point_pillar_scatter": {"PointPillarScatter": encoder.PointPillarScatter},
    "opencood.models.sub_modules.base_bev_backbone_resnet": {"ResNetBEVBackbone": backbone.ResNetBEVBackbone},
    "opencood.models.sub_modules.base_bev_backbone": {"BaseBEVBackbone": backbone.BaseBEVBackbone},
    "opencood.models.sub_modules.downsample_conv": {"DownsampleConv": backbone.DownsampleConv, "DoubleConv": backbone.DoubleConv},
    "opencood.models.sub_modules.naive_compress": {"NaiveCompressor": backbone.NaiveCompressor},
    "opencood.models.sub_modules.torch_transformation_utils": {"warp_affine_simple": fusion.warp_affine_simple},
    "opencood.models.fuse_modules.fusion_in_one": {"AttFusion": fusion.AttFusion, "MaxFusion": fusion.MaxFusion,
                                                   "regroup": fusion.regroup, "warp_feature": fusion.warp_feature},
    "opencood.models.fuse_modules.fuse_utils": {"regroup": fusion.regroup},
    "opencood.utils.transformation_utils": {"normalize_pairwise_tfm": pose.normalize_pairwise_tfm, "x_to_world": pose.x_to_world,
                                            "get_pairwise_transformation": pose.get_pairwise_transformation},
    "opencood.utils.pose_utils": {"generate_noise": pose.generate_noise},
    "opencood.utils.box_utils": {"nms_rotated": postprocess.nms_rotated},
    "opencood.data_utils.post_processor.voxel_postprocessor": {"VoxelPostprocessor": postprocess.VoxelPostprocessor},
    "opencood.data_utils.post_processor": {"build_postprocessor": postprocess.build_postprocessor,
                                           "VoxelPostprocessor": postprocess.VoxelPostprocessor},
    "opencood.hypes_yaml.yaml_utils": {"load_yaml": config.load_yaml, "load_point_pillar_params": config.load_point_pillar_params},
    "opencood.tools.train_utils": {"create_model": detector.build_model, "to_device": detector.to_device, "load_saved_model": detector.load_saved_model},
    # rows N and the "next" rows of SURVEY section 8
    "opencood.pcdet_utils.iou3d_nms.iou3d_nms_utils": {"nms_gpu": pcdet.nms_gpu, "boxes_iou_bev": pcdet.boxes_iou_bev,
                                                       "boxes_iou3d_gpu": pcdet.boxes_iou3d_gpu, "nms_normal_gpu": pcdet.nms_normal_gpu},
    "opencood.models.point_pillar_uncertainty": {"PointPillarUncertainty": detector.PointPillarUncertainty},
    "opencood.data_utils.post_processor.uncertainty_voxel_postprocessor": {"UncertaintyVoxelPostprocessor": postprocess.UncertaintyVoxelPostprocessor},
    "opencood.data_utils.pre_processor.sp_voxel_preprocessor": {"SpVoxelPreprocessor": preprocess.SpVoxelPreprocessor},
    "opencood.data_utils.pre_processor": {"build_preprocessor": preprocess.build_preprocessor, "SpVoxelPreprocessor": preprocess.SpVoxelPreprocessor},
    "opencood.models.sub_modules.box_align_v2": {"box_alignment_relative_sample_np": box_align.box_alignment_relative_sample_np,
                                                 "box_alignment_relative_np": box_align.box_alignment_relative_np},
    "opencood.models.sub_modules.pose_graph_optim": {"PoseGraphOptimization2D": box_align.PoseGraphOptimization2D},
    "opencood.utils.eval_utils": {"caluclate_tp_fp": evaluation.caluclate_tp_fp, "calculate_ap": evaluation.calculate_ap, "voc_ap": evaluation.voc_ap,
                                  "eval_final_results": evaluation.eval_final_results},
}


def _ensure_module(name: str) -> types.ModuleType:
    if name in sys.modules:
        return sys.modules[name]
    mod = types.ModuleType(name)
    mod.__dict__["__coalign_amd_synthetic__"] = True
    if "." in name:
        parent_name, _, child = name.rpartition(".")
        parent = _ensure_module(parent_name)
        setattr(parent, child, mod)
    mod.__path__ = []  # behave like a package so that dotted imports of children resolve through sys.modules
    sys.modules[name] = mod
    return mod


def install(override: bool = False) -> Dict[str, str]:
    """Register the aliases.  Returns {module name: 'synthetic' | 'patched' | 'kept'}."""
    report = {}
    real = False
    if "opencood" not in sys.modules or not getattr(sys.modules["opencood"], "__coalign_amd_synthetic__", False):
        try:
            importlib.import_module("opencood")
            real = True
        except Exception:  # noqa: BLE001
            real = False
    for name, attrs in _EXPORTS.items():
        if real:
            try:
                mod = importlib.import_module(name)
            except Exception:  # noqa: BLE001  (module needs an optional dependency that is absent)
                mod = _ensure_module(name)
            if override or getattr(mod, "__coalign_amd_synthetic__", False):
                for k, v in attrs.items():
                    setattr(mod, k, v)
                report[name] = "patched"
            else:
                report[name] = "kept"
        else:
            mod = _ensure_module(name)
            for k, v in attrs.items():
                setattr(mod, k, v)
            report[name] = "synthetic"
    return report
