"""ctypes binding of ``include/coalign_amd.h`` (the C ABI of the gfx950 kernels).

There is NO fallback: if ``coalign_amd/lib/libcoalign_hip.so`` is missing and cannot be built with hipcc the
import of this module's ``lib()`` raises, and every op in :mod:`coalign_amd.ops` raises on non-GPU tensors.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_size_t, c_uint8, c_uint32, c_void_p

from . import build as _build

_LIB = None

P = c_void_p  # every device / host buffer crosses the ABI as a plain pointer

# name -> (restype, argtypes); mirrors include/coalign_amd.h one to one
SIGNATURES = {
    "coalign_abi_version": (c_int, []),
    "coalign_status_string": (c_char_p, [c_int]),
    "coalign_last_hip_error": (c_char_p, []),
    "coalign_pillar_scatter_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "coalign_pillar_vfe_scatter": (c_int, [P, P, P, c_int, c_int, P, P, P, P, P, P, c_float, c_int, c_int, c_int,
                                           POINTER(c_double), POINTER(c_double), c_int, c_int, c_int, P, P, P, c_size_t, P]),
    "coalign_pillar_vfe_scatter_nhwc": (c_int, [P, P, P, c_int, c_int, P, P, P, P, P, P, c_float, c_int, c_int, c_int,
                                                POINTER(c_double), POINTER(c_double), c_int, c_int, c_int, P, P, P, c_size_t, P]),
    "coalign_pillar_encode_persistent": (c_int, [P, P, P, c_int, c_int, P, P, P, P, P, P, c_float, c_int, c_int, c_int,
                                                 POINTER(c_double), POINTER(c_double), c_int, c_int, c_int, P, P, c_int, P, P, P]),
    "coalign_pillar_encode_stream": (c_int, [P, P, P, c_int, P, c_int, P, P, P, P, P, P, c_float, c_int, c_int, c_int,
                                             POINTER(c_double), POINTER(c_double), c_int, c_int, c_int, P, P, P, P, c_int, P]),
    "coalign_sparse_canvas_stamp_bytes": (c_size_t, [c_int, c_int, c_int]),
    "coalign_sparse_canvas_state_bytes": (c_size_t, []),
    "coalign_pillar_folded_param_bytes": (c_size_t, []),
    "coalign_pillar_fold_params": (c_int, [P, P, P, P, P, P, c_float, c_int, c_int, P, P]),
    "coalign_pillar_encode_sparse": (c_int, [P, P, P, c_int, P, c_int, P, c_int, c_int, POINTER(c_double), POINTER(c_double),
                                             c_int, c_int, c_int, P, P, P, P]),
    "coalign_pillar_encode_sparse_frame": (c_int, [P, c_int, c_int, P, c_int, c_int, POINTER(c_double), POINTER(c_double), c_int, c_int, c_int, P, P, P, P]),
    "coalign_conv3x3_emu_sparse": (c_int, [P, c_int, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P]),
    "coalign_pointwise_conv_emu_sparse": (c_int, [P, c_int, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "coalign_scatter_to_bev": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, P, P, c_size_t, P]),
    "coalign_warp_fuse": (c_int, [P, c_int, c_int, c_int, c_int, P, POINTER(c_int32), c_int, c_int, P, c_int, c_int, P]),
    "coalign_normalize_pairwise": (c_int, [P, c_int, c_int, c_int, c_double, c_double, P, P]),
    "coalign_warp_fuse_rows": (c_int, [P, c_int, c_int, c_int, c_int, P, POINTER(c_int32), c_int, POINTER(c_int32), c_int, P, c_int, c_int, P]),
    "coalign_warp_fuse_nhwc": (c_int, [c_int, POINTER(P), POINTER(c_int32), POINTER(c_int32), POINTER(c_int32), POINTER(P), POINTER(c_int32),
                                       POINTER(c_int32), c_int, P, POINTER(c_int32), c_int, P]),
    "coalign_anchor_decode_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "coalign_anchor_decode": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_float, c_float, c_int, P, c_int, P, P, P,
                                      P, P, P, P, P, P, c_size_t, P]),
    "coalign_anchor_decode_first": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_float, c_float, c_int, P, c_int, P, P, P,
                                            P, P, P, P, P, P, c_size_t, P, c_int, P]),
    "coalign_nms_rotated_workspace_bytes": (c_size_t, [c_int, c_int]),
    "coalign_nms_rotated": (c_int, [P, c_int, c_int, P, P, c_int, P, c_float, c_int, P, P, P, c_size_t, P]),
    "coalign_nms_rotated_gather": (c_int, [P, P, P, c_int, P, c_float, c_int, P, P, POINTER(c_double), P, P, P, P, c_size_t, P]),
    "coalign_gather_in_range": (c_int, [P, P, P, P, c_int, POINTER(c_double), P, P, P, P]),
    "coalign_iou_rotated_matrix": (c_int, [P, c_int, c_int, c_int, P, c_int, c_int, c_int, P, P]),
    "coalign_boxes_iou_bev": (c_int, [P, c_int, P, c_int, P, P]),
    "coalign_boxes_overlap_bev": (c_int, [P, c_int, P, c_int, P, P]),
    "coalign_pcdet_nms_workspace_bytes": (c_size_t, [c_int]),
    "coalign_pcdet_nms": (c_int, [P, c_int, c_float, c_int, P, P, P, c_size_t, P]),
    "coalign_bias_act": (c_int, [P, P, P, c_int, c_int, c_int, c_int, P]),
    "coalign_fill_words": (c_int, [P, c_size_t, c_uint32, P]),
    "coalign_voxelize_capacity": (c_int64, [c_int64, c_int, POINTER(c_double), POINTER(c_double), c_int]),
    "coalign_voxelize_workspace_bytes": (c_size_t, [POINTER(c_int64), c_int, POINTER(c_double), POINTER(c_double), c_int]),
    "coalign_conv3x3_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "coalign_conv3x3_bias_act": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P, c_size_t, P]),
    "coalign_conv3x3_emu_weight_bytes": (c_size_t, [c_int, c_int, c_int]),
    "coalign_conv3x3_emu_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "coalign_conv3x3_emu_weight_bytes_ex": (c_size_t, [c_int, c_int, c_int, c_int]),
    "coalign_conv3x3_emu_workspace_bytes_ex": (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "coalign_conv3x3_emu_bias_act": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, c_size_t, P]),
    "coalign_conv3x3_emu_ex": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P, c_size_t, P]),
    "coalign_sp_map_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "coalign_sp_pack": (c_int, [P, c_int, P, c_int, c_int, c_int, c_int, P, P]),
    "coalign_sp_unpack": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "coalign_conv3x3_sp_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "coalign_conv3x3_sp": (c_int, [P, P, P, P, c_int, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P, c_size_t, P]),
    "coalign_conv3x3_sp_both": (c_int, [P, P, P, P, c_int, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P, c_size_t, P]),
    "coalign_sp_rows_bytes": (c_size_t, [c_int, c_int]),
    "coalign_sp_pack_rows": (c_int, [P, c_int, P, c_int, P, P, P]),
    "coalign_conv3x3_sp_s2": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P, P]),
    "coalign_conv3x3_sp_s2_sparse": (c_int, [P, c_int, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P, P]),
    "coalign_conv1x1_sp_weight_bytes": (c_size_t, [c_int, c_int]),
    "coalign_conv3x3_sp_s2_skip": (c_int, [P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P, P]),
    "coalign_conv3x3_sp_s2_skip_sparse": (c_int, [P, c_int, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P, P]),
    "coalign_pointwise_conv": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "coalign_pointwise_conv_ex": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "coalign_pointwise_emu_weight_bytes": (c_size_t, [c_int, c_int]),
    "coalign_pointwise_conv_emu": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "coalign_pointwise_conv_emu_sp": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P]),
    "coalign_pointwise_conv_emu_sp_multi": (c_int, [c_int, POINTER(P), POINTER(P), POINTER(P), POINTER(c_int32), POINTER(c_int32), POINTER(c_int32), POINTER(c_int32), POINTER(c_int32),
                                            POINTER(c_int32), POINTER(c_int32), P, c_int, c_int, c_int, P, P]),
    "coalign_heads_sp": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "coalign_pose_graph_workspace_bytes": (c_size_t, [c_int]),
    "coalign_pose_graph_optimize": (c_int, [c_int, P, P, P, c_int, P, P, P, P, P, P, c_int, P, P, c_size_t, P]),
    "coalign_voxelize": (c_int, [P, POINTER(c_int64), c_int, POINTER(c_double), POINTER(c_double), c_int, c_int, c_int,
                                 POINTER(c_double), P, P, P, c_int64, P, P, c_size_t, P]),
}


# include/coalign_amd_lab.h: entry points of the laboratory library only (measured, not adopted)
LAB_SIGNATURES = {
    "coalign_conv3x3_wino_weight_bytes": (c_size_t, [c_int, c_int]),
    "coalign_conv3x3_wino": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P]),
}

_LAB_LIB = None


class CoalignHipError(RuntimeError):
    pass


def lib_path() -> str:
    return _build.LIB_PATH


def lib() -> ctypes.CDLL:
    """Load (building first if needed) the gfx950 library.  Raises if that is impossible."""
    global _LIB
    if _LIB is not None:
        return _LIB
    import torch  # noqa: F401  -- load PyTorch-ROCm's HIP runtime first so both share one libamdhip64.so.7
    lab = os.environ.get("COALIGN_LAB", "0")             # tools/ only: "1" = the laboratory build with its ablation switches, "vec" = + packed fp32 allowed (build.py)
    path = _build.LABVEC_LIB_PATH if lab == "vec" else _build.LAB_LIB_PATH if lab == "1" else _build.LIB_PATH
    if not os.path.exists(path):
        try:
            _build.build(lab=lab == "1", vectorize=lab == "vec")
        except Exception as exc:  # noqa: BLE001
            raise CoalignHipError(
                f"{path} is missing and could not be built ({exc}); the CoAlign hot path has no CPU fallback") from exc
    handle = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(handle, name)  # AttributeError here == header / library mismatch
        fn.restype = res
        fn.argtypes = args
    if handle.coalign_abi_version() != 2:
        raise CoalignHipError(f"ABI version mismatch: library reports {handle.coalign_abi_version()}, binding expects 2")
    _LIB = handle
    return handle


def lab_lib() -> ctypes.CDLL:
    """The laboratory library (product sources + -DCOALIGN_LAB + include/coalign_amd_lab.h's kernels), loaded beside the product library: the Winograd tests
    and tools call through it; nothing in the detector's default routes does."""
    global _LAB_LIB
    if _LAB_LIB is not None:
        return _LAB_LIB
    import torch  # noqa: F401
    path = _build.LAB_LIB_PATH
    if not os.path.exists(path):
        try:
            _build.build(lab=True)
        except Exception as exc:  # noqa: BLE001
            raise CoalignHipError(f"{path} is missing and could not be built ({exc})") from exc
    handle = ctypes.CDLL(path)
    for name, (res, args) in {**SIGNATURES, **LAB_SIGNATURES}.items():
        fn = getattr(handle, name)
        fn.restype = res
        fn.argtypes = args
    _LAB_LIB = handle
    return handle


def check(status: int, what: str) -> None:
    if status != 0:
        L = lib()
        msg = L.coalign_status_string(status).decode()
        if status == -5:
            msg += " [" + L.coalign_last_hip_error().decode() + "]"
        raise CoalignHipError(f"{what}: {msg}")
