"""Torch-tensor front end of the C ABI: allocates outputs / workspaces with torch (PyTorch-ROCm owns device
memory and streams), passes raw ``data_ptr()``s and the current stream to the gfx950 kernels.

Every function requires CUDA(ROCm) tensors and raises otherwise -- the hot path has no CPU implementation
in this package (the CPU restatement lives in ``oracle/`` and is test infrastructure only).
"""
from __future__ import annotations

import ctypes
from typing import Optional, Sequence, Tuple

import numpy as np
import torch

from . import hip

FUSE_ATT, FUSE_MAX, FUSE_NONE = 0, 1, 2

# Optional in-pipeline timing: set to a dict to have every op bracket its launches with HIP events recorded on the
# stream the kernels are launched on (bench.py reads the pairs after synchronising).  None = no overhead.
PROFILE = None


def timed(name: str):
    """Bracket a region with HIP events when ``PROFILE`` is active (used by bench.py for the per-stage breakdown)."""
    return _Timed(name)


class _Timed:
    def __init__(self, name: str):
        self.name = name

    def __enter__(self):
        if PROFILE is not None:
            self.s = torch.cuda.Event(enable_timing=True)
            self.e = torch.cuda.Event(enable_timing=True)
            self.s.record()
        return self

    def __exit__(self, *exc):
        if PROFILE is not None:
            self.e.record()
            PROFILE.setdefault(self.name, []).append((self.s, self.e))
        return False


def _stream() -> ctypes.c_void_p:
    """The current HIP stream of the current device -- which, inside an op, is the device of its tensors (``_device_op``)."""
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _device_op(fn):
    """Run ``fn`` with the device of its first GPU tensor argument current: outputs / workspaces are allocated there, and the
    kernels go to THAT device's current stream (not to the stream of whatever device happens to be current in the caller)."""
    import functools

    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        for a in args:
            if isinstance(a, DecodeBuffers):
                a = a.counts
            elif isinstance(a, SparseCanvas):
                a = a.feats
            if torch.is_tensor(a) and a.is_cuda:
                if a.device.index != torch.cuda.current_device():
                    with torch.cuda.device(a.device):
                        return fn(*args, **kwargs)
                break
        return fn(*args, **kwargs)
    return wrapper


def _ptr(t: Optional[torch.Tensor]) -> ctypes.c_void_p:
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def _rows_ptr(t: torch.Tensor) -> ctypes.c_void_p:
    """Pointer to the storage position of a (possibly zero-row) row tensor: ``data_ptr()`` of an empty view may be reported as NULL."""
    p = t.data_ptr()
    if p == 0 and t.untyped_storage().nbytes() > 0:
        p = t.untyped_storage().data_ptr() + t.storage_offset() * t.element_size()
    return ctypes.c_void_p(p)


def _need_gpu(*tensors: torch.Tensor) -> None:
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise hip.CoalignHipError("coalign_amd ops run on the MI355X only: got a CPU tensor (no CPU fallback exists)")


def _f32c(t: torch.Tensor) -> torch.Tensor:
    return t.contiguous() if t.dtype == torch.float32 else t.float().contiguous()


def _dbl3(v: Sequence[float]):
    return (ctypes.c_double * 3)(float(v[0]), float(v[1]), float(v[2]))


@_device_op
def pillar_vfe_scatter(voxel_features: torch.Tensor, voxel_num_points: torch.Tensor, voxel_coords: torch.Tensor,
                       weight: torch.Tensor, bias: Optional[torch.Tensor], bn: Optional[Tuple[torch.Tensor, ...]], bn_eps: float,
                       use_absolute_xyz: bool, with_distance: bool, voxel_size: Sequence[float], range_min: Sequence[float],
                       n_agents: int, ny: int, nx: int, channels_last: bool = False,
                       canvas_cache: Optional[dict] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """-> (pillar_features [M, C], canvas [n_agents, C, ny, nx]).  ``bn`` = (weight, bias, running_mean, running_var).
    ``channels_last``: the canvas comes back in NHWC memory (same logical shape), see coalign_pillar_vfe_scatter_nhwc;
    ``canvas_cache`` (channels-last only): a dict the caller keeps; the canvas is then a PERSISTENT tensor (one per device, stream and
    shape) of which only the rows the previous call wrote are cleared -- 10 MB instead of a 180 MB memset, two launches per frame
    (coalign_pillar_encode_persistent) -- and which the next call with the same cache on the same stream overwrites: for callers that
    consume the canvas before they encode the next frame.  (Also measured and rejected: the memset on a side stream next to the VALU-bound
    encoder -- 71 vs 64 us for the op and 212 vs 250 frames/s for the pipeline; cross-stream events cost more than the overlap wins.)"""
    _need_gpu(voxel_features, voxel_num_points, voxel_coords, weight)
    L = hip.lib()
    vf = _f32c(voxel_features)
    M, P = vf.shape[0], vf.shape[1]
    if vf.dim() != 3 or vf.shape[2] != 4:
        raise ValueError(f"voxel_features must be [M, P, 4], got {tuple(vf.shape)}")
    npts = voxel_num_points.to(torch.int32).contiguous()
    coords = voxel_coords.to(torch.int32).contiguous()
    w = _f32c(weight)
    C = w.shape[0]
    dev = vf.device
    feats = torch.empty((M, C), dtype=torch.float32, device=dev)
    channels_last = bool(channels_last) and P <= 32 and C <= 64 and C > 1 and ny * nx > 1
    entry = None
    if channels_last and canvas_cache is not None and C % 4 == 0:
        key = (str(dev), torch.cuda.current_stream(dev).cuda_stream, n_agents, C, ny, nx)
        entry = canvas_cache.get(key)
        if entry is None or entry["M"] < 0:              # first use (or a failed call before): a zeroed canvas, an all -1 cell map
            entry = canvas_cache[key] = {"canvas": torch.empty((n_agents, C, ny, nx), dtype=torch.float32, device=dev, memory_format=torch.channels_last).zero_(),
                                         "cellmap": torch.full((n_agents * ny * nx,), -1, dtype=torch.int32, device=dev),
                                         "dest": torch.empty(max(M, 1), dtype=torch.int32, device=dev), "M": 0}
        canvas = entry["canvas"]
    else:
        canvas = torch.empty((n_agents, C, ny, nx), dtype=torch.float32, device=dev,
                             memory_format=torch.channels_last if channels_last else torch.contiguous_format)
    bnp = [None] * 4 if bn is None else [_f32c(t) for t in bn]
    b = None if bias is None else _f32c(bias)
    args = (_ptr(vf), _ptr(npts), _ptr(coords), M, P, _ptr(w), _ptr(b), _ptr(bnp[0]), _ptr(bnp[1]), _ptr(bnp[2]), _ptr(bnp[3]), float(bn_eps), C,
            int(use_absolute_xyz), int(with_distance), _dbl3(voxel_size), _dbl3(range_min), n_agents, ny, nx, _ptr(feats))
    with _Timed("pillar_vfe_scatter"):
        if entry is not None:
            need = max(M, entry["M"], 1)
            if entry["dest"].numel() < need:
                grown = torch.empty(need, dtype=torch.int32, device=dev)
                grown[: entry["dest"].numel()] = entry["dest"]
                entry["dest"] = grown
            m_prev, entry["M"] = entry["M"], -1            # an exception below leaves the entry marked unusable
            hip.check(L.coalign_pillar_encode_persistent(*args, _ptr(entry["dest"]), m_prev, _ptr(canvas), _ptr(entry["cellmap"]), _stream()),
                      "coalign_pillar_encode_persistent")
            entry["M"] = M
        else:
            ws_bytes = L.coalign_pillar_scatter_workspace_bytes(n_agents, ny, nx)
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
            fn = L.coalign_pillar_vfe_scatter_nhwc if channels_last else L.coalign_pillar_vfe_scatter
            hip.check(fn(*args, _ptr(canvas), _ptr(ws), ws_bytes, _stream()), "coalign_pillar_vfe_scatter")
    return feats, canvas


@_device_op
def pillar_encode_stream(voxel_features: torch.Tensor, voxel_num_points: torch.Tensor, voxel_coords: torch.Tensor, count_dev: torch.Tensor,
                         weight: torch.Tensor, bias: Optional[torch.Tensor], bn: Optional[Tuple[torch.Tensor, ...]], bn_eps: float,
                         use_absolute_xyz: bool, with_distance: bool, voxel_size: Sequence[float], range_min: Sequence[float],
                         n_agents: int, ny: int, nx: int, canvas_cache: dict, unique_cells: bool = True,
                         want_features: bool = False) -> Tuple[Optional[torch.Tensor], torch.Tensor]:
    """The persistent-canvas encoder behind a producer that keeps its pillar count on the device (``ops.voxelize``): the arrays are
    capacity sized, ``count_dev`` (int32 device tensor, first element) says how many rows are valid; nothing here reads it on the host, so
    voxeliser -> encoder -> backbone is one stream of launches (one captured graph).  ``canvas_cache``: as in ``pillar_vfe_scatter`` (one
    persistent channels-last canvas per device, stream and shape).  -> (pillar_features [capacity, C] or None, canvas [n_agents, C, ny, nx])."""
    _need_gpu(voxel_features, voxel_num_points, voxel_coords, count_dev, weight)
    L = hip.lib()
    vf = _f32c(voxel_features)
    cap, P = vf.shape[0], vf.shape[1]
    if vf.dim() != 3 or vf.shape[2] != 4:
        raise ValueError(f"voxel_features must be [M, P, 4], got {tuple(vf.shape)}")
    if count_dev.dtype != torch.int32:
        raise ValueError("count_dev must be an int32 device tensor")
    npts = voxel_num_points.to(torch.int32).contiguous()
    coords = voxel_coords.to(torch.int32).contiguous()
    w = _f32c(weight)
    C = w.shape[0]
    dev = vf.device
    if P > 32 or C > 64 or C % 4:
        raise hip.CoalignHipError("pillar_encode_stream: the channels-last encoder needs P <= 32 and C <= 64 (a multiple of 4)")
    # (the capacity is part of the key: the C entry point clears "the rows the previous call wrote" out of a slot list of THIS capacity -- a smaller
    #  capacity on the same canvas would leave the previous frame's rows beyond it uncleared)
    key = ("stream", str(dev), torch.cuda.current_stream(dev).cuda_stream, n_agents, C, ny, nx, bool(unique_cells), int(cap))
    entry = canvas_cache.get(key)
    if entry is None:
        entry = canvas_cache[key] = {
            "canvas": torch.empty((n_agents, C, ny, nx), dtype=torch.float32, device=dev, memory_format=torch.channels_last).zero_(),
            "cellmap": None if unique_cells else torch.full((n_agents * ny * nx,), -1, dtype=torch.int32, device=dev),
            "dest": torch.zeros(cap + 1, dtype=torch.int32, device=dev)}
    canvas = entry["canvas"]
    feats = torch.empty((cap, C), dtype=torch.float32, device=dev) if want_features else None
    bnp = [None] * 4 if bn is None else [_f32c(t) for t in bn]
    b = None if bias is None else _f32c(bias)
    with _Timed("pillar_encode_stream"):
        hip.check(L.coalign_pillar_encode_stream(_ptr(vf), _ptr(npts), _ptr(coords), cap, _ptr(count_dev), P, _ptr(w), _ptr(b), _ptr(bnp[0]), _ptr(bnp[1]),
                                                 _ptr(bnp[2]), _ptr(bnp[3]), float(bn_eps), C, int(use_absolute_xyz), int(with_distance), _dbl3(voxel_size),
                                                 _dbl3(range_min), n_agents, ny, nx, _ptr(feats), _ptr(entry["dest"]), _ptr(canvas), _ptr(entry["cellmap"]),
                                                 int(bool(unique_cells)), _stream()), "coalign_pillar_encode_stream")
    return feats, canvas


_SP_RANGE_FLAGS: dict = {}


def sp_range_flag(device, create: bool = True) -> Optional[torch.Tensor]:
    """The device word every SplitMap-producing kernel ORs bit 0 into when a value left the pair's range (|x| > 65504; include/coalign_amd.h (9e)).  One per
    device, never cleared by the kernels.  Created on first use outside a graph capture (``FramePipeline`` creates it before capturing); ``None`` if it does not
    exist yet and ``create`` is false or a capture is running (the kernels then skip the report)."""
    dev = torch.device(device)
    key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device())
    f = _SP_RANGE_FLAGS.get(key)
    if f is None and create and not torch.cuda.is_current_stream_capturing():
        f = _SP_RANGE_FLAGS[key] = torch.zeros(1, dtype=torch.int32, device=dev)
    return f


def sp_range_exceeded(device, clear: bool = True) -> bool:
    """True if a SplitMap value exceeded the fp16 split's operating range since the last clearing call (synchronises the device)."""
    f = sp_range_flag(device, create=False)
    if f is None:
        return False
    hit = bool(int(f.item()) & 1)
    if hit and clear:
        f.zero_()
    return hit


class SplitMap:
    """An activation map stored as sp16 pairs in the matrix instructions' operand order (csrc/conv3x3_sp.hip, include/coalign_amd.h (9e)): what the 3x3
    convolutions of the fp16 mode hand to each other inside a ResNet stage.  ``shape`` is the logical [N, C, H, W]; ``data`` the float16 tensor
    [N, C / 16, 4, H, W, 8] (plane = 2 * channel half + term); ``dense()`` returns the float32 tensor the pairs stand for (the values rounded to 22 bits)."""

    def __init__(self, data: torch.Tensor):
        if data.dtype != torch.float16 or data.dim() != 6 or data.shape[2] != 4 or data.shape[5] != 8 or not data.is_contiguous():
            raise ValueError("SplitMap data: contiguous float16 [N, C / 16, 4, H, W, 8]")
        self.data = data
        self.shape = (data.shape[0], data.shape[1] * 16, data.shape[3], data.shape[4])
        self.device, self.dtype, self.is_cuda = data.device, torch.float32, data.is_cuda

    @staticmethod
    def empty(N: int, C: int, H: int, W: int, device) -> "SplitMap":
        if C % 16:
            raise ValueError("SplitMap: C % 16 == 0")
        return SplitMap(torch.empty((N, C // 16, 4, H, W, 8), dtype=torch.float16, device=device))

    @staticmethod
    def pack(x: torch.Tensor) -> "SplitMap":
        """float32 [N, C, H, W] (NCHW or channels-last memory) -> SplitMap (``coalign_sp_pack``)."""
        _need_gpu(x)
        nhwc = x.dtype == torch.float32 and is_channels_last(x)
        xc = x if nhwc else _f32c(x)
        N, C, H, W = xc.shape
        out = SplitMap.empty(N, C, H, W, xc.device)
        hip.check(hip.lib().coalign_sp_pack(_ptr(xc), int(nhwc), _ptr(out.data), N, C, H, W, _ptr(sp_range_flag(xc.device)), _stream()), "coalign_sp_pack")
        return out

    def dense(self, channels_last: bool = False) -> torch.Tensor:
        N, C, H, W = self.shape
        y = torch.empty((N, C, H, W), dtype=torch.float32, device=self.device, memory_format=torch.channels_last if channels_last else torch.contiguous_format)
        if channels_last and not nhwc_memory(y):
            y = torch.empty((N, H, W, C), dtype=torch.float32, device=self.device).permute(0, 3, 1, 2)
        hip.check(hip.lib().coalign_sp_unpack(_ptr(self.data), _ptr(y), int(channels_last), N, C, H, W, _stream()), "coalign_sp_unpack")
        return y

    def dense_reference(self) -> torch.Tensor:
        """The same values computed with torch ops from ``data`` (tests: the layout's definition, independent of the unpack kernel)."""
        d = self.data.float()
        v = d[:, :, 0::2] + d[:, :, 1::2] / 1024.0                    # [N, C/16, 2 halves, H, W, 8]
        N, C, H, W = self.shape
        return v.permute(0, 1, 2, 5, 3, 4).reshape(N, C, H, W)


class SparseCanvas:
    """The BEV canvas of one batch as the sparse pair (feature rows, cell stamps) of csrc/pillar_sparse.hip instead of a dense tensor: what
    ``PillarVFE`` + ``PointPillarScatter`` hand to the first ResNet stage on the fast path.  ``shape`` is the dense tensor's; ``dense()`` materialises it
    (the reference's ``spatial_features``) for anyone who needs the tensor."""

    def __init__(self, feats, stamps, state, coords, n_agents, C, ny, nx, count_dev=None, owner: Optional[dict] = None):
        self.feats, self.stamps, self.state, self.coords = feats, stamps, state, coords
        # every canvas of one (device, stream, grid) shares ONE stamp map, and only the LATEST encode through it is in the stamps: `owner` is that map's cache
        # entry, `generation` the encode this object came from.  The consumers refuse a canvas that a later encode has overwritten (check_current) instead of
        # reading the newer frame's rows out of this object's `feats`; on the device a stamp naming a row beyond `feats` reads as empty (M_rows, (9d) / (10b)).
        self.owner, self.generation = owner, None if owner is None else owner.get("generation")
        if feats.numel() == 0 and feats.untyped_storage().nbytes() == 0:      # the consumers take a non-NULL row pointer even when every stamp is stale
            self.feats = torch.empty((1, C), dtype=torch.float32, device=feats.device)[:0]
        self.n_agents, self.C, self.ny, self.nx, self.count_dev = n_agents, C, ny, nx, count_dev
        self.shape = (n_agents, C, ny, nx)
        self.device, self.dtype = feats.device, feats.dtype
        self.is_cuda = True

    def check_current(self) -> None:
        if self.owner is not None and self.owner.get("generation") != self.generation:
            raise hip.CoalignHipError("stale SparseCanvas: a later pillar_encode_sparse on the same device, stream and grid has overwritten the shared stamp map "
                                      "(consume a canvas -- or .dense() it -- before the next encode, or encode on another stream / with another canvas_cache)")

    def dense(self) -> torch.Tensor:
        if self.coords is None:
            raise hip.CoalignHipError("a SparseCanvas encoded through a frame record has no coordinate array of its own: densify from the frame's voxel_coords")
        if self.count_dev is not None:
            m = int(self.count_dev[0].item())
            return scatter_to_bev(self.feats[:m], self.coords[:m], self.n_agents, self.ny, self.nx)
        return scatter_to_bev(self.feats, self.coords, self.n_agents, self.ny, self.nx)


# The frame tag of a sparse canvas is 32 bits and stamps are ordered by (tag, row) through atomicMax: a tag that wrapped would lose against the stale stamps of
# the frames before it.  A stamp map is therefore re-zeroed (stamps + frame words, on the launching stream) after this many launches -- 2^31 frames are 50 days
# of one lane at 500 frames/s; callers that replay a captured launch count the replays themselves (FramePipeline does) and call reset_sparse_canvases.
SPARSE_TAG_RESET_AFTER = 1 << 31


def reset_sparse_canvases(canvas_cache: dict) -> int:
    """Zero the stamps and frame words of every sparse canvas of ``canvas_cache`` on the current stream; -> number of canvases reset."""
    n = 0
    for key, entry in canvas_cache.items():
        if isinstance(key, tuple) and key and key[0] == "sparse":
            entry["stamps"].zero_()
            entry["state"].zero_()
            entry["calls"] = 0
            entry["generation"] = entry.get("generation", 0) + 1      # canvases encoded before the reset are gone
            n += 1
    return n


@_device_op
def pillar_fold_params(weight: torch.Tensor, bias: Optional[torch.Tensor], bn: Optional[Tuple[torch.Tensor, ...]], bn_eps: float, use_absolute_xyz: bool) -> torch.Tensor:
    """The encoder's channel parameters in the form ``pillar_encode_sparse`` uses (include/coalign_amd.h (1b)): once per weight set, like the split
    weight images of the convolutions."""
    _need_gpu(weight)
    L = hip.lib()
    w = _f32c(weight)
    out = torch.empty(L.coalign_pillar_folded_param_bytes() // 4, dtype=torch.float32, device=w.device)
    bnp = [None] * 4 if bn is None else [_f32c(t) for t in bn]
    b = None if bias is None else _f32c(bias)
    hip.check(L.coalign_pillar_fold_params(_ptr(w), _ptr(b), _ptr(bnp[0]), _ptr(bnp[1]), _ptr(bnp[2]), _ptr(bnp[3]), float(bn_eps), w.shape[0], int(use_absolute_xyz),
                                           _ptr(out), _stream()), "coalign_pillar_fold_params")
    return out


class FrameRecordUnsupported(hip.CoalignHipError):
    """The configured route cannot read its inputs through a ``PillarFrameRecord``; the caller copies the frame into the graph's buffers instead."""


class PillarFrameRecord:
    """``coalign_pillar_frame`` (include/coalign_amd.h (1c)) for a replayed graph: ``words`` is a 4-word int64 DEVICE view (array pointers | pillar count) the
    pillar launch reads when it starts, ``host`` the pinned words the owner fills per frame (``set``) and copies over as part of ONE small host-to-device
    transfer.  ``capacity`` sizes the launch; a frame may hold fewer pillars, never more."""

    def __init__(self, words: torch.Tensor, host: torch.Tensor, capacity: int):
        if words.dtype != torch.int64 or words.numel() != 4 or not words.is_cuda or host.dtype != torch.int64 or host.numel() != 4:
            raise ValueError("a frame record is four int64 words on the device and four on the (pinned) host")
        self.words, self.host, self.capacity = words, host, int(capacity)
        self._host_np = host.numpy()

    @staticmethod
    def admits(vf: torch.Tensor, npts: torch.Tensor, coords: torch.Tensor, device) -> bool:
        """The arrays are read IN PLACE: float32 / int32, contiguous, on the launch's device (anything else takes the copying route)."""
        return (vf.is_cuda and vf.device == device and npts.device == device and coords.device == device and vf.dtype == torch.float32 and vf.dim() == 3
                and vf.shape[2] == 4 and npts.dtype == torch.int32 and coords.dtype == torch.int32 and coords.dim() == 2 and coords.shape[1] == 4
                and vf.is_contiguous() and npts.is_contiguous() and coords.is_contiguous() and npts.shape[0] == vf.shape[0] == coords.shape[0])

    def set(self, vf: torch.Tensor, npts: torch.Tensor, coords: torch.Tensor) -> None:
        M = int(vf.shape[0])
        if M > self.capacity:
            raise ValueError(f"{M} pillars in a frame record of capacity {self.capacity}")
        self._host_np[:] = (_rows_ptr(vf).value or 0, _rows_ptr(npts).value or 0, _rows_ptr(coords).value or 0, M)      # (little endian: the count is the low half of word 3)


@_device_op
def pillar_encode_sparse(voxel_features: torch.Tensor, voxel_num_points: torch.Tensor, voxel_coords: torch.Tensor, weight: torch.Tensor,
                         bias: Optional[torch.Tensor], bn: Optional[Tuple[torch.Tensor, ...]], bn_eps: float, use_absolute_xyz: bool,
                         voxel_size: Sequence[float], range_min: Sequence[float], n_agents: int, ny: int, nx: int, canvas_cache: dict,
                         count_dev: Optional[torch.Tensor] = None, folded: Optional[torch.Tensor] = None, frame: Optional[PillarFrameRecord] = None) -> SparseCanvas:
    """PillarVFE + PointPillarScatter in ONE launch (include/coalign_amd.h (1b)): feature rows [M, C] + 8-byte cell stamps.  ``canvas_cache`` keeps the
    stamp map and the frame-tag words of this (device, stream, grid): they persist across frames and are never cleared (a stamp is valid only with
    the current tag).  ``count_dev``: optional int32 device tensor holding the pillar count (capacity-sized arrays).  ``folded``: ``pillar_fold_params``
    of the same weights; a caller that keeps none gets them folded here (one more small launch per call).  ``frame``: the launch reads its three arrays and the
    count through this device record (include/coalign_amd.h (1c)); the tensors passed here then only give the shapes."""
    _need_gpu(voxel_features, voxel_num_points, voxel_coords, weight)
    L = hip.lib()
    vf = _f32c(voxel_features)
    if vf.dim() != 3 or vf.shape[2] != 4:
        raise ValueError(f"voxel_features must be [M, P, 4], got {tuple(vf.shape)}")
    M, P = vf.shape[0], vf.shape[1]
    if frame is not None:
        if count_dev is not None:
            raise ValueError("a frame record carries the pillar count itself")
        M, npts, coords = frame.capacity, None, None
    else:
        npts = voxel_num_points.to(torch.int32).contiguous()
        coords = voxel_coords.to(torch.int32).contiguous()
    C = weight.shape[0]
    dev = vf.device
    if P > 32 or C > 64:
        raise hip.CoalignHipError("pillar_encode_sparse: P <= 32 and C <= 64")
    if folded is None:
        folded = pillar_fold_params(weight, bias, bn, bn_eps, use_absolute_xyz)
    if folded.dtype != torch.float32 or folded.numel() * 4 != L.coalign_pillar_folded_param_bytes() or folded.device != dev:
        raise ValueError("folded: the tensor pillar_fold_params returned for these weights")
    key = ("sparse", str(dev), torch.cuda.current_stream(dev).cuda_stream, n_agents, ny, nx)
    entry = canvas_cache.get(key)
    if entry is None:
        entry = canvas_cache[key] = {"stamps": torch.zeros(n_agents * ny * nx, dtype=torch.int64, device=dev), "state": torch.zeros(L.coalign_sparse_canvas_state_bytes() // 4, dtype=torch.int32, device=dev)}
    entry["calls"] = entry.get("calls", 0) + 1
    entry["generation"] = entry.get("generation", 0) + 1      # (SparseCanvas.check_current)
    if entry["calls"] >= SPARSE_TAG_RESET_AFTER:             # (never inside a captured frame: a capture happens in a map's first calls)
        entry["stamps"].zero_()
        entry["state"].zero_()
        entry["calls"] = 1
    feats = torch.empty((max(M, 1), C), dtype=torch.float32, device=dev)[:M]      # (an empty frame still hands its consumers a valid row pointer: ADVICE r04)
    if frame is not None:
        with _Timed("pillar_encode_sparse"):
            hip.check(L.coalign_pillar_encode_sparse_frame(_ptr(frame.words), M, P, _ptr(folded), C, int(use_absolute_xyz), _dbl3(voxel_size), _dbl3(range_min), n_agents,
                                                           ny, nx, _ptr(feats), _ptr(entry["stamps"]), _ptr(entry["state"]), _stream()), "coalign_pillar_encode_sparse_frame")
        sc = SparseCanvas(feats, entry["stamps"], entry["state"], None, n_agents, C, ny, nx, None, owner=entry)
        sc.count_word = frame.words[3:4].view(torch.int32)[0:1]      # the record's pillar count (little endian: the low half of word 3), for sp_pack_rows
        return sc
    with _Timed("pillar_encode_sparse"):
        hip.check(L.coalign_pillar_encode_sparse(_ptr(vf), _ptr(npts), _ptr(coords), M, _ptr(count_dev), P, _ptr(folded), C, int(use_absolute_xyz), _dbl3(voxel_size),
                                                 _dbl3(range_min), n_agents, ny, nx, _ptr(feats), _ptr(entry["stamps"]), _ptr(entry["state"]), _stream()),
                  "coalign_pillar_encode_sparse")
    return SparseCanvas(feats, entry["stamps"], entry["state"], coords, n_agents, C, ny, nx, count_dev, owner=entry)


@_device_op
def conv3x3_emu_sparse(sc: SparseCanvas, w_split: torch.Tensor, bias: torch.Tensor, cout: int, relu: bool, terms: int, out_channels_last: bool, out_split: bool = False):
    """The strided first convolution of the backbone reading a SparseCanvas (include/coalign_amd.h (9d)); tap-pair weight image.  ``out_split`` (terms 16):
    the result is a ``SplitMap`` (the input of ``conv3x3_sp``)."""
    L = hip.lib()
    sc.check_current()
    N, Cin, H, W = sc.shape
    Ho, Wo = (H + 1) // 2, (W + 1) // 2
    if w_split.numel() != L.coalign_conv3x3_emu_weight_bytes(Cin, cout, terms):
        raise ValueError("split weight image does not match (Cin, Cout, terms)")
    if out_split:
        if terms != 16:
            raise ValueError("SplitMap outputs belong to the fp16 split (terms = 16)")
        out = SplitMap.empty(N, cout, Ho, Wo, sc.device)
        y = out.data
    else:
        out = y = torch.empty((N, cout, Ho, Wo), dtype=torch.float32, device=sc.device, memory_format=torch.channels_last if out_channels_last else torch.contiguous_format)
    with _Timed("conv3x3_emu_sparse"):
        hip.check(L.coalign_conv3x3_emu_sparse(_rows_ptr(sc.feats), int(sc.feats.shape[0]), _ptr(sc.stamps), _ptr(sc.state), _ptr(w_split), _ptr(_f32c(bias)), _ptr(y), N, Cin, cout, H, W,
                                               int(relu), terms, 2 if out_split else int(out_channels_last), _ptr(sp_range_flag(sc.device)) if out_split else None, _stream()),
                  "coalign_conv3x3_emu_sparse")
    return out


@_device_op
def pointwise_conv_sparse(sc: SparseCanvas, w_emu: torch.Tensor, bias: torch.Tensor, cout: int, relu: bool, out_channels_last: bool) -> torch.Tensor:
    """The 1 x 1 / stride-2 skip convolution reading a SparseCanvas (include/coalign_amd.h (10b)); ``w_emu``: ``pack_pointwise_emu_weight`` image."""
    L = hip.lib()
    sc.check_current()
    N, Cin, H, W = sc.shape
    if w_emu.dtype != torch.int16 or w_emu.dim() != 5 or w_emu.shape[1] * 16 != Cin:
        raise ValueError("pointwise_conv_sparse needs the split-bf16 weight image of this Cin")
    Ho, Wo = (H + 1) // 2, (W + 1) // 2
    y = torch.empty((N, cout, Ho, Wo), dtype=torch.float32, device=sc.device, memory_format=torch.channels_last if out_channels_last else torch.contiguous_format)
    with _Timed("pointwise_conv_sparse"):
        hip.check(L.coalign_pointwise_conv_emu_sparse(_rows_ptr(sc.feats), int(sc.feats.shape[0]), _ptr(sc.stamps), _ptr(sc.state), _ptr(w_emu), _ptr(_f32c(bias)), _ptr(y), N, Cin, H, W, cout,
                                                      w_emu.shape[0] * 32, int(relu), int(out_channels_last), _stream()), "coalign_pointwise_conv_emu_sparse")
    return y


@_device_op
def scatter_to_bev(pillar_features: torch.Tensor, voxel_coords: torch.Tensor, n_agents: int, ny: int, nx: int) -> torch.Tensor:
    """pillar_features [M, C] + coords (agent, z, y, x) -> canvas [n_agents, C, ny, nx]."""
    _need_gpu(pillar_features, voxel_coords)
    L = hip.lib()
    pf = _f32c(pillar_features)
    coords = voxel_coords.to(torch.int32).contiguous()
    M, C = pf.shape
    canvas = torch.empty((n_agents, C, ny, nx), dtype=torch.float32, device=pf.device)
    ws_bytes = L.coalign_pillar_scatter_workspace_bytes(n_agents, ny, nx)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=pf.device)
    hip.check(L.coalign_scatter_to_bev(_ptr(pf), _ptr(coords), M, C, n_agents, ny, nx, _ptr(canvas), _ptr(ws), ws_bytes, _stream()),
              "coalign_scatter_to_bev")
    return canvas


@_device_op
def warp_fuse(x: torch.Tensor, theta: torch.Tensor, group_len: Sequence[int], mode: int,
              out_hw: Optional[Tuple[int, int]] = None, rows: Optional[Sequence[int]] = None) -> torch.Tensor:
    """x [n_total, C, H, W] f32, theta [n_total, 2, 3] f64 (device) -> fused [len(group_len), C, Ho, Wo]
    (or [n_total, C, Ho, Wo] for FUSE_NONE).  ``rows``: per frame, the row of x holding logical agent i (default identity)."""
    _need_gpu(x, theta)
    L = hip.lib()
    xc = _f32c(x)
    th = theta.to(device=xc.device, dtype=torch.float64).contiguous()
    n_total, C, H, W = xc.shape
    Ho, Wo = (H, W) if out_hw is None else (int(out_hw[0]), int(out_hw[1]))
    groups = [int(g) for g in group_len]
    n_out = n_total if mode == FUSE_NONE else len(groups)
    out = torch.empty((n_out, C, Ho, Wo), dtype=torch.float32, device=xc.device)
    gl = (ctypes.c_int32 * max(1, len(groups)))(*groups)
    rw = None
    if rows is not None:
        if len(rows) != n_total:
            raise ValueError("rows must have one entry per agent")
        rw = (ctypes.c_int32 * max(1, n_total))(*[int(r) for r in rows])
    with _Timed(f"warp_fuse_C{C}"):
        hip.check(L.coalign_warp_fuse_rows(_ptr(xc), n_total, C, H, W, _ptr(th), gl, len(groups), rw, mode, _ptr(out), Ho, Wo, _stream()),
                  "coalign_warp_fuse_rows")
    return out


@_device_op
def normalize_pairwise(pairwise_t_matrix: torch.Tensor, H: int, W: int, den_x: float, den_y: float) -> torch.Tensor:
    """[..., 4, 4] float64 (device) -> [..., 2, 3] float64: normalize_pairwise_tfm in one launch (include/coalign_amd.h)."""
    _need_gpu(pairwise_t_matrix)
    t = pairwise_t_matrix.contiguous()
    if t.dtype != torch.float64 or t.shape[-2:] != (4, 4):
        raise ValueError("pairwise_t_matrix must be float64 [..., 4, 4]")
    out = torch.empty(tuple(t.shape[:-2]) + (2, 3), dtype=torch.float64, device=t.device)
    n = t.numel() // 16
    hip.check(hip.lib().coalign_normalize_pairwise(_ptr(t), n, int(H), int(W), float(den_x), float(den_y), _ptr(out), _stream()),
              "coalign_normalize_pairwise")
    return out


def warp_fuse_nhwc_ok(x: torch.Tensor) -> bool:
    """Channels-last map the one-launch fusion kernel serves: [n <= 8, C in {64, 128, 256}, H, W] float32 in NHWC memory."""
    return x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[1] in (64, 128, 256) and x.shape[0] <= 8 and is_channels_last(x)


@_device_op
def warp_fuse_nhwc(xs: Sequence[torch.Tensor], theta: torch.Tensor, mode: int, rows: Optional[Sequence[int]] = None,
                   out_hw: Optional[Sequence[Tuple[int, int]]] = None) -> list:
    """Up to three channels-last feature scales of ONE frame, ``xs[i]`` [n, C_i, H_i, W_i] (NHWC memory), theta [n, 2, 3] f64 ->
    per scale the fused map [1, C_i, Ho_i, Wo_i] (ATT / MAX) or the warped maps [n, C_i, Ho_i, Wo_i] (NONE), channels-last."""
    _need_gpu(*xs, theta)
    L = hip.lib()
    k = len(xs)
    n = xs[0].shape[0]
    if not 1 <= k <= 3 or any(not warp_fuse_nhwc_ok(x) or x.shape[0] != n for x in xs):
        raise ValueError("warp_fuse_nhwc needs 1-3 channels-last float32 maps [n <= 8, C in {64, 128, 256}, H, W] of one frame")
    th = theta.to(device=xs[0].device, dtype=torch.float64).contiguous()
    if th.shape[0] != n:
        raise ValueError("one theta row per agent")
    hw = [tuple(x.shape[2:]) for x in xs] if out_hw is None else [(int(a), int(b)) for a, b in out_hw]
    outs = [torch.empty((n if mode == FUSE_NONE else 1, x.shape[1], h, w), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
            for x, (h, w) in zip(xs, hw)]
    vp = ctypes.c_void_p * k
    i32 = ctypes.c_int32 * k
    rw = None
    if rows is not None:
        if len(rows) != n:
            raise ValueError("rows must have one entry per agent")
        rw = (ctypes.c_int32 * n)(*[int(r) for r in rows])
    with _Timed("warp_fuse_nhwc"):
        hip.check(L.coalign_warp_fuse_nhwc(k, vp(*[x.data_ptr() for x in xs]), i32(*[x.shape[1] for x in xs]), i32(*[x.shape[2] for x in xs]),
                                           i32(*[x.shape[3] for x in xs]), vp(*[o.data_ptr() for o in outs]), i32(*[h for h, _ in hw]),
                                           i32(*[w for _, w in hw]), n, _ptr(th), rw, mode, _stream()), "coalign_warp_fuse_nhwc")
    return outs


class DecodeBuffers:
    """Caller-owned candidate buffers + workspaces of the post-processing kernels (reused across frames)."""

    def __init__(self, capacity: int, A: int, H: int, W: int, top: int, device):
        L = hip.lib()
        self.capacity, self.top = capacity, top
        self.frame_words = torch.zeros(67, dtype=torch.int32, device=device)  # cleared once per frame by ONE launch (reset_frame); copied to the host as ONE block
        self.counts = self.frame_words[:64]                                    # chained per-agent totals
        self.status = self.frame_words[64:65]
        self.cand_index = torch.empty(capacity, dtype=torch.int32, device=device)
        self.cand_score = torch.empty(capacity, dtype=torch.float32, device=device)
        self.cand_box7 = torch.empty((capacity, 7), dtype=torch.float32, device=device)
        self.cand_corners = torch.empty((capacity, 8, 3), dtype=torch.float32, device=device)
        self.cand_keep = torch.empty(capacity, dtype=torch.uint8, device=device)
        self.dec_ws_bytes = L.coalign_anchor_decode_workspace_bytes(A, H, W)
        self.dec_ws = torch.empty(max(1, self.dec_ws_bytes), dtype=torch.uint8, device=device)
        self.nms_ws_bytes = L.coalign_nms_rotated_workspace_bytes(capacity, top)
        self.nms_ws = torch.empty(max(1, self.nms_ws_bytes), dtype=torch.uint8, device=device)
        self.keep = torch.empty(top, dtype=torch.int32, device=device)
        self.keep_count = self.frame_words[65:66]
        self.out_flat = torch.empty(top * 25, dtype=torch.float32, device=device)      # corners | scores in ONE allocation: a finished frame is taken out with one copy launch
        self.out_corners = self.out_flat[: top * 24].view(top, 8, 3)
        self.out_scores = self.out_flat[top * 24:]
        self.out_count = self.frame_words[66:67]
        self.host = torch.zeros(67, dtype=torch.int32).pin_memory()  # the frame words of the last frame: chained totals | status | kept | final


    def reset_frame(self) -> None:
        """counts / status of a new frame: one hand-written fill launch on the current stream (no torch fill in the captured frame)."""
        with torch.cuda.device(self.frame_words.device):
            hip.check(hip.lib().coalign_fill_words(_ptr(self.frame_words), self.frame_words.numel(), 0, _stream()), "coalign_fill_words")


@_device_op
def anchor_decode(buf: DecodeBuffers, slot: int, cls: torch.Tensor, reg: torch.Tensor, dir_: Optional[torch.Tensor],
                  anchors_f32: torch.Tensor, score_thr: float, dir_offset: float, num_bins: int, order: str,
                  transform: Optional[torch.Tensor], clear_frame: bool = False) -> None:
    """Append one agent's candidates after slot ``slot`` of ``buf.counts`` (slot 0 must hold 0).  ``clear_frame`` (round 6, ``coalign_anchor_decode_first``): this
    is the frame's first decode call -- its first launch zeroes ``buf.frame_words`` (what ``reset_frame`` did with a launch of its own)."""
    _need_gpu(cls, reg, anchors_f32)
    L = hip.lib()
    cls_c, reg_c = _f32c(cls), _f32c(reg)
    dir_c = None if dir_ is None else _f32c(dir_)
    if cls_c.dim() == 4:
        if cls_c.shape[0] != 1:
            raise ValueError("post-processing handles one frame at a time (batch size 1), like the reference")
        cls_c, reg_c = cls_c[0], reg_c[0]
        dir_c = None if dir_c is None else dir_c[0]
    A, H, W = cls_c.shape
    T = None if transform is None else _f32c(transform)
    args = (_ptr(cls_c), _ptr(reg_c), _ptr(dir_c), _ptr(anchors_f32), A, H, W, int(num_bins),
            float(score_thr), float(dir_offset), int(order == "hwl"), _ptr(T), buf.capacity,
            ctypes.c_void_p(buf.counts.data_ptr() + 4 * slot),
            ctypes.c_void_p(buf.counts.data_ptr() + 4 * (slot + 1)),
            _ptr(buf.cand_index), _ptr(buf.cand_score), _ptr(buf.cand_box7), _ptr(buf.cand_corners),
            _ptr(buf.cand_keep), _ptr(buf.status), _ptr(buf.dec_ws), buf.dec_ws_bytes)
    if clear_frame and _os.environ.get("COALIGN_PP_FILL") == "1":      # (measurement switch: the separate fill launch of rounds 2-5)
        buf.reset_frame()
        clear_frame = False
    if clear_frame:
        if slot != 0:
            raise ValueError("the frame's words are cleared by its FIRST decode call (slot 0)")
        hip.check(L.coalign_anchor_decode_first(*args, _ptr(buf.frame_words), buf.frame_words.numel(), _stream()), "coalign_anchor_decode_first")
    else:
        hip.check(L.coalign_anchor_decode(*args, _stream()), "coalign_anchor_decode")


@_device_op
def nms_rotated_device(boxes: torch.Tensor, scores: torch.Tensor, iou_thr: float, top: int = 1000,
                       valid: Optional[torch.Tensor] = None, k_dev: Optional[torch.Tensor] = None,
                       keep: Optional[torch.Tensor] = None, keep_count: Optional[torch.Tensor] = None,
                       ws: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """boxes [K, 8, 3] or [K, 4, 2] -> (keep int32 [top] device, keep_count int32 [1] device); no host sync."""
    _need_gpu(boxes, scores)
    L = hip.lib()
    b = _f32c(boxes)
    s = _f32c(scores)
    K = b.shape[0]
    rows, cols = (b.shape[1], b.shape[2]) if b.dim() == 3 else (4, 2)
    dev = b.device
    keep = torch.empty(top, dtype=torch.int32, device=dev) if keep is None else keep
    keep_count = torch.zeros(1, dtype=torch.int32, device=dev) if keep_count is None else keep_count
    ws_bytes = L.coalign_nms_rotated_workspace_bytes(K, top)
    if ws is None:
        ws = torch.empty(max(1, ws_bytes), dtype=torch.uint8, device=dev)
    v = None if valid is None else valid.to(torch.uint8).contiguous()
    hip.check(L.coalign_nms_rotated(_ptr(b), rows, cols, _ptr(s), _ptr(v), K, _ptr(k_dev), float(iou_thr), top, _ptr(keep),
                                    _ptr(keep_count), _ptr(ws), ws.numel(), _stream()), "coalign_nms_rotated")
    return keep, keep_count


@_device_op
def nms_rotated_gather(corners: torch.Tensor, scores: torch.Tensor, iou_thr: float, top: int, valid: Optional[torch.Tensor], k_dev: Optional[torch.Tensor],
                       keep: torch.Tensor, keep_count: torch.Tensor, limit_range: Sequence[float], out_corners: torch.Tensor, out_scores: torch.Tensor,
                       out_count: torch.Tensor, ws: torch.Tensor) -> None:
    """``nms_rotated_device`` + ``gather_in_range`` as one C-ABI call (rank, bitmask, walk + gather: three launches); corners [K, 8, 3]."""
    _need_gpu(corners, scores, keep)
    L = hip.lib()
    b, s_ = _f32c(corners), _f32c(scores)
    if b.dim() != 3 or tuple(b.shape[1:]) != (8, 3):
        raise ValueError(f"corners must be [K, 8, 3], got {tuple(b.shape)}")
    v = None if valid is None else valid.to(torch.uint8).contiguous()
    r = (ctypes.c_double * 6)(*[float(x) for x in limit_range])
    hip.check(L.coalign_nms_rotated_gather(_ptr(b), _ptr(s_), _ptr(v), b.shape[0], _ptr(k_dev), float(iou_thr), int(top), _ptr(keep), _ptr(keep_count), r,
                                           _ptr(out_corners), _ptr(out_scores), _ptr(out_count), _ptr(ws), ws.numel(), _stream()), "coalign_nms_rotated_gather")


@_device_op
def gather_in_range(corners: torch.Tensor, scores: torch.Tensor, keep: torch.Tensor, keep_count: torch.Tensor,
                    limit_range: Sequence[float], out_corners: torch.Tensor, out_scores: torch.Tensor,
                    out_count: torch.Tensor) -> None:
    _need_gpu(corners, scores, keep)
    L = hip.lib()
    r = (ctypes.c_double * 6)(*[float(v) for v in limit_range])
    hip.check(L.coalign_gather_in_range(_ptr(corners), _ptr(scores), _ptr(keep), _ptr(keep_count), keep.numel(), r,
                                        _ptr(out_corners), _ptr(out_scores), _ptr(out_count), _stream()), "coalign_gather_in_range")


@_device_op
def iou_rotated_matrix(boxes_a: torch.Tensor, boxes_b: torch.Tensor) -> torch.Tensor:
    """[Na, 8, 3] / [Na, 4, 2] corners x [Nb, ...] corners -> float32 IoU matrix [Na, Nb] (float64 clipping inside)."""
    _need_gpu(boxes_a, boxes_b)
    L = hip.lib()
    a, b = _f32c(boxes_a), _f32c(boxes_b)
    out = torch.zeros((a.shape[0], b.shape[0]), dtype=torch.float32, device=a.device)
    if a.shape[0] and b.shape[0]:
        hip.check(L.coalign_iou_rotated_matrix(_ptr(a), a.shape[1], a.shape[2], a.shape[0], _ptr(b), b.shape[1], b.shape[2], b.shape[0],
                                               _ptr(out), _stream()), "coalign_iou_rotated_matrix")
    return out


@_device_op
def boxes_iou_bev(boxes_a: torch.Tensor, boxes_b: torch.Tensor) -> torch.Tensor:
    """OpenPCDet-semantics fp32 BEV IoU matrix [Na, Nb] of (x, y, z, dx, dy, dz, heading) boxes."""
    _need_gpu(boxes_a, boxes_b)
    L = hip.lib()
    a, b = _f32c(boxes_a), _f32c(boxes_b)
    out = torch.zeros((a.shape[0], b.shape[0]), dtype=torch.float32, device=a.device)
    hip.check(L.coalign_boxes_iou_bev(_ptr(a), a.shape[0], _ptr(b), b.shape[0], _ptr(out), _stream()), "coalign_boxes_iou_bev")
    return out


@_device_op
def bias_act_(y: torch.Tensor, bias: Optional[torch.Tensor], residual: Optional[torch.Tensor] = None, relu: bool = True) -> torch.Tensor:
    """In place ``y = act(y + bias[c] (+ residual))`` on a contiguous NCHW float32 tensor (fused conv epilogue)."""
    _need_gpu(y, bias, residual)
    if y.dtype != torch.float32 or not y.is_contiguous():
        raise ValueError("bias_act_ needs a contiguous float32 NCHW tensor")
    if residual is not None and (residual.shape != y.shape or not residual.is_contiguous() or residual.dtype != torch.float32):
        raise ValueError("residual must match y (contiguous float32)")
    N, C = y.shape[0], y.shape[1]
    HW = y.numel() // max(1, N * C)
    hip.check(hip.lib().coalign_bias_act(_ptr(y), _ptr(bias), _ptr(residual), N, C, HW, int(relu), _stream()), "coalign_bias_act")
    return y


VOX_FILTER_EGO, VOX_FILTER_RANGE = 1, 2   # COALIGN_VOX_FILTER_* of include/coalign_amd.h


@_device_op
def voxelize(points: torch.Tensor, cloud_offsets: Sequence[int], voxel_size: Sequence[float], lidar_range: Sequence[float],
             max_points: int, max_voxels: int, ego_filter: bool = False, filter_range: Optional[Sequence[float]] = None):
    """points [N, 4] (the clouds of a batch concatenated; ``cloud_offsets`` = host ints [n_clouds + 1]) ->
    (voxels [cap, max_points, 4], coords [cap, 4] int32 (cloud, z, y, x), num_points [cap] int32, counts [n_clouds + 1] int32).
    Only the first ``counts[-1]`` rows are meaningful; nothing here synchronises -- the caller slices after reading counts."""
    _need_gpu(points)
    L = hip.lib()
    pts = _f32c(points)
    if pts.dim() != 2 or pts.shape[1] != 4:
        raise ValueError(f"points must be [N, 4], got {tuple(pts.shape)}")
    n_clouds = len(cloud_offsets) - 1
    if int(cloud_offsets[-1]) != pts.shape[0]:
        raise ValueError("cloud_offsets[-1] must equal the number of points")
    off = (ctypes.c_int64 * (n_clouds + 1))(*[int(o) for o in cloud_offsets])
    vs, rg = _dbl3(voxel_size), (ctypes.c_double * 6)(*[float(v) for v in lidar_range])
    fr = None if filter_range is None else (ctypes.c_double * 6)(*[float(v) for v in filter_range])
    flags = (VOX_FILTER_EGO if ego_filter else 0) | (VOX_FILTER_RANGE if filter_range is not None else 0)
    cap = L.coalign_voxelize_capacity(pts.shape[0], n_clouds, vs, rg, int(max_voxels))
    if cap < 0:
        raise ValueError("invalid voxel grid / max_voxels")
    dev = pts.device
    voxels = torch.empty((cap, max_points, 4), dtype=torch.float32, device=dev)
    coords = torch.empty((cap, 4), dtype=torch.int32, device=dev)
    num = torch.empty((cap,), dtype=torch.int32, device=dev)
    counts = torch.empty((n_clouds + 1,), dtype=torch.int32, device=dev)
    ws_bytes = L.coalign_voxelize_workspace_bytes(off, n_clouds, vs, rg, int(max_voxels))
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=dev)
    with _Timed("voxelize"):
        hip.check(L.coalign_voxelize(_ptr(pts), off, n_clouds, vs, rg, int(max_points), int(max_voxels), flags, fr, _ptr(voxels),
                                     _ptr(coords), _ptr(num), cap, _ptr(counts), _ptr(ws), ws_bytes, _stream()), "coalign_voxelize")
    return voxels, coords, num, counts


@_device_op
def pose_graph_optimize(vertex_offsets: torch.Tensor, edge_offsets: torch.Tensor, n_agents: torch.Tensor, vertices: torch.Tensor,
                        kinds: torch.Tensor, edge_agent: torch.Tensor, edge_landmark: torch.Tensor, edge_meas: torch.Tensor,
                        edge_info: torch.Tensor, max_iterations: int = 1000) -> Tuple[torch.Tensor, torch.Tensor]:
    """Batched Levenberg-Marquardt over agent-object pose graphs (see include/coalign_amd.h (8)); all tensors on the device,
    offsets / indices int32, values float64.  -> (optimised vertices [V, 3], stats [G, 4])."""
    _need_gpu(vertex_offsets, edge_offsets, n_agents, vertices, kinds, edge_agent, edge_landmark, edge_meas, edge_info)
    L = hip.lib()
    i32 = lambda t: t.to(torch.int32).contiguous()
    f64 = lambda t: t.to(torch.float64).contiguous()
    vo, eo, na, kd, ea, el = (i32(t) for t in (vertex_offsets, edge_offsets, n_agents, kinds, edge_agent, edge_landmark))
    out, em, ew = f64(vertices).clone(), f64(edge_meas), f64(edge_info)
    G, V = na.shape[0], out.shape[0]
    stats = torch.zeros((G, 4), dtype=torch.float64, device=out.device)
    ws_bytes = L.coalign_pose_graph_workspace_bytes(V)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=out.device)
    with _Timed("pose_graph_optimize"):
        hip.check(L.coalign_pose_graph_optimize(G, _ptr(vo), _ptr(eo), _ptr(na), V, _ptr(out), _ptr(kd), _ptr(ea), _ptr(el), _ptr(em), _ptr(ew),
                                                int(max_iterations), _ptr(stats), _ptr(ws), ws_bytes, _stream()), "coalign_pose_graph_optimize")
    return out, stats


CONV_KC, CONV_WSTRIDE = 8, 9 * 64 + 32      # kKC / kWStride of csrc/conv3x3.hip
_CONV_WS: dict = {}
_CONV_WS_RETIRED: list = []      # outgrown workspaces stay allocated: HIP graphs captured earlier on the lane still hold their raw pointers.  Bounded: a
                                 # workspace only grows when a LARGER shape arrives on its stream, i.e. at most once per distinct layer shape and stream.


def _conv_workspace(key, ws_bytes: int, device) -> torch.Tensor:
    """One stream-K workspace per (device, stream[, kernel family]): launches on a stream are ordered.  A workspace that has to grow is
    replaced, never freed (a captured graph replays with the pointer it was captured with; flags and partials of different launches on one
    stream never overlap in time, so the retired buffer stays private to those graphs)."""
    ws = _CONV_WS.get(key)
    if ws is None or ws.numel() < ws_bytes:
        if ws is not None:
            _CONV_WS_RETIRED.append(ws)
        ws = _CONV_WS[key] = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=device)
    return ws


def pack_conv3x3_weight(weight: torch.Tensor) -> torch.Tensor:
    """[Cout, Cin, 3, 3] -> [Cout / 64, Cin / 8, 8, 608]: per 64-channel group and 8-input-channel chunk the exact LDS image
    coalign_conv3x3_bias_act streams in ([ci][tap][64 couts] + 32 zero floats of bank padding per input channel)."""
    co, ci, kh, kw = weight.shape
    if (kh, kw) != (3, 3) or co % 64 or ci % CONV_KC:
        raise ValueError(f"conv3x3 kernel needs 3x3 weights with Cout % 64 == 0 and Cin % 8 == 0, got {tuple(weight.shape)}")
    w = weight.detach().float().reshape(co // 64, 64, ci // CONV_KC, CONV_KC, 9).permute(0, 2, 3, 4, 1)   # [g, chunk, k, tap, 64]
    out = torch.zeros((co // 64, ci // CONV_KC, CONV_KC, CONV_WSTRIDE), dtype=torch.float32, device=weight.device)
    out[..., : 9 * 64] = w.reshape(co // 64, ci // CONV_KC, CONV_KC, 9 * 64)
    return out.contiguous()


@_device_op
def conv3x3_bias_act(x: torch.Tensor, w_packed: torch.Tensor, bias: Optional[torch.Tensor], residual: Optional[torch.Tensor] = None,
                     relu: bool = True) -> torch.Tensor:
    """y = act(conv3x3(x, w) + bias (+ residual)), stride 1, padding 1, NCHW float32, on the fp32 matrix cores."""
    _need_gpu(x, w_packed, bias, residual)
    L = hip.lib()
    xc = _f32c(x)
    N, Cin, H, W = xc.shape
    G, n_chunks, _, _ = w_packed.shape
    Cin_w = n_chunks * CONV_KC
    if Cin_w != Cin:
        raise ValueError(f"input has {Cin} channels, packed weight expects {Cin_w}")
    Cout = G * 64
    y = torch.empty((N, Cout, H, W), dtype=torch.float32, device=xc.device)
    res = None if residual is None else _f32c(residual)
    if res is not None and res.shape != y.shape:
        raise ValueError("residual shape mismatch")
    ws_bytes = L.coalign_conv3x3_workspace_bytes(N, Cin, Cout, H, W)
    key = (xc.device, torch.cuda.current_stream(xc.device).cuda_stream)
    ws = _conv_workspace(key, ws_bytes, xc.device)
    b = torch.zeros(Cout, dtype=torch.float32, device=xc.device) if bias is None else _f32c(bias)
    with _Timed("conv3x3_bias_act"):
        hip.check(L.coalign_conv3x3_bias_act(_ptr(xc), _ptr(w_packed), _ptr(b), _ptr(res), _ptr(y),
                                             N, Cin, Cout, H, W, int(relu), _ptr(ws), ws.numel(), _stream()), "coalign_conv3x3_bias_act")
    return y


def pack_conv3x3_emu_weight(weight: torch.Tensor, terms: int = 3, tap_major: bool = False) -> torch.Tensor:
    """[Cout, Cin, 3, 3] fp32 -> the split-bf16 LDS image of csrc/conv3x3_emu.hip, as uint8:
    [Cout / 64][Cin / 8][5 steps][terms][2 k-groups][64 cout][8 cin] bf16 (tap = 2 * step + k-group, the tenth tap is zero;
    term 0 = bf16(w), term 1 = bf16(w - term 0), term 2 = bf16(w - term 0 - term 1)) followed by 16 zero bytes; ``terms`` = 16: two fp16 terms of the
    per-channel scaled weights (below) followed by 16 zero bytes, [Cout] float32 2^-k_c and [Cout] float32 2^k_c.
    ``tap_major`` (Cin % 16 == 0): [Cout / 64][Cin / 16][9 taps][terms][2 channel halves][64 cout][8 cin] -- one matrix
    instruction = the 16 channels of one tap, no zero tap (COALIGN_LAYOUT_W_TAPMAJOR)."""
    co, ci, kh, kw = weight.shape
    if (kh, kw) != (3, 3) or co % 64 or ci % (2 * CONV_KC if tap_major else CONV_KC) or terms not in (2, 3, 16):
        raise ValueError(f"conv3x3_emu needs 3x3 weights with Cout % 64 == 0, Cin % 8 == 0 (16 tap-major) and terms in (2, 3, 16), got {tuple(weight.shape)}, {terms}")
    if tap_major:
        w = weight.detach().float().reshape(co // 64, 64, ci // (2 * CONV_KC), 2, CONV_KC, 9)
        w = w.permute(0, 2, 5, 3, 1, 4)                                    # [g, interval, tap, channel half, cout, cin]
    else:
        w = weight.detach().float().reshape(co // 64, 64, ci // CONV_KC, CONV_KC, 9)
        w = torch.cat([w, torch.zeros_like(w[..., :1])], dim=-1).reshape(co // 64, 64, ci // CONV_KC, CONV_KC, 5, 2)
        w = w.permute(0, 2, 4, 5, 1, 3)                                    # [g, chunk, step, k-group, cout, cin]
    tail = []
    if terms == 16:
        # fp16 split = sp16 pairs (csrc/common.h).  Every output channel is first multiplied by the power of two 2^k_c that puts its largest weight into
        # [2^13, 2^14) (exact), so that both terms are normal fp16 numbers whatever the channel's scale (BatchNorm-folded weights are mostly below 2^-3, where
        # an unscaled second term is an fp16 subnormal); the kernel multiplies bias + residual by 2^k_c and the finished tile by 2^-k_c (both exact).
        # w~ = w 2^k_c rounded to 22 significant bits; term 0 = fp16(w~) to nearest, term 1 = (w~ - term 0) * 2^10, exact.
        amax = weight.detach().float().abs().reshape(co, -1).amax(dim=1)
        k = torch.where(amax > 0, 14 - torch.frexp(amax)[1], torch.zeros_like(amax, dtype=torch.int32)).clamp(-60, 60).to(torch.int32)
        scale = torch.ldexp(torch.ones_like(amax), k)
        cshape = (co // 64, 1, 1, 1, 64, 1)                                # the output channel sits on axis 4 of both images
        ws = (w * scale.reshape(cshape)).contiguous()
        ws = ((ws.view(torch.int32) + 2) & -4).view(torch.float32)         # 22 significant bits, ties away from zero (sp16_round)
        hi = ws.half()
        lo = ((ws - hi.float()) * 1024.0).half()
        parts = [hi, lo]
        tail = [torch.ldexp(torch.ones_like(amax), -k).contiguous().view(torch.uint8).reshape(-1), scale.contiguous().view(torch.uint8).reshape(-1)]
    else:
        parts, rest = [], w
        for _ in range(terms):
            t = rest.bfloat16()
            parts.append(t)
            rest = rest - t.float()
    img = torch.stack(parts, dim=3).contiguous()                           # [g, chunk, step, term, k-group, cout, cin]
    flat = img.view(torch.uint8).reshape(-1)
    out = torch.cat([flat, torch.zeros(16, dtype=torch.uint8, device=flat.device)] + tail)
    assert out.numel() == hip.lib().coalign_conv3x3_emu_weight_bytes_ex(ci, co, terms, int(tap_major))
    return out


LAYOUT_NCHW, LAYOUT_OUT_NHWC, LAYOUT_IN_NHWC, LAYOUT_W_TAPMAJOR, LAYOUT_OUT_SP = 0, 1, 2, 4, 8      # COALIGN_LAYOUT_* of include/coalign_amd.h


def is_channels_last(t: torch.Tensor) -> bool:
    """A 4-d tensor whose memory order is (N, H, W, C) and not at the same time plain NCHW (C > 1, H * W > 1)."""
    return t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last) and not t.is_contiguous()


@_device_op
def conv3x3_emu_bias_act(x: torch.Tensor, w_split: torch.Tensor, bias: torch.Tensor, cout: int, residual: Optional[torch.Tensor] = None,
                         relu: bool = True, terms: int = 3, stride: int = 1, out_channels_last: bool = False, out_split: bool = False):
    """y = act(conv3x3(x, w, stride, padding 1) + bias (+ residual)) with every fp32 product evaluated from 16-bit terms on the matrix cores, fp32 accumulation
    (csrc/conv3x3_emu.hip): ``terms`` = 16 -- the detector's default arithmetic -- sp16 pairs on the fp16 cores (22-bit operands, scale free), 3 / 2 = 3- / 2-way
    bf16 split (the signature's default of 3 dates from rounds 2-3; the detector passes ``backbone.CONV_EMU_TERMS``).  A channels-last ``x`` is read in place by the
    stride-2 variant; ``out_channels_last`` (stride 1) returns a tensor of logical shape [N, C, H, W] in channels-last memory; ``out_split`` (terms 16, no
    residual: the strided layers and the tap-major stride-1 image on an NCHW input) returns a ``SplitMap``."""
    _need_gpu(x, w_split, bias, residual)
    L = hip.lib()
    layout = LAYOUT_NCHW
    if x.dtype == torch.float32 and stride == 2 and is_channels_last(x):
        xc, layout = x, LAYOUT_IN_NHWC
    else:
        xc = _f32c(x)
    N, Cin, H, W = xc.shape
    tapk = 0                                               # the two weight images differ in size (9 : 10): recognised by it
    if w_split.numel() != L.coalign_conv3x3_emu_weight_bytes(Cin, cout, terms):
        if stride != 1 or w_split.numel() != L.coalign_conv3x3_emu_weight_bytes_ex(Cin, cout, terms, 1):
            raise ValueError("split weight image does not match (Cin, Cout, terms)")
        tapk = LAYOUT_W_TAPMAJOR
    if stride not in (1, 2) or (stride == 2 and (residual is not None or (out_channels_last and layout != LAYOUT_IN_NHWC))):
        raise ValueError("stride 2 takes no residual and writes NCHW (channels-last only from a channels-last input)")
    Ho, Wo = (H + stride - 1) // stride, (W + stride - 1) // stride
    out = None
    if out_split:
        if terms != 16 or residual is not None or out_channels_last or (stride == 1 and (not tapk or layout != LAYOUT_NCHW)):
            raise ValueError("SplitMap output: terms 16, no residual; stride 2, or stride 1 with the tap-major image on an NCHW input")
        layout |= LAYOUT_OUT_SP
        out = SplitMap.empty(N, cout, Ho, Wo, xc.device)
        y = out.data
    elif out_channels_last:
        layout |= LAYOUT_OUT_NHWC
        y = torch.empty((N, cout, Ho, Wo), dtype=torch.float32, device=xc.device, memory_format=torch.channels_last)
    else:
        y = torch.empty((N, cout, Ho, Wo), dtype=torch.float32, device=xc.device)
    res = None if residual is None else _f32c(residual)
    if res is not None and res.shape != y.shape:
        raise ValueError("residual shape mismatch")
    ws, ws_bytes = None, 0
    layout |= tapk
    if stride == 1 and (layout & 3) == LAYOUT_NCHW:
        ws_bytes = L.coalign_conv3x3_emu_workspace_bytes_ex(N, Cin, cout, H, W, terms, layout)
    if ws_bytes:
        key = (xc.device, torch.cuda.current_stream(xc.device).cuda_stream, "emu")
        ws = _conv_workspace(key, ws_bytes, xc.device)
    with _Timed("conv3x3_emu_bias_act"):
        hip.check(L.coalign_conv3x3_emu_ex(_ptr(xc), _ptr(w_split), _ptr(_f32c(bias)), _ptr(res), _ptr(y), N, Cin, cout, H, W, int(stride),
                                           int(relu), terms, layout, _ptr(sp_range_flag(xc.device)) if out_split else None, _ptr(ws), 0 if ws is None else ws.numel(), _stream()),
                  "coalign_conv3x3_emu_ex")
    return y if out is None else out


_SP_WS: dict = {}
_SP_WS_RETIRED: list = []
SP_WS_MIN_BYTES = 24 << 20          # the largest stream-K workspace of the shipped configs (256 workgroups x 8 wavefronts x 8 KB + flags) is 16.8 MB


def _sp_workspace(device, ws_bytes: int) -> torch.Tensor:
    """The stream-K workspace of ``coalign_conv3x3_sp`` for the CURRENT stream of ``device``: zero-initialised once (the launches leave their flag words zero
    again), one per stream (launches on a stream are ordered), sized for every shipped shape on first use so that it never grows inside a captured frame; a
    workspace that still has to grow is replaced and kept referenced (captured graphs hold its pointer)."""
    key = (str(device), torch.cuda.current_stream(device).cuda_stream)
    ws = _SP_WS.get(key)
    if ws is None or ws.numel() < ws_bytes:
        if ws is not None:
            _SP_WS_RETIRED.append(ws)
        ws = _SP_WS[key] = torch.zeros(max(ws_bytes, SP_WS_MIN_BYTES), dtype=torch.uint8, device=device)
    return ws


# Measurement switch (round 6): COALIGN_SP_GEO="25x88:121,50x176:81" overrides the tile geometry `coalign_conv3x3_sp` picks for maps of the listed H x W
# (codes: include/coalign_amd.h (9e)); read once at import, the dict is read per call.
import os as _os
SP_GEO_OVERRIDE = {tuple(int(v) for v in k.split("x")): int(g) for k, g in (item.split(":") for item in _os.environ.get("COALIGN_SP_GEO", "").split(",") if item)}

SP_RES_NONE, SP_RES_SP, SP_RES_NHWC = 0, 1, 2      # residual_kind of coalign_conv3x3_sp
SP_OUT_SP, SP_OUT_NHWC = 1, 2                      # out_kind


@_device_op
def conv3x3_sp(x: "SplitMap", w_split: torch.Tensor, bias: torch.Tensor, cout: int, residual=None, relu: bool = True, out_split: bool = True, geometry: int = 0,
               out_both: bool = False):
    """y = act(conv3x3(x, w, stride 1, padding 1) + bias (+ residual)) on a ``SplitMap`` input (include/coalign_amd.h (9e), csrc/conv3x3_sp.hip): the fp16
    mode's arithmetic with the operand split done by the producer.  ``w_split``: the tap-major terms-16 image of ``pack_conv3x3_emu_weight``; ``residual``: a
    SplitMap, a float32 tensor (converted to channels-last memory if it is not) or None; returns a SplitMap (``out_split``) or a float32 tensor of logical
    shape [N, C, H, W] in channels-last memory; ``out_both`` (round 6, ``coalign_conv3x3_sp_both``): the pair (channels-last float32 tensor, SplitMap of it)."""
    if not isinstance(x, SplitMap):
        raise TypeError("conv3x3_sp reads a SplitMap")
    if out_both:
        out_split = False
    _need_gpu(x.data, w_split, bias)
    L = hip.lib()
    N, Cin, H, W = x.shape
    if geometry == 0 and SP_GEO_OVERRIDE:
        geometry = SP_GEO_OVERRIDE.get((H, W), 0)
    if w_split.numel() != L.coalign_conv3x3_emu_weight_bytes_ex(Cin, cout, 16, 1):
        raise ValueError("conv3x3_sp needs the tap-major terms-16 weight image of (Cin, Cout)")
    res_kind, res_t = SP_RES_NONE, None
    if isinstance(residual, SplitMap):
        res_kind, res_t = SP_RES_SP, residual.data
        if residual.shape != (N, cout, H, W):
            raise ValueError("residual shape mismatch")
    elif residual is not None:
        res_kind, res_t = SP_RES_NHWC, to_nhwc(residual)
        if tuple(res_t.shape) != (N, cout, H, W):
            raise ValueError("residual shape mismatch")
    if out_split:
        out = SplitMap.empty(N, cout, H, W, x.device)
        y = out.data
    else:
        out = y = torch.empty((N, cout, H, W), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
        if not nhwc_memory(y):
            out = y = torch.empty((N, H, W, cout), dtype=torch.float32, device=x.device).permute(0, 3, 1, 2)
    ws_bytes = L.coalign_conv3x3_sp_workspace_bytes(N, Cin, cout, H, W, int(geometry))
    ws = _sp_workspace(x.device, ws_bytes) if ws_bytes else None
    if out_both:
        both = SplitMap.empty(N, cout, H, W, x.device)
        with _Timed("conv3x3_sp"):
            hip.check(L.coalign_conv3x3_sp_both(_ptr(x.data), _ptr(w_split), _ptr(_f32c(bias)), _ptr(res_t), res_kind, _ptr(y), _ptr(both.data), N, Cin, cout, H, W, int(relu), int(geometry),
                                                _ptr(sp_range_flag(x.device)), _ptr(ws), 0 if ws is None else ws.numel(), _stream()), "coalign_conv3x3_sp_both")
        return out, both
    with _Timed("conv3x3_sp"):
        hip.check(L.coalign_conv3x3_sp(_ptr(x.data), _ptr(w_split), _ptr(_f32c(bias)), _ptr(res_t), res_kind, _ptr(y), SP_OUT_SP if out_split else SP_OUT_NHWC,
                                       N, Cin, cout, H, W, int(relu), int(geometry), _ptr(sp_range_flag(x.device)) if out_split else None,
                                       _ptr(ws), 0 if ws is None else ws.numel(), _stream()), "coalign_conv3x3_sp")
    return out


@_device_op
def sp_pack_rows(sc: "SparseCanvas") -> torch.Tensor:
    """The feature rows of a ``SparseCanvas`` as sp16 rows [M, C / 16, 4, 8] float16 (``coalign_sp_pack_rows``, include/coalign_amd.h (9f)): what the LDS-DMA
    gather of ``conv3x3_sp_s2`` reads.  Rows at and beyond the device-side count (if the canvas has one) are left untouched."""
    L = hip.lib()
    M, C = int(sc.feats.shape[0]), sc.C
    out = torch.empty((max(M, 1), C // 16, 4, 8), dtype=torch.float16, device=sc.device)[:M]
    with _Timed("sp_pack_rows"):
        hip.check(L.coalign_sp_pack_rows(_rows_ptr(sc.feats), M, _ptr(sc.count_dev) if sc.count_dev is not None else _ptr(getattr(sc, "count_word", None)), C, _rows_ptr(out),
                                         _ptr(sp_range_flag(sc.device)), _stream()), "coalign_sp_pack_rows")
    return out


def pack_conv1x1_sp_weight(weight: torch.Tensor) -> torch.Tensor:
    """[Cout, Cin] (or [Cout, Cin, 1, 1]) fp32 -> the sp16 image of the 1 x 1 skip convolution that rides in ``conv3x3_sp_s2`` (include/coalign_amd.h (9g)): the
    centre-tap slice of the tap-major terms-16 image of ``pack_conv3x3_emu_weight`` (same per-output-channel scale rule), as uint8."""
    w = weight.detach().float().reshape(weight.shape[0], weight.shape[1])
    co, ci = w.shape
    w3 = torch.zeros((co, ci, 3, 3), dtype=torch.float32, device=w.device)
    w3[:, :, 1, 1] = w
    img = pack_conv3x3_emu_weight(w3, 16, True)
    body = co * ci * 9 * 4                                    # two fp16 terms per weight
    taps = img[:body].view(torch.float16).reshape(co // 64, ci // 16, 9, 2 * 2 * 64 * 8)
    out = torch.cat([taps[:, :, 4].contiguous().view(torch.uint8).reshape(-1), img[body:]])
    assert out.numel() == hip.lib().coalign_conv1x1_sp_weight_bytes(ci, co)
    return out


@_device_op
def conv3x3_sp_s2(x, w_split: torch.Tensor, bias: torch.Tensor, cout: int, relu: bool = True, w_skip: Optional[torch.Tensor] = None):
    """y = act(conv3x3(x, w, stride 2, padding 1) + bias) on split operands (include/coalign_amd.h (9f), csrc/conv3x3_sp_s2.hip): ``x`` a ``SplitMap`` or a
    ``SparseCanvas`` (its rows are packed to sp16 rows first); ``w_split``: the tap-major terms-16 image of ``pack_conv3x3_emu_weight``; returns a ``SplitMap``.
    ``w_skip`` (``pack_conv1x1_sp_weight``; (9g)): the block's 1 x 1 / stride-2 skip convolution rides along as a tenth tap -- returns (SplitMap, skip map: float32
    of logical shape [N, Cout, Ho, Wo] in channels-last memory, no bias, no ReLU)."""
    L = hip.lib()
    N, Cin, H, W = x.shape
    Ho, Wo = (H + 1) // 2, (W + 1) // 2
    if w_split.numel() != L.coalign_conv3x3_emu_weight_bytes_ex(Cin, cout, 16, 1):
        raise ValueError("conv3x3_sp_s2 needs the tap-major terms-16 weight image of (Cin, Cout)")
    if w_skip is not None and w_skip.numel() != L.coalign_conv1x1_sp_weight_bytes(Cin, cout):
        raise ValueError("conv3x3_sp_s2: the skip image does not match (Cin, Cout)")
    out = SplitMap.empty(N, cout, Ho, Wo, x.device)
    skip = None if w_skip is None else torch.empty((N, Ho, Wo, cout), dtype=torch.float32, device=x.device).permute(0, 3, 1, 2)
    flag = _ptr(sp_range_flag(x.device))
    b = _f32c(bias)
    if isinstance(x, SparseCanvas):
        x.check_current()
        rows = sp_pack_rows(x)
        with _Timed("conv3x3_sp_s2_sparse"):
            if w_skip is None:
                hip.check(L.coalign_conv3x3_sp_s2_sparse(_rows_ptr(rows), int(rows.shape[0]), _ptr(x.stamps), _ptr(x.state), _ptr(w_split), _ptr(b), _ptr(out.data),
                                                         N, Cin, cout, H, W, int(relu), flag, _stream()), "coalign_conv3x3_sp_s2_sparse")
            else:
                hip.check(L.coalign_conv3x3_sp_s2_skip_sparse(_rows_ptr(rows), int(rows.shape[0]), _ptr(x.stamps), _ptr(x.state), _ptr(w_split), _ptr(b), _ptr(w_skip), _ptr(out.data),
                                                              _ptr(skip), N, Cin, cout, H, W, int(relu), flag, _stream()), "coalign_conv3x3_sp_s2_skip_sparse")
        return out if w_skip is None else (out, skip)
    if not isinstance(x, SplitMap):
        raise TypeError("conv3x3_sp_s2 reads a SplitMap or a SparseCanvas")
    _need_gpu(x.data, w_split, bias)
    with _Timed("conv3x3_sp_s2"):
        if w_skip is None:
            hip.check(L.coalign_conv3x3_sp_s2(_ptr(x.data), _ptr(w_split), _ptr(b), _ptr(out.data), N, Cin, cout, H, W, int(relu), flag, _stream()), "coalign_conv3x3_sp_s2")
        else:
            hip.check(L.coalign_conv3x3_sp_s2_skip(_ptr(x.data), _ptr(w_split), _ptr(b), _ptr(w_skip), _ptr(out.data), _ptr(skip), N, Cin, cout, H, W, int(relu), flag, _stream()),
                      "coalign_conv3x3_sp_s2_skip")
    return out if w_skip is None else (out, skip)


def conv3x3_sp_is_split(N: int, Cin: int, cout: int, H: int, W: int, geometry: int = 0) -> bool:
    """True when ``conv3x3_sp`` cuts this shape's tiles between workgroups (stream-K: another, still deterministic, summation order)."""
    return hip.lib().coalign_conv3x3_sp_workspace_bytes(N, Cin, cout, H, W, int(geometry)) > 0


def pack_conv3x3_wino_weight(weight: torch.Tensor) -> torch.Tensor:
    """[Cout, Cin, 3, 3] fp32 (BatchNorm folded) -> the operand image of ``coalign_conv3x3_wino`` (include/coalign_amd_lab.h, laboratory library), uint8:
    U = G g G^T in float64, split into three bf16 terms (term 0 = bf16(U), term 1 = bf16(U - term 0), term 2 = bf16(U - term 0 - term 1), the
    residuals exact in float64), stored [Cout / 64][Cin / 16][h][wave = 4 c + i][jj][term][lane][8]: the 16 bytes lane ``l`` of wavefront (i, c) loads for
    transform position (i, 2 h + jj) are U[i, 2 h + jj, 64 g + 32 c + l % 32, 16 k + 8 (l // 32) : + 8]."""
    co, ci, kh, kw = weight.shape
    if (kh, kw) != (3, 3) or co % 64 or ci % 16:
        raise ValueError(f"conv3x3_wino needs 3x3 weights with Cout % 64 == 0 and Cin % 16 == 0, got {tuple(weight.shape)}")
    Gm = torch.tensor([[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]], dtype=torch.float64, device=weight.device)
    U = torch.einsum("ia,ocab,jb->ijoc", Gm, weight.detach().double(), Gm)                 # [4 i, 4 j, Cout, Cin]
    parts, rest = [], U
    for _ in range(3):
        t = rest.float().bfloat16()
        parts.append(t)
        rest = rest - t.double()
    T = torch.stack(parts, 0)                                                              # [term, i, j, Cout, Cin] bf16
    T = T.view(3, 4, 2, 2, co // 64, 2, 32, ci // 16, 2, 8)                                # [term, i, h, jj, g, c, m, k, half, e]
    img = T.permute(4, 7, 2, 5, 1, 3, 0, 8, 6, 9).contiguous()                             # [g, k, h, c, i, jj, term, half, m, e]
    out = torch.cat([img.view(torch.uint8).reshape(-1), torch.zeros(16, dtype=torch.uint8, device=img.device)])      # + 16 zero bytes (zero padding source)
    assert out.numel() == hip.lab_lib().coalign_conv3x3_wino_weight_bytes(ci, co)
    return out


def nhwc_memory(t: torch.Tensor) -> bool:
    """A 4-d float32 tensor of logical shape [N, C, H, W] whose memory is dense [N, H, W, C]."""
    return t.dim() == 4 and t.dtype == torch.float32 and t.permute(0, 2, 3, 1).is_contiguous()


def to_nhwc(t: torch.Tensor) -> torch.Tensor:
    return t if nhwc_memory(t) else t.float().contiguous(memory_format=torch.channels_last)


@_device_op
def conv3x3_wino(x: torch.Tensor, u_split: torch.Tensor, bias: torch.Tensor, cout: int, residual: Optional[torch.Tensor] = None,
                 relu: bool = True, tile_block_w: int = 0) -> torch.Tensor:
    """y = act(conv3x3(x, w, stride 1, padding 1) + bias (+ residual)) as Winograd F(2x2, 3x3) on the split-bf16 matrix cores
    (csrc/conv3x3_wino.hip).  x / residual / y: logical [N, C, H, W] in channels-last memory (x and residual are converted if they are not)."""
    _need_gpu(x, u_split, bias, residual)
    L = hip.lab_lib()          # laboratory library only (include/coalign_amd_lab.h)
    xc = to_nhwc(x)
    N, Cin, H, W = xc.shape
    if u_split.numel() != L.coalign_conv3x3_wino_weight_bytes(Cin, cout):
        raise ValueError("transformed weight image does not match (Cin, Cout)")
    y = torch.empty((N, cout, H, W), dtype=torch.float32, device=xc.device, memory_format=torch.channels_last)
    if not nhwc_memory(y):                                 # (C == 1 or H * W == 1 cannot occur here: Cout % 64 == 0)
        y = torch.empty((N, H, W, cout), dtype=torch.float32, device=xc.device).permute(0, 3, 1, 2)
    res = None if residual is None else to_nhwc(residual)
    if res is not None and res.shape != y.shape:
        raise ValueError("residual shape mismatch")
    with _Timed("conv3x3_wino"):
        hip.check(L.coalign_conv3x3_wino(_ptr(xc), _ptr(u_split), _ptr(_f32c(bias)), _ptr(res), _ptr(y), N, Cin, cout, H, W, int(relu),
                                         int(tile_block_w), _stream()), "coalign_conv3x3_wino")
    return y


def pack_pointwise_weight(weight: torch.Tensor, transposed: bool) -> torch.Tensor:
    """ConvTranspose2d weight [Cin, Cout, k, k] -> [Cin, Cout * k * k] (a view of the same layout); Conv2d 1x1 weight
    [Cout, Cin, 1, 1] -> [Cin, Cout padded to a multiple of 32] (zero columns)."""
    w = weight.detach().float()
    if transposed:
        return w.reshape(w.shape[0], -1).contiguous()
    co, ci = w.shape[0], w.shape[1]
    out = torch.zeros((ci, (co + 31) // 32 * 32), dtype=torch.float32, device=w.device)
    out[:, :co] = w.reshape(co, ci).t()
    return out


def pack_pointwise_emu_weight(w_packed: torch.Tensor) -> torch.Tensor:
    """[Cin, M] fp32 (``pack_pointwise_weight``; Cin % 16 == 0, M % 32 == 0) -> the pre-split operand image of ``coalign_pointwise_conv_emu``
    (include/coalign_amd.h): int16 [M / 32, Cin / 16, 3 terms, 64 lanes, 8] = the bf16 bit patterns of term t of
    W[16 step + 8 (lane // 32) + j, 32 tile + lane % 32].  Terms: bf16(w), bf16(w - h), bf16(w - h - m), round-to-nearest-even (torch's
    float32 -> bfloat16 conversion, the rounding of v_cvt_pk_bf16_f32)."""
    w = w_packed.detach().float()
    Cin, M = w.shape
    if Cin % 16 or M % 32:
        raise ValueError("pointwise emu weights: Cin % 16 == 0 and M % 32 == 0")
    h = w.to(torch.bfloat16)
    r1 = w - h.float()
    m = r1.to(torch.bfloat16)
    lo = (r1 - m.float()).to(torch.bfloat16)
    terms = torch.stack([h, m, lo], 0).view(torch.int16)                       # [3, Cin, M]
    img = terms.view(3, Cin // 16, 2, 8, M // 32, 32)                           # [t, step, half, j, tile, row]
    img = img.permute(4, 1, 0, 2, 5, 3).contiguous()                           # [tile, step, t, half, row, j]
    return img.view(M // 32, Cin // 16, 3, 64, 8)


@_device_op
def pointwise_conv(x: torch.Tensor, w_packed: torch.Tensor, bias: torch.Tensor, cout: int, up: int = 1, in_stride: int = 1, relu: bool = True,
                   out: Optional[torch.Tensor] = None, c_off: int = 0, out_channels_last: bool = False) -> torch.Tensor:
    """One-launch pointwise layer (include/coalign_amd.h (10)): ``up`` > 1 = non-overlapping transposed convolution, ``in_stride`` 2 =
    1x1 stride-2 convolution.  ``out`` [N, Ctot, H', W'] + ``c_off`` select a channel slice of a larger (concatenated) tensor.
    ``w_packed``: [Cin, M] float32 (fp32 matrix cores) or the int16 image of ``pack_pointwise_emu_weight`` (split-bf16 matrix cores)."""
    _need_gpu(x, w_packed, bias)
    emu = w_packed.dtype == torch.int16
    L = hip.lib()
    nhwc = x.dtype == torch.float32 and is_channels_last(x) and x.shape[1] % 4 == 0        # read in place, no NCHW copy
    xc = x if nhwc else _f32c(x)
    N, Cin, Hin, Win = xc.shape
    Ho, Wo = (Hin + in_stride - 1) // in_stride * up, (Win + in_stride - 1) // in_stride * up
    if isinstance(out, SplitMap):                          # round 5: the layer writes its channel slice of a SplitMap (the shrink header's input)
        if not emu or out_channels_last or tuple(out.shape[2:]) != (Ho, Wo) or out.shape[0] != N or cout % 16 or c_off % 16 or c_off + cout > out.shape[1]:
            raise ValueError("SplitMap output: split-bf16 weight image, [N, Ctot, Ho, Wo] map, Cout and c_off multiples of 16")
        if w_packed.dim() != 5 or w_packed.shape[1] * 16 != Cin or not w_packed.is_contiguous() or w_packed.shape[0] * 32 != cout * up * up:
            raise ValueError("split weight image does not match (Cin, Cout * up * up)")
        with _Timed("pointwise_conv"):
            hip.check(L.coalign_pointwise_conv_emu_sp(_ptr(xc), _ptr(w_packed), _ptr(_f32c(bias)), _ptr(out.data), N, Cin, Hin, Win, in_stride, cout, up, w_packed.shape[0] * 32,
                                                      out.shape[1], c_off, int(relu), int(nhwc), _ptr(sp_range_flag(xc.device)), _stream()), "coalign_pointwise_conv_emu_sp")
        return out
    if out_channels_last:                                  # (up = 1) a fresh [N, cout, Ho, Wo] tensor in channels-last memory
        if out is not None or up != 1 or cout % 4:
            raise ValueError("channels-last output: up = 1, Cout % 4 == 0, no output slice")
        out = torch.empty((N, cout, Ho, Wo), dtype=torch.float32, device=xc.device, memory_format=torch.channels_last)
    elif out is None:
        out = torch.empty((N, cout, Ho, Wo), dtype=torch.float32, device=xc.device)
    if tuple(out.shape[2:]) != (Ho, Wo) or out.shape[0] != N or not (nhwc_memory(out) if out_channels_last else out.is_contiguous()) or out.dtype != torch.float32:
        raise ValueError("output buffer shape / layout mismatch")
    nhwc_flags = int(nhwc) | (2 if out_channels_last else 0)
    with _Timed("pointwise_conv"):
        if emu:
            if w_packed.dim() != 5 or w_packed.shape[1] * 16 != Cin or not w_packed.is_contiguous():
                raise ValueError("split weight image does not match Cin")
            hip.check(L.coalign_pointwise_conv_emu(_ptr(xc), _ptr(w_packed), _ptr(_f32c(bias)), _ptr(out), N, Cin, Hin, Win, in_stride, cout, up,
                                                   w_packed.shape[0] * 32, out.shape[1], c_off, int(relu), nhwc_flags, _stream()), "coalign_pointwise_conv_emu")
        else:
            hip.check(L.coalign_pointwise_conv_ex(_ptr(xc), _ptr(w_packed), _ptr(_f32c(bias)), _ptr(out), N, Cin, Hin, Win, in_stride, cout, up,
                                                  w_packed.shape[1], out.shape[1], c_off, int(relu), nhwc_flags, _stream()), "coalign_pointwise_conv")
    return out


@_device_op
@_device_op
def pointwise_heads_split(layers, out: "SplitMap", relu: bool = True) -> "SplitMap":
    """The up-sampling heads of all scales in ONE launch (``coalign_pointwise_conv_emu_sp_multi``, include/coalign_amd.h (10d)): ``layers`` = [(x, w_emu, bias, cout,
    up, c_off), ...] (at most four), each writing its channel slice of the SplitMap ``out``.  Bit-identical to one ``pointwise_conv(..., out=out)`` per layer."""
    L = hip.lib()
    n = len(layers)
    N = out.shape[0]
    keep, xs, ws, bs = [], [], [], []
    ci, hi, wi, co, ups, offs, nh = [], [], [], [], [], [], []
    for (x, w_emu, bias, cout, up, c_off) in layers:
        _need_gpu(x, w_emu, bias)
        nhwc = x.dtype == torch.float32 and is_channels_last(x) and x.shape[1] % 4 == 0
        xc = x if nhwc else _f32c(x)
        b = _f32c(bias)
        if w_emu.dtype != torch.int16 or w_emu.dim() != 5 or w_emu.shape[1] * 16 != xc.shape[1] or not w_emu.is_contiguous() or w_emu.shape[0] * 32 != cout * up * up or xc.shape[0] != N:
            raise ValueError("pointwise_heads_split: split weight image of (Cin, Cout * up * up), one batch size")
        keep += [xc, b]
        xs.append(xc.data_ptr()); ws.append(w_emu.data_ptr()); bs.append(b.data_ptr())
        ci.append(xc.shape[1]); hi.append(xc.shape[2]); wi.append(xc.shape[3]); co.append(cout); ups.append(up); offs.append(c_off); nh.append(int(nhwc))
    arr_p = lambda v: (ctypes.c_void_p * n)(*v)
    arr_i = lambda v: (ctypes.c_int32 * n)(*v)
    with _Timed("pointwise_heads_split"):
        hip.check(L.coalign_pointwise_conv_emu_sp_multi(n, arr_p(xs), arr_p(ws), arr_p(bs), arr_i(ci), arr_i(hi), arr_i(wi), arr_i(co), arr_i(ups), arr_i(offs), arr_i(nh), _ptr(out.data),
                                                        N, out.shape[1], int(relu), _ptr(sp_range_flag(out.device)), _stream()), "coalign_pointwise_conv_emu_sp_multi")
    return out


def pack_heads_sp_weight(weight: torch.Tensor) -> torch.Tensor:
    """[M <= 32, Cin(, 1, 1)] merged head weights -> the image ``heads_sp`` reads (include/coalign_amd.h (10e)): ``pack_conv1x1_sp_weight`` of the weights padded to 64 rows."""
    w = weight.detach().float().reshape(weight.shape[0], weight.shape[1])
    if w.shape[0] > 32:
        raise ValueError("heads_sp serves at most 32 head channels")
    pad = torch.zeros((64, w.shape[1]), dtype=torch.float32, device=w.device)
    pad[: w.shape[0]] = w
    return pack_conv1x1_sp_weight(pad)


@_device_op
def heads_sp(x: "SplitMap", w_sp: torch.Tensor, bias: torch.Tensor, M: int) -> torch.Tensor:
    """The merged 1 x 1 heads on a SplitMap (``coalign_heads_sp``, (10e)): -> [N, M, H, W] float32, no activation."""
    if not isinstance(x, SplitMap):
        raise TypeError("heads_sp reads a SplitMap")
    _need_gpu(x.data, w_sp, bias)
    L = hip.lib()
    N, Cin, H, W = x.shape
    if w_sp.numel() != L.coalign_conv1x1_sp_weight_bytes(Cin, 64):
        raise ValueError("heads_sp: the weight image does not match Cin")
    y = torch.empty((N, M, H, W), dtype=torch.float32, device=x.device)
    with _Timed("heads_sp"):
        hip.check(L.coalign_heads_sp(_ptr(x.data), _ptr(w_sp), _ptr(_f32c(bias)), _ptr(y), N, Cin, M, H, W, _stream()), "coalign_heads_sp")
    return y


def boxes_overlap_bev(boxes_a: torch.Tensor, boxes_b: torch.Tensor) -> torch.Tensor:
    """OpenPCDet-semantics fp32 BEV overlap AREA matrix [Na, Nb] of (x, y, z, dx, dy, dz, heading) boxes."""
    _need_gpu(boxes_a, boxes_b)
    L = hip.lib()
    a, b = _f32c(boxes_a), _f32c(boxes_b)
    out = torch.zeros((a.shape[0], b.shape[0]), dtype=torch.float32, device=a.device)
    hip.check(L.coalign_boxes_overlap_bev(_ptr(a), a.shape[0], _ptr(b), b.shape[0], _ptr(out), _stream()), "coalign_boxes_overlap_bev")
    return out


@_device_op
def pcdet_nms(boxes_sorted: torch.Tensor, thresh: float, normal: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    """boxes_sorted [n, 7] (descending score order) -> (keep int32 [n] positions, keep_count int32 [1]), both on the device."""
    _need_gpu(boxes_sorted)
    L = hip.lib()
    b = _f32c(boxes_sorted)
    n = b.shape[0]
    keep = torch.empty(max(n, 1), dtype=torch.int32, device=b.device)
    cnt = torch.zeros(1, dtype=torch.int32, device=b.device)
    ws_bytes = L.coalign_pcdet_nms_workspace_bytes(n)
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=b.device)
    hip.check(L.coalign_pcdet_nms(_ptr(b), n, float(thresh), int(normal), _ptr(keep), _ptr(cnt), _ptr(ws), ws_bytes, _stream()), "coalign_pcdet_nms")
    return keep, cnt
