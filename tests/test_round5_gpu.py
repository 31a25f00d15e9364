"""Round-5 GPU tests (all through the C ABI).

* the fp16 split ("sp16 pairs", csrc/common.h) is scale free: swept over weight scales, activation scales and heavy-tailed draws against a float64
  convolution, admitted by the rule that admitted the bf16 3-way split (no worse than the native fp32 matrix kernel on the same data);
* parity against the CPU oracle on TRAINED-LIKE parameters (coalign_amd.synthetic.fill_parameters_trained_like_: per-layer folded scales 2^-8 ... 2^4,
  Student-t weights, near-zero BatchNorm weights, dead channels) on the workloads of BASELINE configs 2, 3 and 4
  (opencood/models/sub_modules/resblock.py:53-69 is fp32 at any weight scale; opencood/tools/train_utils.py:29-74 is what gets loaded).
"""
import math

import numpy as np
import pytest
import torch

from oracle import coalign_oracle as oracle
from coalign_amd import ops
from coalign_amd.config import builtin_config
from coalign_amd.detector import build_model, to_device
from coalign_amd.postprocess import build_postprocessor
from coalign_amd.synthetic import calibrate_heads_, fill_parameters_trained_like_, make_frame

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
T = torch.from_numpy

FP16_SHAPES = [(5, 64, 64, 100, 352), (5, 128, 128, 50, 176), (5, 256, 256, 25, 88), (1, 384, 256, 100, 352), (1, 256, 256, 100, 352), (2, 64, 64, 100, 252)]


def conv64(x, w, b, r, stride=1):
    """relu(conv3x3(x, w, pad 1) + b + r) in float64 as nine matrix products (rocBLAS dgemm; MIOpen's float64 convolution is far slower)."""
    N, Ci, H, W = x.shape
    xp = torch.nn.functional.pad(x.double(), (1, 1, 1, 1))
    Ho, Wo = (H + stride - 1) // stride, (W + stride - 1) // stride
    out = torch.zeros((N, w.shape[0], Ho, Wo), dtype=torch.float64, device=x.device)
    wd = w.double()
    for dy in range(3):
        for dx in range(3):
            out += torch.einsum("oc,nchw->nohw", wd[:, :, dy, dx], xp[:, :, dy:dy + stride * Ho:stride, dx:dx + stride * Wo:stride])
    out += b.double().view(1, -1, 1, 1)
    if r is not None:
        out += r.double()
    return torch.relu(out)


def draw(shape, g, heavy):
    z = torch.randn(shape, generator=g, device=DEV)
    if not heavy:
        return z
    chi = (torch.randn((3,) + tuple(shape), generator=g, device=DEV) ** 2).sum(0) / 3.0
    return z / chi.sqrt() / 3.0 ** 0.5                      # Student-t, nu = 3, unit variance


def sweep_case(shape, w_rms, x_scale, heavy, seed):
    N, Ci, Co, H, W = shape
    g = torch.Generator(device=DEV).manual_seed(seed)
    x = draw((N, Ci, H, W), g, heavy) * x_scale
    w = draw((Co, Ci, 3, 3), g, heavy) * w_rms
    s_out = w_rms * x_scale * (9 * Ci) ** 0.5
    b = torch.randn(Co, generator=g, device=DEV) * 0.1 * s_out
    r = torch.randn((N, Co, H, W), generator=g, device=DEV) * s_out
    return x, w, b, r


CELLS = [(wr, xs, hv) for wr in (2e-1, 2e-2, 2e-3, 2e-4) for xs in (1e-2, 1.0, 1e2) for hv in (False, True)]


@pytest.mark.parametrize("shape", FP16_SHAPES)
def test_fp16_split_is_scale_free_against_float64(shape):
    """every cell of weight scale x activation scale x {Gaussian, Student-t(3)} on every stride-1 backbone shape: the fp16 split's error
    against the float64 convolution is <= max(native fp32 kernel's error on the same data, 2e-6 of the output scale) -- round 4's unscaled split was
    9e-6 ... 9e-5 at w_rms 2e-3 ... 2e-4.  Residual + ReLU as in resblock.py:53-69; NCHW and channels-last outputs."""
    N, Ci, Co, H, W = shape
    worst = 0.0
    for ci, (wr, xs, hv) in enumerate(CELLS):
        x, w, b, r = sweep_case(shape, wr, xs, hv, seed=1000 * sum(shape) + ci)
        want = conv64(x, w, b, r)
        scale = float(want.abs().max())
        err = lambda y: float((y.double() - want).abs().max()) / scale
        w16 = ops.pack_conv3x3_emu_weight(w, 16, True)
        e16 = err(ops.conv3x3_emu_bias_act(x, w16, b, Co, r, True, 16))
        enat = err(ops.conv3x3_bias_act(x, ops.pack_conv3x3_weight(w), b, r, True))
        worst = max(worst, e16 / max(enat, 2e-6))
        if ci % 6 == 0:
            e16cl = err(ops.conv3x3_emu_bias_act(x, w16, b, Co, r, True, 16, out_channels_last=True))
            assert e16cl <= max(enat, 2e-6), (shape, wr, xs, hv, e16cl, enat)
        print(f"\n{shape} w_rms {wr:g} x {xs:g} {'student-t' if hv else 'gauss'}: fp16 split {e16:.2e}, native fp32 {enat:.2e}", end="")
        assert e16 <= max(enat, 2e-6), (shape, wr, xs, hv, e16, enat)
    print(f"\n{shape}: worst fp16-split error / max(native, 2e-6) = {worst:.2f}")


def test_fp16_split_scale_free_with_mixed_channel_scales_and_strided_layers():
    """One layer whose output channels differ by 2^-20 ... 2^6 in weight scale (near-zero BatchNorm weights next to large ones), stride 1 and the
    strided kernels (NCHW / channels-last in and out): every channel is as accurate as the native fp32 kernel allows, measured PER CHANNEL."""
    for (N, Ci, Co, H, W, stride) in ((2, 64, 64, 50, 176, 1), (2, 64, 128, 100, 352, 2), (2, 128, 256, 50, 126, 2)):
        g = torch.Generator(device=DEV).manual_seed(H + Co + stride)
        x = torch.relu(torch.randn((N, Ci, H, W), generator=g, device=DEV)) * 0.05
        cs = torch.logspace(-6, 1.8, Co, device=DEV)[torch.randperm(Co, generator=g, device=DEV)]
        w = torch.randn((Co, Ci, 3, 3), generator=g, device=DEV) * 0.03 * cs.view(-1, 1, 1, 1)
        w[3] = 0
        b = torch.randn(Co, generator=g, device=DEV) * 0.01 * cs
        want = conv64(x, w, b, None, stride)
        ch_scale = want.abs().amax(dim=(0, 2, 3)).clamp_min(1e-30)
        w16 = ops.pack_conv3x3_emu_weight(w, 16, stride == 1)
        variants = [(x, False)]
        if stride == 2:
            xcl = x.contiguous(memory_format=torch.channels_last)
            variants += [(xcl, False), (xcl, True)]
        else:
            variants += [(x, True)]
        for xin, cl_out in variants:
            got = ops.conv3x3_emu_bias_act(xin, w16, b, Co, None, True, 16, stride=stride, out_channels_last=cl_out)
            e = ((got.double() - want).abs().amax(dim=(0, 2, 3)) / ch_scale)
            live = want.abs().amax(dim=(0, 2, 3)) > 0
            assert float(e[live].max()) < 3e-6, (N, Ci, Co, H, W, stride, cl_out, float(e[live].max()))
            assert float(got[:, 3].abs().max()) == float(torch.relu(b[3]))          # the all-zero channel is exactly relu(bias)


def test_fp16_split_operating_range():
    """|x| <= 65504 is the documented range of the mode (22 significant bits down to 2^-14); beyond it the operands clamp at 65504 + 63.97:
    finite results, never an infinity or a NaN."""
    N, Ci, Co, H, W = 1, 16, 64, 8, 32
    g = torch.Generator(device=DEV).manual_seed(3)
    x = torch.randn((N, Ci, H, W), generator=g, device=DEV)
    w = torch.randn((Co, Ci, 3, 3), generator=g, device=DEV) / 12.0
    b = torch.randn(Co, generator=g, device=DEV)
    w16 = ops.pack_conv3x3_emu_weight(w, 16, True)
    for s, bound in ((6.0e4 / float(x.abs().max()), 5e-6), (1e-4, 5e-6), (2.0 ** -13, 5e-6)):
        xs = x * s
        bs = b * s
        want = conv64(xs, w, bs, None)
        got = ops.conv3x3_emu_bias_act(xs, w16, bs, Co, None, True, 16)
        assert float((got.double() - want).abs().max() / want.abs().max()) < bound, s
    got = ops.conv3x3_emu_bias_act(x * 3.0e4, w16, b, Co, None, False, 16)            # |x| up to ~1.2e5
    assert torch.isfinite(got).all()


# ------------------------------------------------------------------------------------------------ parity on trained-like parameters
def rel_err(got, ref):
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    return float((got - ref).abs().max()) / max(float(ref.abs().max()), 1e-30)


def trained_like(config, frame, target, seed):
    h = builtin_config(config)
    model = build_model(h)
    scales = fill_parameters_trained_like_(model, seed=seed)
    model = model.to(DEV).eval()
    pp = build_postprocessor(h["postprocess"], False)
    fd = to_device(frame(h), DEV)
    calibrate_heads_(model, fd, pp.params["target_args"]["score_threshold"], target)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    lo, hi = min(scales.values()), max(scales.values())
    print(f"\n{config}: {len(scales)} scaled layers, maps at 2^{lo:.1f} ... 2^{hi:.1f}")
    return h, model, pp, fd, sd


@pytest.mark.parametrize("case", [("opv2v_coalign", 2, 6000, 77, False), ("opv2v_coalign", 5, 8000, 304, False), ("dairv2x_coalign", 2, 7000, 5, True)],
                         ids=["cfg2_2agents", "cfg3_benchmarked_5x8000", "cfg4_dairv2x"])
def test_trained_like_parameters_vs_oracle_default_arithmetic(case):
    """the default arithmetic on a parameter set with a trained checkpoint's statistics, full geometry, against the CPU oracle:
    head outputs within 1e-4 of their scale (north star: 1e-3), identical candidate count and identical detections from the same logits."""
    config, n_agents, pillars, seed, infra = case
    mk = lambda h: make_frame(h, n_agents, pillars_per_agent=pillars, seed=seed, noise=(0.2, 0.2), infra_agent=infra)
    h, model, pp, fd, sd = trained_like(config, mk, 500, seed=2)
    frame = mk(h)
    with torch.no_grad():
        out = model(fd)
        ref = oracle.coalign_forward(sd, h["model"]["args"], frame)
    for k in ("cls_preds", "reg_preds", "dir_preds"):
        from conftest import assert_elementwise      # (round 6: element-wise; the floor for these parameter sets is set by the fp32 reference's own
        # distance from float64 on them -- tests/test_round6_gpu.py::test_end_to_end_error_against_float64_by_arithmetic measures both sides)
        e = assert_elementwise(out[k], ref[k], f"{config} trained-like {k}", rtol=1e-4, floor=1e-4)
        print(f"{config} trained-like {k}: max |diff| / max |ref| = {e:.2e}")
        assert e < 1e-4, k
    anchors = T(pp.generate_anchor_box())
    boxes, scores = pp.post_process({"ego": {"transformation_matrix": torch.eye(4), "anchor_box": anchors}}, {"ego": out})
    rb, rs, info = oracle.post_process([{k: v.cpu() for k, v in out.items()}], anchors, h["postprocess"])
    assert pp.last_counts["candidates"] == len(info["cand_index"]) > 100
    assert boxes.shape == rb.shape and rb.shape[0] > 20
    np.testing.assert_allclose(scores.cpu().numpy(), rs.numpy(), rtol=3e-7, atol=0)
    np.testing.assert_allclose(boxes.cpu().numpy(), rb.numpy(), rtol=2e-6, atol=2e-5)


# ------------------------------------------------------------------------------------------------ SplitMaps and the producer-split convolution
def round22(x):
    return ((x.contiguous().view(torch.int32) + 2) & -4).view(torch.float32)


def assert_split_map_holds(sm, want, what=""):
    """A SplitMap holds `want` rounded to 22 significant bits: exactly for |value| >= 2^-13, to an absolute 2^-33 below (csrc/common.h)."""
    got, w22 = sm.dense(), round22(want)
    big = want.abs() >= 2.0 ** -13
    assert torch.equal(got[big], w22[big]), what
    if bool((~big).any()):
        assert float((got[~big] - want[~big]).abs().max()) <= 2.0 ** -33, what


def test_split_map_pack_unpack_and_layout():
    """unpack(pack(x)) = x rounded to 22 significant bits (ties away), from NCHW and channels-last inputs, to both output layouts; the unpack kernel agrees
    with the layout's definition evaluated in torch (SplitMap.dense_reference); re-packing a packed map reproduces it bit for bit (pairs are canonical)."""
    g = torch.Generator(device=DEV).manual_seed(1)
    for shape in ((2, 32, 7, 13), (1, 16, 1, 1), (3, 64, 25, 88)):
        x = torch.randn(shape, generator=g, device=DEV) * torch.logspace(-1, 3, shape[1], device=DEV).view(1, -1, 1, 1)
        x = torch.where(x.abs() < 2.0 ** -12, torch.full_like(x, 2.0 ** -12), x)      # (pairs are exact from 2^-13 up; smaller values keep an absolute 2^-34)
        x[0, 0, 0, 0] = 0.0
        want = round22(x)
        for xin in (x, x.contiguous(memory_format=torch.channels_last)):
            sm = ops.SplitMap.pack(xin)
            assert sm.shape == tuple(shape)
            assert torch.equal(sm.dense(), want) and torch.equal(sm.dense(channels_last=True), want)
            assert torch.equal(sm.dense_reference(), want)
            assert torch.equal(ops.SplitMap.pack(sm.dense()).data, sm.data)
        assert float((want - x).abs().max() / x.abs().max()) < 2.0 ** -22
    assert not ops.sp_range_exceeded(DEV)
    ops.SplitMap.pack(torch.full((1, 16, 2, 2), 7.0e4, device=DEV))
    assert ops.sp_range_exceeded(DEV) and not ops.sp_range_exceeded(DEV)          # reported once, then cleared


SP_SHAPES = [(1, 16, 64, 4, 4), (2, 16, 64, 5, 6), (1, 32, 128, 6, 35), (3, 16, 64, 9, 18), (2, 48, 64, 25, 88), (2, 64, 64, 16, 33), (1, 16, 192, 1, 1),
             (5, 32, 64, 7, 3), (3, 32, 64, 26, 40), (2, 32, 64, 50, 48)]


@pytest.mark.parametrize("geometry", [81, 121, 124, 148, 326])      # (326, round 6: 8 x 48 tiles of 32 output channels, 12 wavefronts)
@pytest.mark.parametrize("shape", SP_SHAPES)
def test_conv3x3_sp_equals_consumer_split_kernel_bit_for_bit(shape, geometry):
    """The producer-split kernel (csrc/conv3x3_sp.hip) against the fp16 mode of csrc/conv3x3_emu.hip on the same 22-bit inputs: identical bits, for every
    tile geometry, batches tiled as one tall image (boundaries inside tiles) and per image, ragged widths, no / SplitMap / channels-last float32 residual,
    SplitMap and channels-last float32 outputs, with and without ReLU (resblock.py:53-69)."""
    N, Ci, Co, H, W = shape
    g = torch.Generator(device=DEV).manual_seed(sum(shape) + geometry)
    x = round22(torch.randn((N, Ci, H, W), generator=g, device=DEV))
    w = torch.randn((Co, Ci, 3, 3), generator=g, device=DEV) / (9 * Ci) ** 0.5
    b = torch.randn(Co, generator=g, device=DEV)
    rs = ops.SplitMap.pack(torch.randn((N, Co, H, W), generator=g, device=DEV))
    r = rs.dense()                                     # (what the pairs hold: values below 2^-13 are not exactly the 22-bit rounding)
    w16 = ops.pack_conv3x3_emu_weight(w, 16, True)
    xs = ops.SplitMap.pack(x)
    for res_kind, relu in (("none", True), ("split", True), ("nhwc", False)):
        res_old = None if res_kind == "none" else r
        res_new = None if res_kind == "none" else rs if res_kind == "split" else r.contiguous(memory_format=torch.channels_last)
        want = ops.conv3x3_emu_bias_act(x, w16, b, Co, res_old, relu, 16)
        got_cl = ops.conv3x3_sp(xs, w16, b, Co, res_new, relu, out_split=False, geometry=100000 + geometry)        # (100000: whole tiles, no stream-K cut)
        assert got_cl.shape == want.shape and torch.equal(got_cl, want), (shape, geometry, res_kind, float((got_cl - want).abs().max()))
        got_sp = ops.conv3x3_sp(xs, w16, b, Co, res_new, relu, out_split=True, geometry=100000 + geometry)
        assert_split_map_holds(got_sp, want, (shape, geometry, res_kind))
        if geometry in (81, 148) and Ci >= 32:                  # the same launch with its tiles cut between workgroups (stream-K): another summation order, same sums
            cut = ops.conv3x3_sp(xs, w16, b, Co, res_new, relu, out_split=False, geometry=200000 + geometry)
            assert float((cut - want).abs().max()) <= 2e-6 * float(want.abs().max()), (shape, geometry, res_kind)
            assert torch.equal(cut, ops.conv3x3_sp(xs, w16, b, Co, res_new, relu, out_split=False, geometry=200000 + geometry))
    assert not ops.sp_range_exceeded(DEV)


@pytest.mark.parametrize("shape", [(5, 64, 64, 100, 352), (5, 128, 128, 50, 176), (5, 256, 256, 25, 88), (1, 256, 256, 100, 352), (2, 64, 64, 100, 252), (2, 256, 256, 25, 63)])
def test_conv3x3_sp_backbone_shapes_bit_equal_and_against_float64(shape):
    """The stride-1 backbone shapes with the geometry the product picks: bit-equal to the consumer-split kernel, error against float64 <= 2e-6 of the scale."""
    N, Ci, Co, H, W = shape
    g = torch.Generator(device=DEV).manual_seed(sum(shape))
    x = round22(torch.relu(torch.randn((N, Ci, H, W), generator=g, device=DEV)))
    w = torch.randn((Co, Ci, 3, 3), generator=g, device=DEV) / (9 * Ci) ** 0.5
    b = torch.randn(Co, generator=g, device=DEV)
    rs = ops.SplitMap.pack(torch.randn((N, Co, H, W), generator=g, device=DEV))
    r = rs.dense()                                     # (what the pairs hold: values below 2^-13 are not exactly the 22-bit rounding)
    w16 = ops.pack_conv3x3_emu_weight(w, 16, True)
    xs = ops.SplitMap.pack(x)
    want = ops.conv3x3_emu_bias_act(x, w16, b, Co, r, True, 16)
    got = ops.conv3x3_sp(xs, w16, b, Co, rs, True, out_split=False)
    ref = conv64(x, w, b, r)
    scale = float(ref.abs().max())
    whole = ops.conv3x3_sp(xs, w16, b, Co, rs, True, out_split=False, geometry=100000)
    if shape != (1, 256, 256, 100, 352):                          # (there the consumer-split kernel itself hands long tiles over between workgroups: another summation order)
        assert torch.equal(whole, want)
    assert float((whole - want).abs().max()) / scale < 1e-6
    if ops.conv3x3_sp_is_split(N, Ci, Co, H, W):                  # the product's launch cuts this shape's tiles (stream-K): same sums, another order, deterministic
        assert float((got - whole).abs().max()) / scale < 1e-6
        assert torch.equal(got, ops.conv3x3_sp(xs, w16, b, Co, rs, True, out_split=False))
    else:
        assert torch.equal(got, whole)
    assert float((got.double() - ref).abs().max()) / scale < 2e-6
    got_sp = ops.conv3x3_sp(xs, w16, b, Co, rs, True, out_split=True)
    assert_split_map_holds(got_sp, got)


def test_consumer_split_kernels_write_split_maps():
    """The layers in front of a SplitMap chain: the strided kernels (NCHW / channels-last input) and the tap-major stride-1 kernel on an NCHW input with
    ``out_split``: the SplitMap holds exactly the float32 output of the same kernel rounded to 22 bits."""
    g = torch.Generator(device=DEV).manual_seed(11)
    for (N, Ci, Co, H, W) in ((2, 64, 64, 20, 70), (5, 64, 128, 100, 352), (2, 128, 256, 50, 126)):
        x = torch.randn((N, Ci, H, W), generator=g, device=DEV)
        w = torch.randn((Co, Ci, 3, 3), generator=g, device=DEV) / (9 * Ci) ** 0.5
        b = torch.randn(Co, generator=g, device=DEV)
        ws = ops.pack_conv3x3_emu_weight(w, 16, False)
        for xin in (x, x.contiguous(memory_format=torch.channels_last)):
            want = ops.conv3x3_emu_bias_act(xin, ws, b, Co, None, True, 16, stride=2)
            got = ops.conv3x3_emu_bias_act(xin, ws, b, Co, None, True, 16, stride=2, out_split=True)
            assert isinstance(got, ops.SplitMap) and got.shape == tuple(want.shape)
            assert_split_map_holds(got, want, (N, Ci, Co, H, W))
    for (N, Ci, Co, H, W) in ((1, 384, 256, 100, 352), (2, 32, 64, 9, 40), (5, 128, 128, 50, 176), (5, 256, 256, 25, 88)):
        x = torch.randn((N, Ci, H, W), generator=g, device=DEV)
        w = torch.randn((Co, Ci, 3, 3), generator=g, device=DEV) / (9 * Ci) ** 0.5
        b = torch.randn(Co, generator=g, device=DEV)
        wt = ops.pack_conv3x3_emu_weight(w, 16, True)
        want = ops.conv3x3_emu_bias_act(x, wt, b, Co, None, True, 16)      # (NCHW and SplitMap outputs share the whole-tile / stream-K choice: same summation order)
        got = ops.conv3x3_emu_bias_act(x, wt, b, Co, None, True, 16, out_split=True)
        assert_split_map_holds(got, want, (N, Ci, Co, H, W))


def test_split_map_route_equals_consumer_split_route_on_the_model():
    """The whole detector on the SplitMap route (default) against the same model with COALIGN_SPLIT_MAPS off: the stage outputs differ only by the 22-bit
    rounding of the intermediate maps (the consumer-split kernels round the same values when they read them, so the heads agree to ~1e-6), and both meet
    the oracle bound."""
    from coalign_amd import backbone as bb_mod
    h = builtin_config("opv2v_coalign")
    model = build_model(h)
    from coalign_amd.synthetic import fill_parameters_
    fill_parameters_(model, seed=0, cls_bias=-1.5)
    model = model.to(DEV).eval()
    fd = to_device(make_frame(h, 2, pillars_per_agent=6000, seed=77, noise=(0.2, 0.2)), DEV)
    assert bb_mod.split_maps_active()
    with torch.no_grad():
        out_sp = model(fd)
        saved = bb_mod.SPLIT_MAPS
        try:
            bb_mod.SPLIT_MAPS = False
            out_cs = model(fd)
        finally:
            bb_mod.SPLIT_MAPS = saved
    for k in ("cls_preds", "reg_preds", "dir_preds"):
        e = rel_err(out_sp[k], out_cs[k])
        print(f"\nsplit-map route vs consumer-split route {k}: {e:.2e}", end="")
        assert e < 2e-5, k
    assert not ops.sp_range_exceeded(DEV)


def test_empty_frame_and_empty_agent_on_the_sparse_canvas_route():
    """a frame (or a rank's block of agents: bench.py --mode gather with more ranks than agents) without a single pillar goes through the default
    sparse-canvas route -- every stamp stale, the maps are what the biases make of an all-zero canvas -- and equals the dense-canvas route."""
    from coalign_amd import detector as det
    from coalign_amd.synthetic import fill_parameters_
    h = builtin_config("mini_coalign")
    model = build_model(h)
    fill_parameters_(model, seed=0)
    model = model.to(DEV).eval()
    full = to_device(make_frame(h, 2, pillars_per_agent=60, seed=9, noise=(0.2, 0.2)), DEV)
    pl = full["processed_lidar"]
    only_ego = pl["voxel_coords"][:, 0] == 0
    cases = {"empty frame": {k: v[:0] for k, v in pl.items()}, "agent 1 empty": {k: v[only_ego] for k, v in pl.items()}}
    for name, sub in cases.items():
        fr = dict(full, processed_lidar=sub)
        with torch.no_grad():
            out = model(fr)
            saved = det.SPARSE_CANVAS
            try:
                det.SPARSE_CANVAS = False
                ref = model(fr)
            finally:
                det.SPARSE_CANVAS = saved
        for k in out:
            assert torch.isfinite(out[k]).all() and rel_err(out[k], ref[k]) < 1e-5, (name, k)


def test_conv3x3_sp_stream_k_hand_over_stress():
    """The stream-K hand-over (partial sums through L2-bypassing stores / loads, flags the consumers reset) under load: hundreds of launches on two streams
    at once, each with its own workspace, every result bit-equal to the first; the flag words are zero again after every launch."""
    shapes = [(5, 256, 256, 25, 88), (2, 128, 128, 30, 50)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    cases = []
    for (N, Ci, Co, H, W), st in zip(shapes, streams):
        g = torch.Generator(device=DEV).manual_seed(H)
        x = ops.SplitMap.pack(torch.relu(torch.randn((N, Ci, H, W), generator=g, device=DEV)))
        w16 = ops.pack_conv3x3_emu_weight(torch.randn((Co, Ci, 3, 3), generator=g, device=DEV) / (9 * Ci) ** 0.5, 16, True)
        b = torch.randn(Co, generator=g, device=DEV)
        geo = 200000 + (148 if H == 25 else 81)
        assert ops.conv3x3_sp_is_split(N, Ci, Co, H, W, geo)
        cases.append((x, w16, b, Co, geo, st))
    torch.cuda.synchronize()
    first, outs = [], [[], []]
    for i, (x, w16, b, Co, geo, st) in enumerate(cases):
        with torch.cuda.stream(st):
            first.append(ops.conv3x3_sp(x, w16, b, Co, None, True, out_split=False, geometry=geo))
    torch.cuda.synchronize()
    for rep in range(150):
        for i, (x, w16, b, Co, geo, st) in enumerate(cases):
            with torch.cuda.stream(st):
                outs[i].append(ops.conv3x3_sp(x, w16, b, Co, None, True, out_split=False, geometry=geo))
        if rep % 50 == 49:
            torch.cuda.synchronize()
            for i in range(2):
                assert all(torch.equal(o, first[i]) for o in outs[i]), (rep, i)
                outs[i].clear()
    for ws in ops._SP_WS.values():
        assert int(ws[:2048].view(torch.int32).abs().sum()) == 0          # every flag consumed and reset


# ------------------------------------------------------------------------------------------------ frames read in place (include/coalign_amd.h (1c))
def _record_case(M, cap, seed):
    from coalign_amd.config import builtin_config
    from coalign_amd.synthetic import make_frame
    h = builtin_config("opv2v_coalign")
    margs = h["model"]["args"]
    fr = make_frame(h, 2, pillars_per_agent=max((M + 1) // 2, 1), seed=seed)["processed_lidar"]
    pl = {k: v[:M].to(DEV) for k, v in fr.items()}
    g = torch.Generator().manual_seed(seed)
    w = (torch.randn(64, 10, generator=g) * 0.3).to(DEV)
    bn = tuple(t.to(DEV) for t in (torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.1, torch.randn(64, generator=g) * 0.1, torch.rand(64, generator=g) + 0.5))
    return margs, pl, w, bn


@pytest.mark.parametrize("M,cap", [(5000, 5000), (3001, 4096), (1, 4096), (0, 4096)])
def test_pillar_frame_record_equals_direct_launch(M, cap):
    """coalign_pillar_encode_sparse_frame reads (arrays, count) from a device record: same feature rows and the same cells as the direct launch, for a count below
    the capacity, a single pillar and an empty frame; a second frame at OTHER addresses through the same record words (what a replayed graph does)."""
    margs, pl, w, bn = _record_case(M, cap, 11)
    args = (w, None, bn, 1e-3, True, margs["voxel_size"], margs["lidar_range"][:3], 2, 200, 704)
    want = ops.pillar_encode_sparse(pl["voxel_features"], pl["voxel_num_points"], pl["voxel_coords"], *args, canvas_cache={})
    blob, host = torch.zeros(4, dtype=torch.int64, device=DEV), torch.zeros(4, dtype=torch.int64).pin_memory()
    rec = ops.PillarFrameRecord(blob, host, cap)
    assert ops.PillarFrameRecord.admits(pl["voxel_features"], pl["voxel_num_points"], pl["voxel_coords"], torch.device(DEV))
    cache = {}
    for rnd in range(2):
        cur = pl if rnd == 0 else {k: v.clone() for k, v in pl.items()}
        rec.set(cur["voxel_features"], cur["voxel_num_points"], cur["voxel_coords"])
        blob.copy_(host, non_blocking=True)
        got = ops.pillar_encode_sparse(pl["voxel_features"][:0] if rnd else pl["voxel_features"], pl["voxel_num_points"], pl["voxel_coords"], *args, canvas_cache=cache, frame=rec)
        torch.cuda.synchronize()
        assert got.feats.shape[0] == cap and torch.equal(got.feats[:M], want.feats)
        tag_g, tag_w = int(got.state[0]), int(want.state[0])
        live_g, live_w = (got.stamps >> 32) == tag_g, (want.stamps >> 32) == tag_w
        assert torch.equal(live_g, live_w) and torch.equal(got.stamps[live_g] & 0xffffffff, want.stamps[live_w] & 0xffffffff)
    with pytest.raises(ValueError):
        big = torch.zeros((cap + 1, 32, 4), device=DEV)
        rec.set(big, torch.zeros(cap + 1, dtype=torch.int32, device=DEV), torch.zeros((cap + 1, 4), dtype=torch.int32, device=DEV))
    assert not ops.PillarFrameRecord.admits(pl["voxel_features"], pl["voxel_num_points"].long(), pl["voxel_coords"], torch.device(DEV))


def _pipeline_world(n_frames=6):
    from coalign_amd.config import builtin_config
    from coalign_amd.detector import build_model, to_device
    from coalign_amd.postprocess import build_postprocessor
    from coalign_amd.synthetic import calibrate_heads_, fill_parameters_, make_frame
    h = builtin_config("opv2v_coalign")
    counts = (8000, 8000, 7300, 7800, 8000, 6900)[:n_frames]
    cpu = [make_frame(h, 5, pillars_per_agent=m, seed=700 + i, noise=(0.2, 0.2)) for i, m in enumerate(counts)]
    frames = []
    for f in cpu:
        d = to_device(f, DEV)
        d["record_len"] = [5]
        d["pairwise_t_matrix_host"] = f["pairwise_t_matrix"]
        frames.append(d)
    model = build_model(h)
    fill_parameters_(model, seed=0)
    model = model.to(DEV).eval()
    pp = build_postprocessor(h["postprocess"], False)
    calibrate_heads_(model, frames[0], pp.params["target_args"]["score_threshold"], 600)
    return model, pp, torch.from_numpy(pp.generate_anchor_box()), frames


def test_pipeline_frame_records_equal_copied_inputs_bit_for_bit():
    """FramePipeline(graph=True) on frames read in place (device record + host-normalised poses, ONE small transfer per frame) returns what the copying route of
    rounds 2-4 returns, bit for bit: equal shapes (exact capture), ragged counts (the capacity-sized bucket graph), with and without the host copy of the poses,
    and across recurrences of the same slot."""
    from coalign_amd import pipeline as pl_mod
    model, pp, anchors, frames = _pipeline_world()
    order = [0, 1, 2, 3, 4, 5, 1, 3, 0, 5, 2, 4]
    keep = pl_mod.FRAME_RECORDS
    try:
        pl_mod.FRAME_RECORDS = False
        ref_pipe = pl_mod.FramePipeline(model, pp, anchors, lanes=2, result_lag=1, graph=True)
        want = ref_pipe.run([frames[i] for i in order])
        assert all(s.record is None for d in ref_pipe._slots for s in d.values())
        ref_pipe.close()
        pl_mod.FRAME_RECORDS = True
        for host_pose in (True, False):
            fs = frames if host_pose else [{k: v for k, v in f.items() if k != "pairwise_t_matrix_host"} for f in frames]
            pipe = pl_mod.FramePipeline(model, pp, anchors, lanes=2, result_lag=1, graph=True)
            got = pipe.run([fs[i] for i in order])
            slots = [s for d in pipe._slots for s in d.values()]
            assert slots and all(s.record is not None and (s.affine_view is not None) == host_pose and not any(k.startswith("voxel") for k in s.inputs) for s in slots)
            assert any(s.capacity is not None for s in slots)                  # the ragged frames went through a bucket graph
            assert pipe.frames_in_place == len(order) and pipe.frames_copied == 0
            for i, ((b, s), (wb, ws)) in enumerate(zip(got, want)):
                assert wb is not None and torch.equal(b, wb) and torch.equal(s, ws), f"frame {i} (host poses {host_pose})"
            pipe.close()
    finally:
        pl_mod.FRAME_RECORDS = keep


def test_pipeline_falls_back_to_copies_when_the_encoder_takes_no_record():
    """A model configured off the sparse-canvas route (COALIGN_SPARSE_CANVAS=0: the dense persistent canvas of round 3) raises FrameRecordUnsupported in the
    warm-up; the pipeline then copies frames into the graph's buffers, for good, and the detections equal the synchronous path's."""
    from coalign_amd import detector as det
    from coalign_amd import pipeline as pl_mod
    model, pp, anchors, frames = _pipeline_world(3)
    saved = det.SPARSE_CANVAS
    try:
        det.SPARSE_CANVAS = False
        meta = {"ego": {"transformation_matrix": torch.eye(4, device=DEV), "anchor_box": anchors}}
        with torch.no_grad():
            want = [pp.post_process(meta, {"ego": model(f)}) for f in frames]
        pipe = pl_mod.FramePipeline(model, pp, anchors, lanes=2, result_lag=1, graph=True)
        assert pipe._records_ok
        got = pipe.run(frames)
        assert not pipe._records_ok and pipe.frames_in_place == 0 and pipe.frames_copied == len(frames) and all(s.record is None for d in pipe._slots for s in d.values())
        for (b, s), (wb, ws) in zip(got, want):
            assert torch.equal(b, wb) and torch.equal(s, ws)
        pipe.close()
    finally:
        det.SPARSE_CANVAS = saved


# ------------------------------------------------------------------------------------------------ up-sampling heads writing SplitMaps (include/coalign_amd.h (10c))
@pytest.mark.parametrize("up,cin,hw,n,nhwc", [(1, 64, (100, 352), 1, True), (2, 128, (50, 176), 1, True), (4, 256, (25, 88), 1, True),
                                               (1, 64, (13, 37), 2, False), (2, 128, (9, 21), 2, False), (4, 256, (7, 11), 3, True), (2, 64, (5, 40), 1, False)])
def test_pointwise_heads_write_split_maps_bit_for_bit(up, cin, hw, n, nhwc):
    """coalign_pointwise_conv_emu_sp: a head's channel slice of the concatenated SplitMap holds exactly coalign_sp_pack of the float32 layer's output; the other
    slices are untouched; full-size head shapes of the bench and ragged small ones (pixel counts off the 32-pixel tiles, batch > 1, both input layouts)."""
    g = torch.Generator().manual_seed(31 * up + cin)
    H, W = hw
    cout, ctot, c_off = 128, 384, {1: 0, 2: 128, 4: 256}[up]
    x = torch.randn(n, cin, H, W, generator=g).to(DEV)
    if nhwc:
        x = x.contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cin, cout, up, up, generator=g) / math.sqrt(cin)).to(DEV)
    b = torch.randn(cout, generator=g).to(DEV)
    im = ops.pack_pointwise_emu_weight(ops.pack_pointwise_weight(w, True))
    ref = torch.zeros(n, ctot, H * up, W * up, device=DEV)
    ops.pointwise_conv(x, im, b, cout, up=up, relu=True, out=ref, c_off=c_off)
    want = ops.SplitMap.pack(ref)
    got = ops.SplitMap(torch.zeros_like(want.data))
    out = ops.pointwise_conv(x, im, b, cout, up=up, relu=True, out=got, c_off=c_off)
    torch.cuda.synchronize()
    assert out is got and float(ref.abs().max()) > 0.5
    assert torch.equal(got.data.view(torch.int16), want.data.view(torch.int16))
    with pytest.raises(ValueError):
        ops.pointwise_conv(x, im, b, cout, up=up, relu=True, out=got, c_off=8)


def test_shrink_header_on_split_maps_from_the_heads_equals_float_route():
    """Detector tail (up-sampling heads -> shrink header -> 1x1 heads) with the heads writing the concatenated map as a SplitMap and BOTH shrink convolutions on
    conv3x3_sp, against the route of the first half of round 5 (float32 concatenation, consumer-split kernel for the first convolution): the same 22-bit inputs,
    the same products; only the stream-K cut of the first convolution (hence its summation order) differs."""
    from coalign_amd import backbone as bb
    h = builtin_config("opv2v_coalign")
    model = build_model(h)
    fill_parameters_trained_like_(model, seed=5)
    model = model.to(DEV).eval()
    g = torch.Generator().manual_seed(77)
    fused = [torch.relu(torch.randn(1, c, hh, ww, generator=g)).to(DEV).contiguous(memory_format=torch.channels_last) for c, hh, ww in ((64, 100, 352), (128, 50, 176), (256, 25, 88))]
    with torch.no_grad():
        assert model.shrink_flag and model.shrink_conv.takes_split_maps()
        sm = model.backbone.decode_multiscale_feature(fused, out_split=True)
        assert isinstance(sm, ops.SplitMap) and sm.shape == (1, 384, 100, 352)
        dense = model.backbone.decode_multiscale_feature(fused)
        assert torch.equal(sm.data.view(torch.int16), ops.SplitMap.pack(dense).data.view(torch.int16))
        y_new = model.shrink_conv(sm)
        y_old = model.shrink_conv(dense)
        torch.cuda.synchronize()
    scale = float(y_old.abs().max())
    assert scale > 0 and float((y_new - y_old).abs().max()) <= 2e-6 * scale


def test_pipeline_queue_depth_equals_synchronous_path_bit_for_bit():
    """FramePipeline(lanes=2, queue_depth=3): six pipeline lanes on two HIP streams (a stream's next frames are enqueued while its current one runs; the host only waits for
    the oldest frame) return the synchronous path's detections bit for bit, in order, for equal and ragged pillar counts, with every result-lag setting, and through the
    bench's phases (submit bursts separated by drains from an idle GPU)."""
    from coalign_amd import pipeline as pl_mod
    model, pp, anchors, frames = _pipeline_world()
    meta = {"ego": {"transformation_matrix": torch.eye(4, device=DEV), "anchor_box": anchors}}
    with torch.no_grad():
        want = [pp.post_process(meta, {"ego": model(f)}) for f in frames]
    torch.cuda.synchronize()
    for lag in (5, 2, 0):
        pipe = pl_mod.FramePipeline(model, pp, anchors, lanes=2, queue_depth=3, result_lag=lag, graph=True)
        assert pipe.n_lanes == 6 and pipe.n_streams == 2 and len({s.cuda_stream for s in pipe.streams}) == 2 and pipe.streams[0] is pipe.streams[2] is pipe.streams[4]
        order, got = [], []
        for phase in ([0, 1, 2, 3, 4, 5], [i % 6 for i in range(13)], [5, 4, 3, 2, 1, 0, 0, 1, 2]):
            for i in phase:
                got += pipe.submit(frames[i])
            got += pipe.drain()
            torch.cuda.synchronize()
            order += phase
        assert [idx for idx, _, _ in got] == list(range(len(order)))
        for i, (_, b, s) in zip(order, got):
            assert torch.equal(b, want[i][0]) and torch.equal(s, want[i][1]), f"frame {i}, result lag {lag}"
        assert pipe.frames_in_place == len(order)
        pipe.close()
