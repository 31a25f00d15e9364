"""GPU tests of the strided SplitMap convolution (round 6; csrc/conv3x3_sp_s2.hip, include/coalign_amd.h (9f)), all through the C ABI.

The layer is the stride-2 first convolution of a ResNet stage (opencood/models/sub_modules/resblock.py:53-69 with stride 2, :150-174).  A stride-2 / pad-1 3x3
convolution's output (y, x) is the stride-1 / pad-1 convolution's output (2 y, 2 x), and ``coalign_conv3x3_sp_s2`` runs the SAME operations in the SAME order as
``coalign_conv3x3_sp``: the two must agree bit for bit on the same SplitMap -- and, through it, with the float64 yardstick of tests/test_round5_gpu.py.
"""
import pytest
import torch

from coalign_amd import backbone, ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

# (N, Cin, Cout, H, W): the three strided layers of the 5-agent OPV2V workload (the first reads the sparse canvas in the product), DAIR's, and ragged shapes
S2_SHAPES = [(5, 64, 64, 200, 704), (5, 64, 128, 100, 352), (5, 128, 256, 50, 176), (2, 64, 128, 100, 252), (2, 32, 64, 37, 45), (1, 16, 64, 9, 70), (3, 48, 192, 8, 8), (1, 64, 64, 1, 1),
             (2, 64, 64, 5, 131)]


def conv64(x, w, b, stride):
    """relu(conv3x3(x, w, stride, pad 1) + b) in float64 as nine matrix products."""
    N, Ci, H, W = x.shape
    xp = torch.nn.functional.pad(x.double(), (1, 1, 1, 1))
    Ho, Wo = (H + stride - 1) // stride, (W + stride - 1) // stride
    out = torch.zeros((N, w.shape[0], Ho, Wo), dtype=torch.float64, device=x.device)
    for dy in range(3):
        for dx in range(3):
            out += torch.einsum("oc,nchw->nohw", w.double()[:, :, dy, dx], xp[:, :, dy:dy + stride * Ho:stride, dx:dx + stride * Wo:stride])
    return torch.relu(out + b.double().view(1, -1, 1, 1))


def _case(shape, seed, relu=True):
    N, Ci, Co, H, W = shape
    g = torch.Generator(device=DEV).manual_seed(seed)
    x = torch.randn((N, Ci, H, W), generator=g, device=DEV)
    w = torch.randn((Co, Ci, 3, 3), generator=g, device=DEV) * (1.0 / (9 * Ci) ** 0.5)
    b = torch.randn(Co, generator=g, device=DEV) * 0.1
    return x, w, b


@pytest.mark.parametrize("shape", S2_SHAPES)
@pytest.mark.parametrize("relu", [True, False])
def test_strided_split_convolution_equals_the_stride_one_kernel_at_even_pixels(shape, relu):
    x, w, b = _case(shape, 7)
    img = ops.pack_conv3x3_emu_weight(w, 16, tap_major=True)
    xs = ops.SplitMap.pack(x)
    want = ops.conv3x3_sp(xs, img, b, shape[2], None, relu, out_split=True).dense()[:, :, ::2, ::2]
    got = ops.conv3x3_sp_s2(xs, img, b, shape[2], relu)
    assert got.shape == tuple(want.shape)
    assert torch.equal(got.dense(), want)
    assert torch.equal(got.dense(), got.dense_reference())


def test_strided_split_convolution_against_float64():
    shape = (5, 64, 128, 100, 352)
    x, w, b = _case(shape, 3)
    img = ops.pack_conv3x3_emu_weight(w, 16, tap_major=True)
    got = ops.conv3x3_sp_s2(ops.SplitMap.pack(x), img, b, shape[2], True).dense().double()
    ref = conv64(x, w, b, 2)
    err = float((got - ref).abs().max()) / float(ref.abs().max())
    print(f"\nconv3x3_sp_s2 vs float64: {err:.2e} of the scale")
    assert err <= 2e-6


def _sparse_canvas(n_agents, ny, nx, pillars, seed, count_below_capacity=False):
    """A SparseCanvas from the one-launch pillar op on random pillars (duplicate cells included: the larger row wins)."""
    from coalign_amd.config import builtin_config
    from coalign_amd.detector import build_model
    from coalign_amd.synthetic import fill_parameters_, make_frame
    h = builtin_config("opv2v_coalign")
    model = build_model(h)
    fill_parameters_(model, seed=seed)
    model = model.to(DEV).eval()
    margs = h["model"]["args"]
    pl = make_frame(h, n_agents, pillars_per_agent=pillars, seed=seed)["processed_lidar"]
    pfn = model.pillar_vfe.pfn_layers[0]
    bn = (pfn.norm.weight, pfn.norm.bias, pfn.norm.running_mean, pfn.norm.running_var)
    gx, gy, _ = [int(v) for v in margs["point_pillar_scatter"]["grid_size"]]
    assert (gy, gx) == (ny, nx)
    count_dev = None
    vf, npts, coords = pl["voxel_features"].to(DEV), pl["voxel_num_points"].to(DEV), pl["voxel_coords"].to(DEV)
    if count_below_capacity:
        count_dev = torch.tensor([vf.shape[0] - 1234], dtype=torch.int32, device=DEV)
    return ops.pillar_encode_sparse(vf, npts, coords, pfn.linear.weight, None, bn, 1e-3, True, margs["voxel_size"], margs["lidar_range"][:3], n_agents, ny, nx, canvas_cache={},
                                    count_dev=count_dev)


@pytest.mark.parametrize("pillars,below", [(8000, False), (3000, True), (40000, False)])
def test_strided_split_convolution_reads_the_sparse_canvas(pillars, below):
    """The LDS-DMA gather through the cell stamps: bit-equal to the dense route on the densified canvas (zeros where no pillar lives)."""
    sc = _sparse_canvas(3, 200, 704, pillars, 11, below)
    g = torch.Generator(device=DEV).manual_seed(5)
    w = torch.randn((64, 64, 3, 3), generator=g, device=DEV) * (1.0 / 24.0)
    b = torch.randn(64, generator=g, device=DEV) * 0.1
    img = ops.pack_conv3x3_emu_weight(w, 16, tap_major=True)
    dense = ops.SplitMap.pack(sc.dense())
    want = ops.conv3x3_sp_s2(dense, img, b, 64, True)
    got = ops.conv3x3_sp_s2(sc, img, b, 64, True)
    assert torch.equal(got.data, want.data)
    ref = ops.conv3x3_sp(dense, img, b, 64, None, True, out_split=True).dense()[:, :, ::2, ::2]
    assert torch.equal(got.dense(), ref)
    assert float(got.dense().abs().max()) > 0


def test_sp_pack_rows_is_the_split_of_the_rows():
    sc = _sparse_canvas(2, 200, 704, 5000, 3)
    rows = ops.sp_pack_rows(sc)
    M, C = sc.feats.shape
    d = rows.float()                                           # [M, C/16, 4, 8]
    v = (d[:, :, 0::2] + d[:, :, 1::2] / 1024.0).reshape(M, C)  # planes (2 * half + term) -> [M, C/16, 2 halves, 8]
    r = ((sc.feats.contiguous().view(torch.int32) + 2) & -4).view(torch.float32)      # 22 significant bits, ties away from zero
    assert torch.equal(v, r)


def test_model_on_the_strided_split_route_stays_within_the_suite_tolerance():
    """Whole multiscale backbone, S2_SPLIT = all vs 0: the strided layers change their summation order (16-channel intervals of nine taps instead of tap pairs of
    8 channels), nothing else: the stage outputs agree to float32 rounding."""
    from coalign_amd.config import builtin_config
    from coalign_amd.detector import build_model
    from coalign_amd.synthetic import fill_parameters_
    h = builtin_config("opv2v_coalign")
    model = build_model(h)
    fill_parameters_(model, seed=0)
    model = model.to(DEV).eval()
    sc = _sparse_canvas(3, 200, 704, 6000, 11)
    saved = backbone.S2_SPLIT
    try:
        outs = {}
        for mode in ("0", "sparse", "all"):
            backbone.S2_SPLIT = mode
            with torch.no_grad():
                outs[mode] = [f.clone() for f in model.backbone.get_multiscale_feature(sc)]
    finally:
        backbone.S2_SPLIT = saved
    for mode in ("sparse", "all"):
        for a, b in zip(outs[mode], outs["0"]):
            e = float((a - b).abs().max()) / float(b.abs().max())
            print(f"\nS2_SPLIT={mode}: {tuple(a.shape)} {e:.2e} of the scale")
            assert e <= 2e-6


@pytest.mark.parametrize("shape", [(5, 256, 256, 25, 88), (2, 256, 256, 25, 88), (2, 256, 256, 25, 63), (1, 64, 128, 19, 50)])
def test_32_channel_tiles_are_bit_equal_to_the_64_channel_geometries_and_both_outputs_agree(shape):
    """Round 6: ``coalign_conv3x3_sp`` with 8 x 48 tiles of 32 output channels (geometry 326: the 25 x 88 maps fill 256 CUs) -- same sums in the same order as
    the 64-channel geometries; and ``coalign_conv3x3_sp_both``: the SplitMap beside the channels-last map is ``coalign_sp_pack`` of it, bit for bit."""
    N, Ci, Co, H, W = shape
    g = torch.Generator(device=DEV).manual_seed(sum(shape))
    x = ops.SplitMap.pack(torch.randn((N, Ci, H, W), generator=g, device=DEV))
    w = ops.pack_conv3x3_emu_weight(torch.randn((Co, Ci, 3, 3), generator=g, device=DEV) / (9 * Ci) ** 0.5, 16, True)
    b = torch.randn(Co, generator=g, device=DEV)
    rs = ops.SplitMap.pack(torch.randn((N, Co, H, W), generator=g, device=DEV))
    for res in (None, rs, rs.dense(channels_last=True)):
        want = ops.conv3x3_sp(x, w, b, Co, res, True, out_split=True, geometry=100148)
        want_cl = ops.conv3x3_sp(x, w, b, Co, res, True, out_split=False, geometry=100148)      # (float32: not rounded to the pairs' 22 bits)
        for geo in (100326, 0):
            got = ops.conv3x3_sp(x, w, b, Co, res, True, out_split=True, geometry=geo)
            assert torch.equal(got.data, want.data), (shape, geo)
            cl = ops.conv3x3_sp(x, w, b, Co, res, True, out_split=False, geometry=geo)
            assert torch.equal(cl, want_cl), (shape, geo)
            y, ysp = ops.conv3x3_sp(x, w, b, Co, res, True, geometry=geo, out_both=True)
            assert torch.equal(y, cl) and torch.equal(ysp.data, ops.SplitMap.pack(y).data) and torch.equal(ysp.data, want.data), (shape, geo)


@pytest.mark.parametrize("N,hw", [(1, (100, 352)), (2, (48, 72))])
def test_heads_in_one_launch_equal_one_launch_per_scale(N, hw):
    """Round 6: ``coalign_pointwise_conv_emu_sp_multi`` -- the three up-sampling heads (base_bev_backbone_resnet.py:121-138) as one launch write the same SplitMap,
    bit for bit, as one ``coalign_pointwise_conv_emu_sp`` launch per scale."""
    from coalign_amd.backbone import PointwisePack
    H, W = hw
    g = torch.Generator(device=DEV).manual_seed(N + H)
    layers, c_off = [], 0
    for cin, up in ((64, 1), (128, 2), (256, 4)):
        x = torch.relu(torch.randn((N, cin, H // up, W // up), generator=g, device=DEV)).contiguous(memory_format=torch.channels_last)
        wt = torch.randn((cin, 128, up, up), generator=g, device=DEV) / cin ** 0.5            # ConvTranspose2d weight [Cin, Cout, k, k]
        b = torch.randn(128, generator=g, device=DEV) * 0.1
        layers.append((x, PointwisePack(wt, True).get(), b, 128, up, c_off))
        c_off += 128
    want = ops.SplitMap.empty(N, c_off, H, W, DEV)
    for (x, im, b, cout, up, off) in layers:
        ops.pointwise_conv(x, im, b, cout, up=up, relu=True, out=want, c_off=off)
    got = ops.SplitMap.empty(N, c_off, H, W, DEV)
    got.data.fill_(float("nan"))
    ops.pointwise_heads_split(layers, got)
    assert torch.equal(got.data, want.data)
    assert float(got.dense().abs().max()) > 0


@pytest.mark.parametrize("shape", [(5, 64, 128, 100, 352), (5, 128, 256, 50, 176), (2, 32, 64, 37, 45), (1, 64, 64, 9, 70)])
def test_skip_convolution_as_a_tenth_tap(shape):
    """Round 6 (9g): ``coalign_conv3x3_sp_s2_skip`` -- conv1's SplitMap is that of the plain launch bit for bit; the skip map equals a float64 1 x 1 / stride-2
    convolution of the 22-bit input to 2e-6 of its scale (resblock.py:165-174: ``downsample``, no bias here, no ReLU)."""
    N, Ci, Co, H, W = shape
    x, w, b = _case(shape, 11)
    g = torch.Generator(device=DEV).manual_seed(1)
    wd = torch.randn((Co, Ci, 1, 1), generator=g, device=DEV) / Ci ** 0.5
    img = ops.pack_conv3x3_emu_weight(w, 16, tap_major=True)
    xs = ops.SplitMap.pack(x)
    want = ops.conv3x3_sp_s2(xs, img, b, Co, True)
    got, skip = ops.conv3x3_sp_s2(xs, img, b, Co, True, w_skip=ops.pack_conv1x1_sp_weight(wd))
    assert torch.equal(got.data, want.data)
    ref = torch.einsum("oc,nchw->nohw", wd[:, :, 0, 0].double(), xs.dense().double()[:, :, ::2, ::2])
    assert skip.shape == ref.shape and skip.permute(0, 2, 3, 1).is_contiguous()
    err = float((skip.double() - ref).abs().max()) / float(ref.abs().max())
    print(f"\nskip as a tenth tap vs float64: {err:.2e} of the scale")
    assert err <= 2e-6


def test_skip_as_a_tenth_tap_on_the_sparse_canvas_and_in_the_model():
    sc = _sparse_canvas(3, 200, 704, 6000, 11)
    g = torch.Generator(device=DEV).manual_seed(5)
    w = torch.randn((64, 64, 3, 3), generator=g, device=DEV) / 24.0
    wd = torch.randn((64, 64, 1, 1), generator=g, device=DEV) / 8.0
    b = torch.randn(64, generator=g, device=DEV) * 0.1
    img, simg = ops.pack_conv3x3_emu_weight(w, 16, tap_major=True), ops.pack_conv1x1_sp_weight(wd)
    dense = ops.SplitMap.pack(sc.dense())
    want, want_skip = ops.conv3x3_sp_s2(dense, img, b, 64, True, w_skip=simg)
    got, got_skip = ops.conv3x3_sp_s2(sc, img, b, 64, True, w_skip=simg)
    assert torch.equal(got.data, want.data) and torch.equal(got_skip, want_skip) and float(got_skip.abs().max()) > 0
    # the whole backbone with the skip fused against a pointwise launch of its own: the skip changes its arithmetic (22-bit pairs instead of three bf16 terms)
    from coalign_amd.config import builtin_config
    from coalign_amd.detector import build_model
    from coalign_amd.synthetic import fill_parameters_
    h = builtin_config("opv2v_coalign")
    model = build_model(h)
    fill_parameters_(model, seed=0)
    model = model.to(DEV).eval()
    saved, outs = backbone.S2_SKIP, {}
    try:
        for mode in (False, True):
            backbone.S2_SKIP = mode
            with torch.no_grad():
                outs[mode] = [f.clone() for f in model.backbone.get_multiscale_feature(sc)]
    finally:
        backbone.S2_SKIP = saved
    for a, c in zip(outs[True], outs[False]):
        e = float((a - c).abs().max()) / float(c.abs().max())
        print(f"\nS2_SKIP on / off: {tuple(a.shape)} {e:.2e} of the scale")
        assert e <= 2e-6


@pytest.mark.parametrize("shape", [(1, 256, 20, 100, 352), (2, 64, 32, 37, 45), (1, 128, 6, 9, 70)])
def test_merged_heads_on_a_split_map(shape):
    """Round 6 (10e): ``coalign_heads_sp`` -- the merged 1 x 1 heads (point_pillar_baseline_multiscale.py:123-133) on the shrink header's SplitMap against a float64
    product of the same 22-bit values: within 2e-6 of the scale."""
    N, Ci, M, H, W = shape
    g = torch.Generator(device=DEV).manual_seed(sum(shape))
    x = ops.SplitMap.pack(torch.relu(torch.randn((N, Ci, H, W), generator=g, device=DEV)))
    w = torch.randn((M, Ci, 1, 1), generator=g, device=DEV) / Ci ** 0.5
    w[M // 2] *= 1e-3                                            # a head channel three decades below the others: the per-row power-of-two scale
    b = torch.randn(M, generator=g, device=DEV)
    got = ops.heads_sp(x, ops.pack_heads_sp_weight(w), b, M)
    ref = torch.einsum("oc,nchw->nohw", w[:, :, 0, 0].double(), x.dense().double()) + b.double().view(1, -1, 1, 1)
    assert got.shape == ref.shape and got.is_contiguous()
    for m in range(M):
        err = float((got[:, m].double() - ref[:, m]).abs().max()) / float(ref[:, m].abs().max())
        assert err <= 2e-6, (m, err)


def test_model_with_heads_on_a_split_map_matches_the_pointwise_heads():
    """Whole detector, ``HEADS_SPLIT_IN`` on / off: the head outputs agree to 2e-6 of their scale (22-bit pairs against three bf16 terms on the float32 map)."""
    from coalign_amd.config import builtin_config
    from coalign_amd.detector import build_model, to_device
    from coalign_amd.synthetic import fill_parameters_, make_frame
    h = builtin_config("opv2v_coalign")
    model = build_model(h)
    fill_parameters_(model, seed=0)
    model = model.to(DEV).eval()
    fr = to_device(make_frame(h, 3, pillars_per_agent=5000, seed=5, noise=(0.2, 0.2)), DEV)
    saved, outs = backbone.HEADS_SPLIT_IN, {}
    try:
        for mode in (False, True):
            backbone.HEADS_SPLIT_IN = mode
            with torch.no_grad():
                outs[mode] = {k: v.clone() for k, v in model(fr).items()}
    finally:
        backbone.HEADS_SPLIT_IN = saved
    for k in ("cls_preds", "reg_preds", "dir_preds"):
        e = float((outs[True][k] - outs[False][k]).abs().max()) / float(outs[False][k].abs().max())
        print(f"\n{k}: heads on a SplitMap vs pointwise heads {e:.2e} of the scale")
        assert outs[True][k].shape == outs[False][k].shape and e <= 2e-6, k
