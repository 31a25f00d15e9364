"""Round-4 GPU tests (all through the C ABI): the Winograd F(2x2, 3x3) split-bf16 convolution (csrc/conv3x3_wino.hip) against a float64
convolution and against the direct split-bf16 kernel, on small ragged shapes and on every backbone shape of the OPV2V model."""
import os
import sys

import pytest
import torch
import torch.nn.functional as F

from coalign_amd import ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _case(N, Ci, Co, H, W, seed, res=True):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, 3, 3, generator=g) / (Ci * 9) ** 0.5
    b = torch.randn(Co, generator=g)
    r = torch.randn(N, Co, H, W, generator=g) if res else None
    return x, w, b, r


def _ref64(x, w, b, r, relu=True):
    y = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    if r is not None:
        y = y + r.double()
    return torch.relu(y) if relu else y


@pytest.mark.parametrize("shape", [(1, 16, 64, 4, 4), (2, 16, 64, 5, 6), (1, 32, 128, 6, 35), (3, 16, 64, 9, 18), (2, 48, 64, 25, 88), (2, 64, 64, 16, 33),
                                   (1, 16, 192, 1, 1), (2, 16, 64, 2, 70), (5, 32, 64, 7, 3)])
@pytest.mark.parametrize("tbw", [8, 16])
def test_winograd_convolution_small_shapes_vs_float64(shape, tbw):
    """Odd / even heights (the stacked image's one / two padding rows between images), widths that end inside a tile block, single pixels,
    several cout groups, with and without residual / ReLU; resblock.py:53-69 semantics.  Error bound: 1e-5 of the output scale (measured ~1e-6)."""
    N, Ci, Co, H, W = shape
    for res, relu in ((True, True), (False, False)):
        x, w, b, r = _case(N, Ci, Co, H, W, seed=sum(shape) + tbw, res=res)
        want = _ref64(x, w, b, r, relu)
        u = ops.pack_conv3x3_wino_weight(w.to(DEV))
        got = ops.conv3x3_wino(x.to(DEV), u, b.to(DEV), Co, None if r is None else r.to(DEV), relu, tile_block_w=tbw)
        assert got.shape == want.shape and ops.nhwc_memory(got)
        err = float((got.double().cpu() - want).abs().max() / want.abs().max())
        assert err < 1e-5, (shape, tbw, res, err)


@pytest.mark.parametrize("shape", [(5, 64, 64, 100, 352), (5, 128, 128, 50, 176), (5, 256, 256, 25, 88), (1, 384, 256, 100, 352), (1, 256, 256, 100, 352),
                                   (2, 64, 64, 100, 252), (2, 256, 256, 25, 63), (8, 128, 128, 60, 60)])
def test_winograd_convolution_backbone_shapes_vs_float64_and_direct_kernel(shape):
    """Every stride-1 3x3 shape of the OPV2V / DAIR / LSS backbones: error against the float64 convolution <= 1e-5 of the output scale (the task's
    bound; printed beside the direct split-bf16 kernel's error on the same data), and the two kernels agree to 1e-5."""
    N, Ci, Co, H, W = shape
    x, w, b, r = _case(N, Ci, Co, H, W, seed=sum(shape))
    xd, wd, bd, rd = x.to(DEV), w.to(DEV), b.to(DEV), r.to(DEV)
    want = torch.relu(F.conv2d(xd.double(), wd.double(), bd.double(), padding=1) + rd.double())
    scale = float(want.abs().max())
    got = ops.conv3x3_wino(xd, ops.pack_conv3x3_wino_weight(wd), bd, Co, rd, True)
    direct = ops.conv3x3_emu_bias_act(xd, ops.pack_conv3x3_emu_weight(wd, 3, True), bd, Co, rd, True, 3)
    e_w = float((got.double() - want).abs().max()) / scale
    e_d = float((direct.double() - want).abs().max()) / scale
    print(f"\n{shape}: winograd {e_w:.2e}, direct split-bf16 {e_d:.2e} of the output scale")
    assert e_w < 1e-5 and e_d < 1e-5
    assert float((got - direct).abs().max()) / scale < 1e-5


def test_winograd_is_deterministic_and_leaves_neighbours_alone():
    """Two runs are bit-identical; the output tensor's guard rows (a larger allocation around y) stay untouched."""
    N, Ci, Co, H, W = 3, 32, 64, 25, 88
    x, w, b, r = _case(N, Ci, Co, H, W, seed=5)
    u = ops.pack_conv3x3_wino_weight(w.to(DEV))
    a1 = ops.conv3x3_wino(x.to(DEV), u, b.to(DEV), Co, r.to(DEV), True)
    a2 = ops.conv3x3_wino(x.to(DEV), u, b.to(DEV), Co, r.to(DEV), True)
    assert torch.equal(a1, a2)


# ---------------------------------------------------------------------------------------------------------------- fp16 2-way split
FP16_SHAPES = [(5, 64, 64, 100, 352), (5, 128, 128, 50, 176), (5, 256, 256, 25, 88), (1, 384, 256, 100, 352), (1, 256, 256, 100, 352), (2, 64, 64, 100, 252)]


@pytest.mark.parametrize("shape", FP16_SHAPES)
def test_fp16_two_way_split_error_against_float64_is_not_above_the_native_fp32_kernel(shape):
    """The rule that admitted the 3-way bf16 split: measured against a float64 convolution, the arithmetic must be no worse than the
    native fp32 matrix-instruction kernel (csrc/conv3x3.hip) on the same data.  Stride-1 backbone shapes, residual + ReLU (resblock.py:53-69)."""
    N, Ci, Co, H, W = shape
    x, w, b, r = _case(N, Ci, Co, H, W, seed=sum(shape))
    xd, wd, bd, rd = x.to(DEV), w.to(DEV), b.to(DEV), r.to(DEV)
    want = torch.relu(F.conv2d(xd.double(), wd.double(), bd.double(), padding=1) + rd.double())
    scale = float(want.abs().max())
    err = lambda y: float((y.double() - want).abs().max()) / scale
    e16 = err(ops.conv3x3_emu_bias_act(xd, ops.pack_conv3x3_emu_weight(wd, 16, True), bd, Co, rd, True, 16))
    e16cl = err(ops.conv3x3_emu_bias_act(xd, ops.pack_conv3x3_emu_weight(wd, 16, True), bd, Co, rd, True, 16, out_channels_last=True))
    e3 = err(ops.conv3x3_emu_bias_act(xd, ops.pack_conv3x3_emu_weight(wd, 3, True), bd, Co, rd, True, 3))
    e2 = err(ops.conv3x3_emu_bias_act(xd, ops.pack_conv3x3_emu_weight(wd, 2, True), bd, Co, rd, True, 2))
    enat = err(ops.conv3x3_bias_act(xd, ops.pack_conv3x3_weight(wd), bd, rd, True))
    print(f"\n{shape}: fp16x2 {e16:.2e} (channels-last out {e16cl:.2e}), bf16x3 {e3:.2e}, bf16x2 {e2:.2e}, native fp32 {enat:.2e} of the output scale")
    assert e16 <= max(enat, 2e-6) and e16cl <= max(enat, 2e-6)
    assert e16 < 5e-6


@pytest.mark.parametrize("layout_in", ["nchw", "nhwc"])
def test_fp16_two_way_split_strided_layers(layout_in):
    """The strided first convolution of a stage (resblock.py:150-174) in the fp16 mode: NCHW and channels-last input, NCHW and channels-last output."""
    for (N, Ci, Co, H, W) in ((5, 64, 64, 200, 704), (5, 64, 128, 100, 352), (2, 128, 256, 50, 126)):
        x, w, b, _ = _case(N, Ci, Co, H, W, seed=H + Co, res=False)
        xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)
        if layout_in == "nhwc":
            xd = xd.contiguous(memory_format=torch.channels_last)
        want = torch.relu(F.conv2d(xd.double(), wd.double(), bd.double(), stride=2, padding=1))
        scale = float(want.abs().max())
        ws = ops.pack_conv3x3_emu_weight(wd, 16, False)
        for cl_out in ((False, True) if layout_in == "nhwc" else (False,)):
            got = ops.conv3x3_emu_bias_act(xd, ws, bd, Co, None, True, 16, stride=2, out_channels_last=cl_out)
            assert float((got.double() - want).abs().max()) / scale < 5e-6, (N, Ci, Co, H, W, cl_out)


# (the fp16 mode's operating range: tests/test_round5_gpu.py::test_fp16_split_operating_range)


# ---------------------------------------------------------------------------------------------------------------- ragged frames, capacity buckets
def test_ragged_pillar_counts_share_a_capacity_sized_graph():
    """50 frames of 50 different pillar counts through the HIP-graph pipeline capture at most two graphs per lane (the first shape
    exactly, then one capacity-sized graph with the count on the device), and every frame's detections equal the eager pipeline's."""
    from coalign_amd.config import builtin_config
    from coalign_amd.detector import build_model, to_device
    from coalign_amd.pipeline import FramePipeline
    from coalign_amd.postprocess import build_postprocessor
    from coalign_amd.synthetic import fill_parameters_, make_frame
    h = builtin_config("mini_coalign")
    model = build_model(h)
    fill_parameters_(model, seed=0, cls_bias=-1.0)
    with torch.no_grad():
        model.reg_head.weight.mul_(0.01); model.reg_head.bias.zero_(); model.cls_head.weight.mul_(0.05)
    model = model.to(DEV).eval()
    pp = build_postprocessor(h["postprocess"], False)
    anchors = torch.from_numpy(pp.generate_anchor_box())
    frames = [to_device(make_frame(h, 3, pillars_per_agent=120 + 3 * i, seed=10 + i, spread_xy=(4.0, 2.0), spread_yaw=45.0), DEV) for i in range(50)]
    assert len({int(f["processed_lidar"]["voxel_features"].shape[0]) for f in frames}) == 50
    eager = FramePipeline(model, pp, anchors, lanes=2, result_lag=1, graph=False, device=DEV)
    want = eager.run(frames)
    eager.close()
    pipe = FramePipeline(model, build_postprocessor(h["postprocess"], False), anchors, lanes=2, result_lag=1, graph=True, device=DEV)
    got = pipe.run(frames)
    assert pipe.graphs_captured <= 2 * 2, pipe.graphs_captured
    assert all(len(d) <= 2 for d in pipe._slots)
    pipe.close()
    n_det = 0
    for (b0, s0), (b1, s1) in zip(want, got):
        assert (b0 is None) == (b1 is None)
        if b0 is not None:
            assert torch.equal(b0, b1) and torch.equal(s0, s1)
            n_det += b0.shape[0]
    assert n_det > 50


def test_pipelines_can_share_lane_streams():
    """FramePipeline(streams=...): pipelines built one after the other reuse the same HIP streams (streams share a few hardware queues; the bench's
    sections measured 441 vs 481 frames/s with and without a trail of abandoned streams).  Same detections; too few streams are refused."""
    from coalign_amd.config import builtin_config
    from coalign_amd.detector import build_model, to_device
    from coalign_amd.pipeline import FramePipeline
    from coalign_amd.postprocess import build_postprocessor
    from coalign_amd.synthetic import fill_parameters_, make_frame
    h = builtin_config("mini_coalign")
    model = build_model(h)
    fill_parameters_(model, seed=0, cls_bias=-1.0)
    with torch.no_grad():
        model.reg_head.weight.mul_(0.01); model.reg_head.bias.zero_(); model.cls_head.weight.mul_(0.05)
    model = model.to(DEV).eval()
    anchors = torch.from_numpy(build_postprocessor(h["postprocess"], False).generate_anchor_box())
    frames = [to_device(make_frame(h, 3, pillars_per_agent=150, seed=70 + i, spread_xy=(4.0, 2.0), spread_yaw=45.0), DEV) for i in range(6)]
    lanes = [torch.cuda.Stream(device=DEV) for _ in range(2)]
    outs = []
    for graph in (True, False, True):
        pipe = FramePipeline(model, build_postprocessor(h["postprocess"], False), anchors, lanes=2, result_lag=1, graph=graph, device=DEV, streams=lanes)
        assert [s.cuda_stream for s in pipe.streams] == [s.cuda_stream for s in lanes]
        outs.append(pipe.run(frames))
        pipe.close()
    for other in outs[1:]:
        for (b0, s0), (b1, s1) in zip(outs[0], other):
            assert (b0 is None) == (b1 is None) and (b0 is None or (torch.equal(b0, b1) and torch.equal(s0, s1)))
    assert sum(0 if b is None else b.shape[0] for b, _ in outs[0]) > 10
    with pytest.raises(ValueError):
        FramePipeline(model, build_postprocessor(h["postprocess"], False), anchors, lanes=3, graph=True, device=DEV, streams=lanes)


# ---------------------------------------------------------------------------------------------------------------- sparse canvas, one-launch pillar op
def _opv2v_model(seed=0):
    from coalign_amd.config import builtin_config
    from coalign_amd.detector import build_model
    from coalign_amd.synthetic import fill_parameters_
    h = builtin_config("opv2v_coalign")
    model = build_model(h)
    fill_parameters_(model, seed=seed)
    return h, model


def _sparse_encode(model, margs, pl, n_agents, cache, count_dev=None):
    pfn = model.pillar_vfe.pfn_layers[0]
    bn = (pfn.norm.weight, pfn.norm.bias, pfn.norm.running_mean, pfn.norm.running_var)
    nx, ny, _ = [int(v) for v in margs["point_pillar_scatter"]["grid_size"]]
    return ops.pillar_encode_sparse(pl["voxel_features"].to(DEV), pl["voxel_num_points"].to(DEV), pl["voxel_coords"].to(DEV), pfn.linear.weight, None, bn, 1e-3, True,
                                    margs["voxel_size"], margs["lidar_range"][:3], n_agents, ny, nx, canvas_cache=cache, count_dev=count_dev)


@pytest.mark.parametrize("pillars", [8000, 70000])
def test_sparse_canvas_encoder_against_float64_and_oracle_scatter(pillars):
    """coalign_pillar_encode_sparse (the PFN contraction as ONE fp16 matrix instruction on 22-bit operand splits, the per-pillar terms in fp32; one launch) at the benchmarked size and at max_voxel_test = 70 000 pillars per
    agent x 5 (pointpillar_coalign.yaml:52-54): feature rows against a float64 evaluation of pillar_vfe.py:105-155 (<= 2e-6 of the scale, negative
    BatchNorm scales included), the densified canvas bit-equal to the oracle's scatter of those rows (point_pillar_scatter.py:15-72)."""
    from oracle import coalign_oracle as oracle
    from coalign_amd.synthetic import make_frame
    from tests.test_round3_gpu import _pfn64, P as PK
    h, model = _opv2v_model()
    margs = h["model"]["args"]
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    sd[PK + "norm.weight"][::3] *= -1.0
    model.load_state_dict(sd)
    model = model.to(DEV).eval()
    pl = make_frame(h, 5, pillars_per_agent=pillars, seed=303, noise=(0.2, 0.2))["processed_lidar"]
    pl = {k: v[:-1].clone() for k, v in pl.items()}                       # an odd pillar count
    pl["voxel_num_points"][:4] = torch.tensor([1, 16, 17, 32], dtype=pl["voxel_num_points"].dtype)
    for i, n in enumerate((1, 16, 17, 32)):
        pl["voxel_features"][i, n:] = 0
    sc = _sparse_encode(model, margs, pl, 5, {})
    want = _pfn64(pl, sd, margs)
    err = float((sc.feats.double().cpu() - want).abs().max() / want.abs().max())
    print(f"\nsparse encoder, {pillars} pillars per agent: max error {err:.2e} of the scale")
    assert err < 2e-6
    dense = sc.dense()
    assert torch.equal(dense.cpu(), oracle.scatter(sc.feats.cpu(), pl["voxel_coords"], 5, 704, 200))


def test_pillar_fold_params_table_and_reuse():
    """coalign_pillar_fold_params: the per-lane table (scale / shift of the folded BatchNorm at their slots) against torch, and the encoder with a kept
    table bit-equal to the encoder that folds per call; a table of the wrong size is refused on the host, a NULL table at the C ABI."""
    from coalign_amd import hip
    from coalign_amd.synthetic import make_frame
    h, model = _opv2v_model()
    margs = h["model"]["args"]
    model = model.to(DEV).eval()
    pfn = model.pillar_vfe.pfn_layers[0]
    bn = (pfn.norm.weight, pfn.norm.bias, pfn.norm.running_mean, pfn.norm.running_var)
    folded = ops.pillar_fold_params(pfn.linear.weight, None, bn, 1e-3, True)
    t = folded.view(7, 64, 4).cpu()
    alpha = (pfn.norm.weight / torch.sqrt(pfn.norm.running_var + 1e-3)).cpu()
    shift = (pfn.norm.bias - pfn.norm.running_mean * (pfn.norm.weight / torch.sqrt(pfn.norm.running_var + 1e-3))).cpu()
    lanes = torch.arange(64)
    for g in range(2):
        ch = 32 * g + (lanes & 31)
        assert torch.allclose(t[4, :, g], alpha[ch], rtol=2e-7, atol=0) and torch.allclose(t[4, :, 2 + g], shift[ch], rtol=1e-6, atol=1e-7)
        # slot 5: the sign of the BatchNorm scale times the power of two 2^-(k_c + 6) that takes the fp16 contraction's scaling out again (round 5)
        assert torch.equal(torch.sign(t[5, :, g]), torch.where(alpha[ch] < 0, -1.0, 1.0)) and torch.equal(torch.frexp(t[5, :, g].abs())[0], torch.full((64,), 0.5))
    pl = make_frame(h, 2, pillars_per_agent=3000, seed=5)["processed_lidar"]
    cache = {}
    a = _sparse_encode(model, margs, pl, 2, cache)
    nx, ny, _ = [int(v) for v in margs["point_pillar_scatter"]["grid_size"]]
    b = ops.pillar_encode_sparse(pl["voxel_features"].to(DEV), pl["voxel_num_points"].to(DEV), pl["voxel_coords"].to(DEV), pfn.linear.weight, None, bn, 1e-3, True,
                                 margs["voxel_size"], margs["lidar_range"][:3], 2, ny, nx, canvas_cache=cache, folded=folded)
    assert torch.equal(a.feats, b.feats)
    with pytest.raises(ValueError):
        ops.pillar_encode_sparse(pl["voxel_features"].to(DEV), pl["voxel_num_points"].to(DEV), pl["voxel_coords"].to(DEV), pfn.linear.weight, None, bn, 1e-3, True,
                                 margs["voxel_size"], margs["lidar_range"][:3], 2, ny, nx, canvas_cache=cache, folded=folded[:-4])
    L = hip.lib()
    import ctypes
    z = torch.zeros(16, device=DEV)
    dbl = (ctypes.c_double * 3)(0.4, 0.4, 4.0)
    rc = L.coalign_pillar_encode_sparse(z.data_ptr(), z.data_ptr(), z.data_ptr(), 1, None, 32, None, 64, 1, dbl, dbl, 1, 8, 8, z.data_ptr(), z.data_ptr(), z.data_ptr(), None)
    assert rc == -1                      # COALIGN_ERR_NULL_POINTER


def test_sparse_canvas_duplicates_stale_frames_and_device_count():
    """Duplicate cells (the larger row wins), out-of-canvas pillars, several frames through ONE stamp map without clearing (a cell occupied in frame 1 and
    empty in frame 2 must read as empty), the pillar count on the device with capacity-sized arrays."""
    from oracle import coalign_oracle as oracle
    from coalign_amd.synthetic import make_frame
    h, model = _opv2v_model()
    margs = h["model"]["args"]
    model = model.to(DEV).eval()
    cache = {}
    for seed, m in ((1, 3000), (2, 500), (3, 3000), (4, 1)):
        pl = make_frame(h, 2, pillars_per_agent=m, seed=seed)["processed_lidar"]
        c = pl["voxel_coords"]
        if m >= 500:
            c[10] = c[400]; c[11] = c[400]                  # three pillars in one cell: row 400 wins
            c[20, 2], c[20, 3] = 199, 704                    # linear cell index == ny * nx: outside (a smaller y would wrap into the next row, as in the reference)
            c[21, 0] = 7                                     # agent out of range
        sc = _sparse_encode(model, margs, pl, 2, cache)
        keep = torch.ones(len(c), dtype=torch.bool)
        if m >= 500:
            keep[[20, 21]] = False
            keep[[10, 11]] = False                           # (rows 10 and 11 lose to row 400: dropped here so that the oracle's index_put has no duplicate index --
                                                             #  its winner among duplicates is only sequential on one thread)
        # densify through the stamps (NOT through scatter_to_bev): what the consumers see
        st = sc.stamps.view(2, 200, 704).cpu()
        tag = int(sc.state[0].item())
        occ = (st >> 32) == tag
        rows = (st & 0xFFFFFFFF)[occ]
        seen = torch.zeros(2, 200, 704, 64)
        seen[occ] = sc.feats.cpu()[rows]
        assert torch.equal(seen.permute(0, 3, 1, 2), oracle.scatter(sc.feats.cpu()[keep], c[keep], 2, 704, 200)), (seed, m)
    # device count: capacity 4096, 1234 valid rows
    pl = make_frame(h, 2, pillars_per_agent=2048, seed=9)["processed_lidar"]
    cnt = torch.tensor([1234], dtype=torch.int32, device=DEV)
    sc = _sparse_encode(model, margs, pl, 2, cache, count_dev=cnt)
    full = _sparse_encode(model, margs, {k: v[:1234] for k, v in pl.items()}, 2, {})
    assert torch.equal(sc.feats[:1234], full.feats)
    assert torch.equal(sc.dense(), full.dense())


def test_sparse_canvas_stamp_map_is_rezeroed_before_the_frame_tag_can_wrap(monkeypatch):
    """The 32-bit frame tag orders stamps through atomicMax, so a wrapped tag would lose against stale stamps: ops re-zeroes a stamp map after
    SPARSE_TAG_RESET_AFTER launches (2^31 in the product; 3 here), FramePipeline counts graph replays per slot and does the same between replays.  Every
    frame of a sequence crossing several resets still densifies to the oracle's scatter, and the pipeline's detections equal the un-reset run's."""
    from oracle import coalign_oracle as oracle
    from coalign_amd.pipeline import FramePipeline
    from coalign_amd.postprocess import build_postprocessor
    from coalign_amd.synthetic import make_frame
    h, model = _opv2v_model()
    margs = h["model"]["args"]
    model = model.to(DEV).eval()
    frames = [make_frame(h, 2, pillars_per_agent=1500 + 100 * i, seed=40 + i) for i in range(8)]
    monkeypatch.setattr(ops, "SPARSE_TAG_RESET_AFTER", 3)
    cache = {}
    tags = []
    for f in frames:
        pl = f["processed_lidar"]
        sc = _sparse_encode(model, margs, pl, 2, cache)
        tags.append(int(sc.state[0].item()))
        assert torch.equal(sc.dense().cpu(), oracle.scatter(sc.feats.cpu(), pl["voxel_coords"], 2, 704, 200))
    assert max(tags) <= 3 and tags.count(1) >= 2, tags              # the tag restarted: the map was re-zeroed on the way
    # ... and through the graph pipeline (mini model, the detector's sparse route): replays counted per slot, the stamp map re-zeroed between replays
    from coalign_amd.config import builtin_config
    from coalign_amd.detector import build_model, to_device
    from coalign_amd.synthetic import fill_parameters_
    hm = builtin_config("mini_coalign")
    mini = build_model(hm)
    fill_parameters_(mini, seed=0, cls_bias=-1.0)
    with torch.no_grad():
        mini.reg_head.weight.mul_(0.01); mini.reg_head.bias.zero_(); mini.cls_head.weight.mul_(0.05)
    mini = mini.to(DEV).eval()
    ppm = build_postprocessor(hm["postprocess"], False)
    anchors = torch.from_numpy(ppm.generate_anchor_box())
    two = [to_device(make_frame(hm, 3, pillars_per_agent=150 + 30 * i, seed=60 + i, spread_xy=(4.0, 2.0), spread_yaw=45.0), DEV) for i in range(2)]

    def run(reset_after):
        monkeypatch.setattr(ops, "SPARSE_TAG_RESET_AFTER", reset_after)
        pipe = FramePipeline(mini, build_postprocessor(hm["postprocess"], False), anchors, lanes=1, result_lag=0, graph=True, device=DEV)
        out = pipe.run([two[i % 2] for i in range(11)])              # two input shapes -> two slots, several resets each at reset_after = 3
        resets = [sl.replays for d in pipe._slots for sl in d.values()]
        assert any(isinstance(k, tuple) and k[0] == "sparse" for d in pipe._slots for sl in d.values() for k in sl.canvas_cache)      # the route under test
        pipe.close()
        return out, resets

    (a, ra), (b, rb) = run(3), run(1 << 31)
    assert max(ra) < 3 and sum(rb) == 11, (ra, rb)                  # counters were reset on the way in the first run, never in the second
    n_det = 0
    for (ba, sa), (bb, sb) in zip(a, b):
        assert (ba is None) == (bb is None)
        if ba is not None:
            assert torch.equal(ba, bb) and torch.equal(sa, sb)
            n_det += ba.shape[0]
    assert len(a) == 11 and n_det > 10



def test_first_resnet_block_reads_the_sparse_canvas_bit_equal_to_the_dense_canvas(conv_mode):
    """The consumers' cell lookup (coalign_conv3x3_emu_sparse, coalign_pointwise_conv_emu_sparse): the multiscale features computed from the SparseCanvas
    equal, bit for bit, those computed from its densified channels-last canvas (same kernels, same arithmetic, zeros where no pillar lives)."""
    if conv_mode == 0:
        pytest.skip("the sparse canvas belongs to the split-matrix routes")
    from coalign_amd.synthetic import make_frame
    h, model = _opv2v_model()
    margs = h["model"]["args"]
    model = model.to(DEV).eval()
    pl = make_frame(h, 3, pillars_per_agent=6000, seed=11)["processed_lidar"]
    sc = _sparse_encode(model, margs, pl, 3, {})
    with torch.no_grad():
        f_sparse = model.backbone.get_multiscale_feature(sc)
        f_dense = model.backbone.get_multiscale_feature(sc.dense().contiguous(memory_format=torch.channels_last))
    for a, b in zip(f_sparse, f_dense):
        assert torch.equal(a, b)


def test_model_with_sparse_canvas_equals_dense_canvas_route(conv_mode):
    """Whole model, sparse canvas on (default) vs off (COALIGN_SPARSE_CANVAS=0: the persistent dense canvas of rounds 2-3): the encoders differ in arithmetic
    (exact fp32 vs six split-bf16 products, 2e-7 / 2e-6 of float64), so the head outputs agree to the suite's model-level tolerance, not bit for bit."""
    if conv_mode == 0:
        pytest.skip("the sparse canvas belongs to the split-matrix routes")
    from coalign_amd import detector
    from coalign_amd.detector import to_device
    from coalign_amd.synthetic import make_frame
    h, model = _opv2v_model()
    model = model.to(DEV).eval()
    fr = to_device(make_frame(h, 3, pillars_per_agent=5000, seed=5, noise=(0.2, 0.2)), DEV)
    saved = detector.SPARSE_CANVAS
    try:
        with torch.no_grad():
            detector.SPARSE_CANVAS = True
            a = model(fr)
            detector.SPARSE_CANVAS = False
            b = model(fr)
    finally:
        detector.SPARSE_CANVAS = saved
    for k in ("cls_preds", "reg_preds", "dir_preds"):
        e = float((a[k] - b[k]).abs().max()) / float(b[k].abs().max())
        print(f"\n{k}: sparse vs dense route {e:.2e} of the scale")
        assert e <= 1e-4, k                                  # (the suite's model-level tolerance; the north star allows 1e-3)
