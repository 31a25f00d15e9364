"""Round-3 GPU tests (all through the C ABI): the matrix-core pillar encoder, the points -> detections feeder path, latency mode,
the parallel NMS walk, the stream-K hand-over stress test, cfg 4's correction path at full size."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import coalign_oracle as oracle
from coalign_amd import ops
from coalign_amd.config import builtin_config
from coalign_amd.detector import build_model, to_device
from coalign_amd.synthetic import fill_parameters_, make_frame

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = "pillar_vfe.pfn_layers.0."


def _pfn64(pl, sd, margs):
    """PillarVFE (pillar_vfe.py:105-155, PFNLayer :31-53) evaluated in float64 from the float32 inputs: the yardstick for both encoders."""
    vf = pl["voxel_features"].double()
    npts = pl["voxel_num_points"].double()
    cd = pl["voxel_coords"]
    vs, r = margs["voxel_size"], margs["lidar_range"]
    mean = vf[:, :, :3].sum(1, keepdim=True) / npts.view(-1, 1, 1)
    f32 = lambda v: torch.tensor(v, dtype=torch.float32)
    ctr = torch.stack([cd[:, 3].float() * f32(vs[0]) + f32(vs[0] / 2 + r[0]), cd[:, 2].float() * f32(vs[1]) + f32(vs[1] / 2 + r[1]),
                       cd[:, 1].float() * f32(vs[2]) + f32(vs[2] / 2 + r[2])], 1).double()           # the float32 centre the reference forms
    f = torch.cat([vf, vf[:, :, :3] - mean, vf[:, :, :3] - ctr[:, None, :]], -1)
    mask = (torch.arange(vf.shape[1])[None, :] < pl["voxel_num_points"][:, None]).double()[..., None]
    f = f * mask
    x = f @ sd[P + "linear.weight"].double().t()
    alpha = sd[P + "norm.weight"].double() / torch.sqrt(sd[P + "norm.running_var"].double() + 1e-3)
    x = (x - sd[P + "norm.running_mean"].double()) * alpha + sd[P + "norm.bias"].double()
    return torch.relu(x).max(1)[0]


def _run_pillar(pl, sd, margs, n_agents, cl, use_abs=True):
    nx, ny, _ = [int(v) for v in margs["point_pillar_scatter"]["grid_size"]]
    bn = tuple(sd[P + k].to(DEV) for k in ("norm.weight", "norm.bias", "norm.running_mean", "norm.running_var"))
    return ops.pillar_vfe_scatter(pl["voxel_features"].to(DEV), pl["voxel_num_points"].to(DEV), pl["voxel_coords"].to(DEV), sd[P + "linear.weight"].to(DEV),
                                  None, bn, 1e-3, use_abs, False, margs["voxel_size"], margs["lidar_range"][:3], n_agents, ny, nx, channels_last=cl)


def test_matrix_core_encoder_against_float64():
    """The linearised split-bf16 encoder (pillar_scatter.hip, pair_compute_mx) at the benchmarked size against a float64 evaluation of
    pillar_vfe.py:105-155: error <= 2e-6 of the output scale (the reference's own float32 evaluation -- the oracle -- sits at the same level),
    negative BatchNorm scales (the row max turns into a min), 1- / 16- / 17- / 32-point pillars, an odd pillar count."""
    h = builtin_config("opv2v_coalign")
    margs = h["model"]["args"]
    model = build_model(h)
    fill_parameters_(model, seed=0)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    sd[P + "norm.weight"][::3] *= -1.0                                  # a third of the channels with a negative scale
    pl = make_frame(h, 5, pillars_per_agent=8000, seed=303, noise=(0.2, 0.2))["processed_lidar"]
    pl = {k: v[:-1].clone() for k, v in pl.items()}
    for row, n in ((0, 1), (1, 16), (2, 17), (3, 32), (4, 2)):
        pl["voxel_num_points"][row] = n
        pl["voxel_features"][row, n:] = 0
        if n > 1:
            pl["voxel_features"][row, :n] = pl["voxel_features"][row, :1] + 0.01 * torch.randn(n, 4, generator=torch.Generator().manual_seed(row))
    want = _pfn64(pl, sd, margs)
    scale = float(want.abs().max())
    ref32 = oracle.pillar_vfe(pl["voxel_features"], pl["voxel_num_points"], pl["voxel_coords"], sd, margs["voxel_size"], margs["lidar_range"])
    err_ref = float((ref32.double() - want).abs().max()) / scale
    for cl in (True, False):
        feats, canvas = _run_pillar(pl, sd, margs, 5, cl)
        err = float((feats.cpu().double() - want).abs().max()) / scale
        assert err <= 2e-6, (cl, err, err_ref)
        assert err <= 4 * err_ref + 2e-7, (cl, err, err_ref)            # no worse than a few float32 roundings of the reference's own evaluation
        assert torch.equal(canvas.contiguous().cpu(), oracle.scatter(feats.cpu(), pl["voxel_coords"], 5, 704, 200))


def test_matrix_core_encoder_many_pillars_several_rounds():
    """90 000 pillars: a wavefront of pillar_rows_mx_kernel owns 11 pairs = three LDS-DMA rounds (the benchmarked 40 000 pillars are one round);
    features against the oracle, the canvas an exact copy, persistent canvas == fresh canvas when the big frame follows a small one."""
    h = builtin_config("opv2v_coalign")
    margs = h["model"]["args"]
    model = build_model(h)
    fill_parameters_(model, seed=1)
    sd = model.state_dict()
    big = make_frame(h, 3, pillars_per_agent=30000, seed=77)["processed_lidar"]
    small = make_frame(h, 3, pillars_per_agent=500, seed=78)["processed_lidar"]
    ref = oracle.pillar_vfe(big["voxel_features"], big["voxel_num_points"], big["voxel_coords"], sd, margs["voxel_size"], margs["lidar_range"])
    feats, canvas = _run_pillar(big, sd, margs, 3, True)
    scale = float(ref.abs().max())
    assert float((feats.cpu() - ref).abs().max()) <= 2e-6 * scale
    assert torch.equal(canvas.contiguous().cpu(), oracle.scatter(feats.cpu(), big["voxel_coords"], 3, 704, 200))
    bn = tuple(sd[P + k].to(DEV) for k in ("norm.weight", "norm.bias", "norm.running_mean", "norm.running_var"))
    cache = {}
    for pl in (small, big, small):
        args = (pl["voxel_features"].to(DEV), pl["voxel_num_points"].to(DEV), pl["voxel_coords"].to(DEV), sd[P + "linear.weight"].to(DEV), None, bn, 1e-3, True, False,
                margs["voxel_size"], margs["lidar_range"][:3], 3, 200, 704)
        f1, c1 = ops.pillar_vfe_scatter(*args, channels_last=True, canvas_cache=cache)
        f0, c0 = ops.pillar_vfe_scatter(*args, channels_last=True)
        assert torch.equal(f1, f0) and torch.equal(c1, c0)


def test_matrix_core_encoder_without_absolute_xyz():
    """use_absolute_xyz = False (7 input features; not in the five configs): the same kernel with the centre term dropped."""
    h = builtin_config("mini_coalign")
    margs = h["model"]["args"]
    pl = make_frame(h, 2, pillars_per_agent=300, seed=9, num_points_mode="uniform")["processed_lidar"]
    g = torch.Generator().manual_seed(2)
    sd = {P + "linear.weight": torch.randn(64, 7, generator=g) * 0.3, P + "norm.weight": torch.randn(64, generator=g), P + "norm.bias": torch.randn(64, generator=g) * 0.1,
          P + "norm.running_mean": torch.randn(64, generator=g) * 0.1, P + "norm.running_var": torch.rand(64, generator=g) + 0.5}
    vf = pl["voxel_features"].double()
    mean = vf[:, :, :3].sum(1, keepdim=True) / pl["voxel_num_points"].double().view(-1, 1, 1)
    vs, r = margs["voxel_size"], margs["lidar_range"]
    cd = pl["voxel_coords"]
    f32 = lambda v: torch.tensor(v, dtype=torch.float32)
    ctr = torch.stack([cd[:, 3].float() * f32(vs[0]) + f32(vs[0] / 2 + r[0]), cd[:, 2].float() * f32(vs[1]) + f32(vs[1] / 2 + r[1]),
                       cd[:, 1].float() * f32(vs[2]) + f32(vs[2] / 2 + r[2])], 1).double()
    f = torch.cat([vf[:, :, 3:], vf[:, :, :3] - mean, vf[:, :, :3] - ctr[:, None, :]], -1)
    f = f * (torch.arange(32)[None, :] < pl["voxel_num_points"][:, None]).double()[..., None]
    x = f @ sd[P + "linear.weight"].double().t()
    alpha = sd[P + "norm.weight"].double() / torch.sqrt(sd[P + "norm.running_var"].double() + 1e-3)
    want = torch.relu((x - sd[P + "norm.running_mean"].double()) * alpha + sd[P + "norm.bias"].double()).max(1)[0]
    for cl in (True, False):
        feats, _ = _run_pillar(pl, sd, margs, 2, cl, use_abs=False)
        assert float((feats.cpu().double() - want).abs().max()) <= 2e-6 * float(want.abs().max())


_PILLAR_TESTS = ["tests/test_hip_parity.py::test_pillar_golden_mini", "tests/test_hip_parity.py::test_pillar_fullsize_vs_oracle_and_reference",
                 "tests/test_hip_parity.py::test_pillar_edge_cases", "tests/test_hip_parity.py::test_pillar_dense_duplicates_and_unfused_route",
                 "tests/test_round2_gpu.py::test_pillar_channels_last_canvas_equals_nchw", "tests/test_round2_gpu.py::test_pillar_persistent_canvas_equals_fresh_canvas",
                 "tests/test_round3_gpu.py::test_matrix_core_encoder_against_float64"]


def test_valu_pillar_encoder_in_a_subprocess():
    """COALIGN_PILLAR_MFMA=0 (read at library load) puts the fp32 VALU encoder back on every route: the pillar tests again, in their own process."""
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"] + _PILLAR_TESTS,
                       env=dict(os.environ, PYTHONPATH=ROOT, COALIGN_PILLAR_MFMA="0"), capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-500:])


# ------------------------------------------------------------------------------------------------ the feeder in the loop
@pytest.fixture(scope="module")
def points_world():
    """Four 3-cav frames of raw sweeps (host memory), a calibrated detector, and the same frames voxelised up front (the from-pillars
    form of the very same data: coalign_voxelize, pillar count read back on the host)."""
    from coalign_amd.postprocess import build_postprocessor
    from coalign_amd.preprocess import build_preprocessor
    from coalign_amd.synthetic import calibrate_heads_, make_points_frame
    h = builtin_config("opv2v_coalign")
    pre = build_preprocessor(h["preprocess"], False, DEV)
    raw = [make_points_frame(h, 3, seed=400 + i, noise=(0.2, 0.2)) for i in range(4)]
    pillars = []
    for f in raw:
        out = pre.preprocess_clouds(f["clouds"], ego_filter=True)
        pillars.append({"processed_lidar": {k: out[k] for k in ("voxel_features", "voxel_coords", "voxel_num_points")}, "record_len": [3],
                        "pairwise_t_matrix": f["pairwise_t_matrix"].to(DEV)})
    model = build_model(h)
    fill_parameters_(model, seed=0)
    model = model.to(DEV).eval()
    pp = build_postprocessor(h["postprocess"], False)
    calibrate_heads_(model, pillars[0], pp.params["target_args"]["score_threshold"], 400)
    anchors = torch.from_numpy(pp.generate_anchor_box())
    meta = {"ego": {"transformation_matrix": torch.eye(4, device=DEV), "anchor_box": anchors}}
    with torch.no_grad():
        sync = [pp.post_process(meta, {"ego": model(f)}) for f in pillars]
    torch.cuda.synchronize()
    return {"hypes": h, "pre": pre, "raw": raw, "pillars": pillars, "model": model, "pp": pp, "anchors": anchors, "sync": sync}


def _same(a, b):
    return (a is None and b is None) or (a is not None and b is not None and torch.equal(a, b))


@pytest.mark.parametrize("graph", [False, True], ids=["eager", "hip_graph"])
def test_submit_points_equals_the_from_pillars_path(points_world, graph):
    """FramePipeline.submit_points (pinned host clouds -> async copy -> coalign_voxelize -> coalign_pillar_encode_stream with the pillar
    count on the device -> ... -> NMS) gives, frame for frame and over three rounds of the pool, exactly the detections of the synchronous
    model + post_process on the same clouds voxelised up front: same pillars in the same order, unique-cell route == cell-map route."""
    from coalign_amd.pipeline import FramePipeline
    w = points_world
    assert any(s[0] is not None and s[0].shape[0] > 20 for s in w["sync"]), "the test frames must produce detections"
    pipe = FramePipeline(w["model"], w["pp"], w["anchors"], lanes=3, result_lag=1, graph=graph, preprocessor=w["pre"], points_per_cloud=98304)
    try:
        results = []
        order = [0, 1, 2, 3, 3, 0, 2, 1, 1, 3, 0, 2]
        for i in order:
            results += pipe.submit_points(w["raw"][i])
        results += pipe.drain()
        assert [r[0] for r in results] == list(range(len(order)))
        for (idx, boxes, scores), i in zip(results, order):
            sb, ss = w["sync"][i]
            assert _same(boxes, sb) and _same(scores, ss), f"frame {idx} (pool {i}): from-points result differs from the from-pillars one"
        assert len(pipe.latencies_ms) == len(order) and all(l > 0 for l in pipe.latencies_ms)
    finally:
        pipe.close()


def test_pillar_encode_stream_equals_the_host_count_form(points_world):
    """coalign_pillar_encode_stream (capacity-sized arrays, count on the device) against coalign_pillar_encode_persistent on the sliced
    arrays: same canvas bit for bit over a sequence of frames with different counts, with and without the cell map; rows past the count
    are never touched (feature rows stay at their fill value)."""
    w = points_world
    margs = w["hypes"]["model"]["args"]
    pfn = w["model"].pillar_vfe.pfn_layers[0]
    bn = (pfn.norm.weight, pfn.norm.bias, pfn.norm.running_mean, pfn.norm.running_var)
    cache_a, cache_b, cache_c = {}, {}, {}
    for i in (0, 1, 2, 0, 3):
        clouds = w["raw"][i]["clouds"]
        pts = torch.cat([torch.from_numpy(c) for c in clouds]).to(DEV)
        off = np.concatenate([[0], np.cumsum([len(c) for c in clouds])]).tolist()
        voxels, coords, num, counts = ops.voxelize(pts, off, w["pre"].voxel_size, w["pre"].lidar_range, 32, w["pre"].max_voxels, ego_filter=True)
        m = int(counts[-1])
        args = (pfn.linear.weight, None, bn, 1e-3, True, False, margs["voxel_size"], margs["lidar_range"][:3], 3, 200, 704)
        f_ref, c_ref = ops.pillar_vfe_scatter(voxels[:m], num[:m], coords[:m], *args, channels_last=True, canvas_cache=cache_a)
        for cache, unique in ((cache_b, True), (cache_c, False)):
            f, c = ops.pillar_encode_stream(voxels, num, coords, counts[3:], *args, canvas_cache=cache, unique_cells=unique, want_features=True)
            assert torch.equal(c, c_ref), (i, unique)
            assert torch.equal(f[:m], f_ref), (i, unique)


# ------------------------------------------------------------------------------------------------ fast NMS (rank16 / mask2 / reduce2)
def _nms_inputs(K, seed, dense=False):
    rs = np.random.RandomState(seed)
    b7 = np.zeros((K, 7), np.float32)
    span = (30.0, 12.0) if dense else (100.0, 40.0)
    b7[:, 0] = rs.uniform(-span[0], span[0], K); b7[:, 1] = rs.uniform(-span[1], span[1], K); b7[:, 2] = rs.uniform(-1.5, -0.5, K); b7[:, 3] = 1.5
    b7[:, 4] = rs.uniform(1.4, 2.2, K); b7[:, 5] = rs.uniform(3, 5.5, K); b7[:, 6] = rs.uniform(-3.2, 3.2, K)
    corners = oracle.boxes_to_corners_3d(torch.from_numpy(b7), "hwl")
    scores = torch.from_numpy(rs.uniform(0.2, 1, K).astype(np.float32))
    valid = torch.from_numpy((rs.uniform(0, 1, K) > 0.1).astype(np.uint8))
    return corners, scores, valid


@pytest.mark.parametrize("K,dense", [(1, False), (63, True), (64, True), (65, True), (600, True), (600, False), (1000, True), (1024, True), (2500, False)])
def test_fused_nms_gather_equals_the_separate_calls_and_the_oracle(K, dense):
    """coalign_nms_rotated_gather (rank16 -> mask2 -> reduce2 with the in-range gather) against (a) the oracle's nms_rotated on the valid
    candidates + the range rule, (b) coalign_nms_rotated + coalign_gather_in_range: keep lists, counts and gathered rows identical.
    Dense sets (most pairs overlap: long compacted lists, deep suppression chains), sets of more than `top` candidates, block edges."""
    corners, scores, valid = _nms_inputs(K, 100 + K, dense)
    top = 1000
    rng = [-60.0, -38.0, -3.0, 60.0, 38.0, 1.0]
    c, s, v = corners.to(DEV), scores.to(DEV), valid.to(DEV)
    L = ops.hip.lib()
    ws = torch.empty(L.coalign_nms_rotated_workspace_bytes(K, top), dtype=torch.uint8, device=DEV)
    keep = torch.full((top,), -1, dtype=torch.int32, device=DEV)
    cnt = torch.zeros(1, dtype=torch.int32, device=DEV)
    oc, osc, on = torch.zeros(top, 8, 3, device=DEV), torch.zeros(top, device=DEV), torch.zeros(1, dtype=torch.int32, device=DEV)
    ops.nms_rotated_gather(c, s, 0.15, top, v, None, keep, cnt, rng, oc, osc, on, ws)
    n = int(cnt)
    got = keep[:n].cpu().numpy()
    idx = np.nonzero(valid.numpy())[0]
    want = idx[oracle.nms_rotated(corners.numpy()[idx], scores.numpy()[idx], 0.15)]
    assert np.array_equal(got, want.astype(np.int32)), (K, dense, n, len(want))
    # (b) the two-call form
    k2, c2 = ops.nms_rotated_device(c, s, 0.15, top, valid=v)
    assert int(c2) == n and torch.equal(k2[:n], keep[:n])
    oc2, os2, on2 = torch.zeros_like(oc), torch.zeros_like(osc), torch.zeros_like(on)
    ops.gather_in_range(c, s, k2, c2, rng, oc2, os2, on2)
    m = int(on)
    assert m == int(on2) and torch.equal(oc[:m], oc2[:m]) and torch.equal(osc[:m], os2[:m])
    inside = [i for i in want if bool(((corners[i].double() >= torch.tensor(rng[:3]).double()) & (corners[i].double() <= torch.tensor(rng[3:]).double())).all())]
    assert m == len(inside) and torch.equal(oc[:m].cpu(), corners[inside]) and torch.equal(osc[:m].cpu(), scores[inside])


def test_legacy_nms_kernels_in_a_subprocess():
    """COALIGN_NMS_LEGACY=1 selects round 2's rank / mask / reduce kernels (kept for tops above 1024 and as the A/B reference): the NMS tests
    again, in their own process."""
    tests = ["tests/test_hip_parity.py::test_nms_golden", "tests/test_hip_parity.py::test_nms_edge_cases_and_properties",
             "tests/test_round3_gpu.py::test_fused_nms_gather_equals_the_separate_calls_and_the_oracle"]
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"] + tests,
                       env=dict(os.environ, PYTHONPATH=ROOT, COALIGN_NMS_LEGACY="1"), capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-500:])


# ------------------------------------------------------------------------------------------------ stream-K hand-over
@pytest.mark.parametrize("kind", ["bf16x3", "fp32"])
def test_stream_k_handover_stress(kind):
    """The shrink-header shape (1 x 384 -> 256 at 100 x 352: long tiles, split between workgroups and handed over through write-through
    stores + a flag, conv3x3_emu.hip / conv3x3.hip) launched 2000 times while a second stream keeps the GPU busy with other convolutions and
    copies: every output bit-equal to the first.  (The hand-over is relaxed-atomic + ISA ordering, argued next to the code; this guards it.)"""
    g = torch.Generator().manual_seed(7)
    x = torch.randn(1, 384, 100, 352, generator=g).to(DEV)
    w = (torch.randn(256, 384, 3, 3, generator=g) / (384 * 9) ** 0.5).to(DEV)
    b = torch.randn(256, generator=g).to(DEV)
    x2 = torch.randn(5, 64, 100, 352, generator=g).to(DEV)
    w2 = (torch.randn(64, 64, 3, 3, generator=g) / 24.0).to(DEV)
    b2 = torch.randn(64, generator=g).to(DEV)
    if kind == "bf16x3":
        wp, wp2 = ops.pack_conv3x3_emu_weight(w, 3, True), ops.pack_conv3x3_emu_weight(w2, 3, True)
        run = lambda: ops.conv3x3_emu_bias_act(x, wp, b, 256, None, True, 3)
        other = lambda: ops.conv3x3_emu_bias_act(x2, wp2, b2, 64, None, True, 3)
    else:
        wp, wp2 = ops.pack_conv3x3_weight(w), ops.pack_conv3x3_weight(w2)
        run = lambda: ops.conv3x3_bias_act(x, wp, b, None, True)
        other = lambda: ops.conv3x3_bias_act(x2, wp2, b2, None, True)
    ref = run()
    want = torch.relu(torch.nn.functional.conv2d(x.double(), w.double(), b.double(), padding=1))
    assert float((ref.double() - want).abs().max() / want.abs().max()) < 5e-6
    side = torch.cuda.Stream()
    scratch = torch.empty(64 << 20, dtype=torch.uint8, device=DEV)
    bad = torch.zeros((), dtype=torch.int64, device=DEV)
    n_iter = 2000 if kind == "bf16x3" else 1000
    for it in range(n_iter):
        if it % 4 == 0:                       # keep the second stream a few launches deep: varied co-runners, some of them bandwidth hogs
            with torch.cuda.stream(side):
                other()
                if it % 8 == 0:
                    scratch.zero_()
        y = run()
        bad += (y != ref).any()               # EVERY iteration is compared, on the device (no host synchronisation: launches stay back to back)
        if it % 97 == 96:
            torch.cuda.synchronize()          # ... with launches that start on an idle GPU mixed in
    torch.cuda.synchronize()
    assert int(bad) == 0, int(bad)


# ------------------------------------------------------------------------------------------------ small reference branches
def test_iou_preds_rescoring_golden(golden):
    """voxel_postprocessor.py:335-339: an `iou_preds` head rescales the candidate scores ((clamp(sigmoid, 0, 1) + 1) / 2) ** 4 before NMS.
    Golden: the reference's post_process on the mini model's heads + a random IoU map (tests/golden/dataset_branches.npz)."""
    from coalign_amd.postprocess import build_postprocessor
    g, gb = golden("postprocess.npz"), golden("dataset_branches.npz")
    pp = build_postprocessor(builtin_config("mini_coalign")["postprocess"], False)
    T = torch.from_numpy
    out = {"ego": {"cls_preds": T(g["i_cls"]).to(DEV), "reg_preds": T(g["i_reg"]).to(DEV), "dir_preds": T(g["i_dir"]).to(DEV), "iou_preds": T(gb["iou_preds"]).to(DEV)}}
    boxes, scores = pp.post_process({"ego": {"transformation_matrix": torch.eye(4), "anchor_box": T(g["anchors"])}}, out)
    assert boxes.shape == gb["iou_boxes"].shape
    np.testing.assert_allclose(scores.cpu().numpy(), gb["iou_scores"], rtol=3e-7, atol=0)
    np.testing.assert_allclose(boxes.cpu().numpy(), gb["iou_boxes"], rtol=2e-6, atol=1e-5)
    assert not np.allclose(gb["iou_scores"][: len(g["i_scores"])], g["i_scores"][: len(gb["iou_scores"])])      # the rescoring changed something


# ------------------------------------------------------------------------------------------------ cfg 4: the correction path at full size
def _dair_stage1_hypes():
    """The stage-1 (PointPillarUncertainty) config at DAIR-V2X-C geometry: 504 x 200 canvas, anchors l = 4.5, w = 2 (dairv2x yaml :57-75)."""
    import copy
    from coalign_amd.config import load_point_pillar_params
    h, hd = copy.deepcopy(builtin_config("opv2v_pointpillar_uncertainty")), builtin_config("dairv2x_coalign")
    rng, vox = list(hd["preprocess"]["cav_lidar_range"]), list(hd["preprocess"]["args"]["voxel_size"])
    h["preprocess"]["cav_lidar_range"], h["preprocess"]["args"]["voxel_size"] = rng, vox
    h["model"]["args"]["lidar_range"], h["model"]["args"]["voxel_size"] = rng, vox
    h["postprocess"]["anchor_args"].update({"cav_lidar_range": rng, "l": 4.5, "w": 2, "h": 1.56})
    h["postprocess"]["gt_range"] = rng
    return load_point_pillar_params(h)


def _plant_stage1_heads(objects_agent, anchors, rs):
    """Head maps (cls / reg / unc, [1, A * k, H, W]) whose decode gives exactly `objects_agent` ([K, 7] = x, y, z, h, w, l, yaw in the agent's
    frame): logit +4 at the nearest anchor, -9 elsewhere; regression deltas = the inverse of delta_to_boxes3d (voxel_postprocessor.py:405-450)."""
    H, W, A, _ = anchors.shape
    cls = np.full((1, A, H, W), -9.0, np.float32)
    reg = np.zeros((1, A * 7, H, W), np.float32)
    unc = rs.normal(-2.0, 0.3, (1, A * 3, H, W)).astype(np.float32)
    xs, ys = anchors[0, :, 0, 0], anchors[:, 0, 0, 1]
    for b in objects_agent:
        j, i = int(np.abs(xs - b[0]).argmin()), int(np.abs(ys - b[1]).argmin())
        a = int(np.abs(np.cos(b[6] - anchors[i, j, :, 6])).argmax())          # the anchor yaw (0 / 90 deg) closest to the heading, modulo pi
        an = anchors[i, j, a]
        r = b[6] - an[6]                  # not wrapped: without a direction head the decode returns r + r_anchor, i.e. the heading itself
        d = np.sqrt(an[4] ** 2 + an[5] ** 2)
        delta = [(b[0] - an[0]) / d, (b[1] - an[1]) / d, (b[2] - an[2]) / an[3], np.log(b[3] / an[3]), np.log(b[4] / an[4]), np.log(b[5] / an[5]), r]
        cls[0, a, i, j] = 4.0
        reg[0, a * 7: a * 7 + 7, i, j] = delta
    return cls, reg, unc


def _relative_error(poses, clean):
    """(translation error [m], yaw error [deg]) of T_ego<-infra built from `poses` against the clean one."""
    from coalign_amd.pose import get_pairwise_transformation
    T, Tc = get_pairwise_transformation(poses, 2)[1, 0], get_pairwise_transformation(clean, 2)[1, 0]
    dyaw = np.degrees(np.arctan2(T[1, 0], T[0, 0]) - np.arctan2(Tc[1, 0], Tc[0, 0]))
    return float(np.hypot(*(T[:2, 3] - Tc[:2, 3]))), float(abs((dyaw + 180) % 360 - 180))


def test_cfg4_correction_path_full_geometry_noise_sweep():
    """BASELINE configs[3] end to end at the DAIR-V2X-C geometry (504 x 200, vehicle + road-side unit facing back), pose noise sigma in
    {0, 0.2, 0.4, 0.6} (m / deg) like opencood/tools/inference_w_noise.py:39-214:
      stage 1 on the device (PointPillarUncertainty forward against the oracle; post_process_stage1 on head maps that decode to the two
      agents' views of ONE scene -- a random-init network cannot produce matching detections in two frames, so the detections are planted at
      the head, everything after the head is the product path) -> box alignment (host graph + coalign_pose_graph_optimize) -> corrected poses
      (hook of intermediate_fusion_dataset.py:301-328) -> pairwise matrices -> CoAlign fusion model -> post-process -> TP / FP at IoU 0.3 / 0.5 / 0.7.
    Asserted per sigma: device stage-1 detections == oracle's, device-aligned poses == oracle-aligned (1e-5 m / 1e-4 deg on the same boxes), alignment
    LOWERS the relative pose error (sigma > 0), TP / FP sequences and AP identical to the oracle's run of the same pipeline."""
    import math
    from coalign_amd import box_align, evaluation as ev
    from coalign_amd.pose import generate_noise, get_pairwise_transformation
    from coalign_amd.postprocess import build_postprocessor
    from coalign_amd.synthetic import calibrate_heads_
    from tests.inference_synthetic import plant_ground_truth

    # ---- stage 1: the network at full geometry against the oracle (2 agents)
    h1 = _dair_stage1_hypes()
    assert [int(v) for v in h1["model"]["args"]["point_pillar_scatter"]["grid_size"]][:2] == [504, 200]
    m1 = build_model(h1)
    fill_parameters_(m1, seed=2, cls_bias=-1.0)
    sd1 = {k: v.clone() for k, v in m1.state_dict().items()}
    hd = builtin_config("dairv2x_coalign")
    frame = make_frame(hd, 2, pillars_per_agent=7000, seed=5, infra_agent=True)
    with torch.no_grad():
        out1 = m1.to(DEV).eval()({"processed_lidar": to_device(frame["processed_lidar"], DEV)})
        ref1 = oracle.pointpillar_forward(sd1, h1["model"]["args"], frame)
    for k in ("cls_preds", "reg_preds", "unc_preds"):
        e = float((out1[k].cpu() - ref1[k]).abs().max()) / float(ref1[k].abs().max())
        assert e < 1e-4, (k, e)

    # ---- one scene, two views: objects in the ego (= world) frame, each agent sees the ones inside its range (+ 5 cm of detection noise)
    rs = np.random.RandomState(42)
    clean = [np.zeros(6), np.array([30.0, 5.0, 0.0, 0.0, 170.0, 0.0])]
    gx, gy = np.meshgrid(np.arange(-24, 72, 12.0), np.arange(-30, 31, 10.0))
    world = np.stack([gx.ravel() + rs.uniform(-2, 2, gx.size), gy.ravel() + rs.uniform(-2, 2, gx.size)], 1)
    yaw_w = rs.uniform(-2.5, 2.5, len(world))                 # headings away from +-pi: the two agents' world yaws must not straddle the wrap
    pp1 = build_postprocessor(h1["postprocess"], False)
    anchors1 = pp1.generate_anchor_box()
    rngd = hd["preprocess"]["cav_lidar_range"]
    heads = {"cls_preds": [], "reg_preds": [], "unc_preds": []}
    for pose in clean:
        th = math.radians(pose[4])
        R = np.array([[math.cos(th), math.sin(th)], [-math.sin(th), math.cos(th)]])           # world -> agent
        xy = (world - pose[:2]) @ R.T + rs.normal(0, 0.05, world.shape)
        inside = (xy[:, 0] > rngd[0] + 6) & (xy[:, 0] < rngd[3] - 6) & (xy[:, 1] > rngd[1] + 6) & (xy[:, 1] < rngd[4] - 6)
        obj = np.zeros((int(inside.sum()), 7))
        obj[:, :2], obj[:, 2], obj[:, 3:6], obj[:, 6] = xy[inside], -1.0, [1.56, 2.0, 4.5], yaw_w[inside] - th
        c, r, u = _plant_stage1_heads(obj, anchors1, rs)
        heads["cls_preds"].append(c); heads["reg_preds"].append(r); heads["unc_preds"].append(u)
    heads = {k: torch.from_numpy(np.concatenate(v)) for k, v in heads.items()}
    a1 = torch.from_numpy(anchors1)
    cd, bd, ud = pp1.post_process_stage1({k: v.to(DEV) for k, v in heads.items()}, a1)
    co, bo, uo = oracle.post_process_stage1(heads, a1, h1["postprocess"])
    assert [len(c) for c in cd] == [len(c) for c in co] and min(len(c) for c in cd) >= 20
    for i in range(2):
        np.testing.assert_allclose(cd[i].cpu().numpy(), co[i].numpy(), rtol=1e-5, atol=1e-4)
        assert np.array_equal(ud[i].cpu().numpy(), uo[i].numpy())
    corners_d = [c.cpu().numpy().astype(np.float64) for c in cd]
    corners_o = [c.numpy().astype(np.float64) for c in co]
    unc_d, unc_o = [u.cpu().numpy().astype(np.float64) for u in ud], [u.numpy().astype(np.float64) for u in uo]

    # ---- the fusion model (full geometry), heads calibrated on the clean-pose frame, logits kept clear of the score threshold
    model = build_model(hd)
    fill_parameters_(model, seed=1)
    model = model.to(DEV).eval()
    pp = build_postprocessor(hd["postprocess"], False)
    fd = to_device(frame, DEV)
    calibrate_heads_(model, fd, pp.params["target_args"]["score_threshold"], 400)
    anchors = torch.from_numpy(pp.generate_anchor_box())
    meta = {"ego": {"transformation_matrix": torch.eye(4), "anchor_box": anchors}}
    flags = dict(use_uncertainty=True, landmark_SE2=True, adaptive_landmark=False, normalize_uncertainty=False, abandon_hard_cases=True, drop_hard_boxes=True)
    sigmas = (0.0, 0.2, 0.4, 0.6)
    runs = []
    for s in sigmas:
        g = np.random.RandomState(1000 + int(10 * s))
        noisy = np.array([p + generate_noise(s, s, rng=g) for p in clean])
        ref_d = box_align.box_alignment_relative_sample_np(corners_d, noisy.copy(), uncertainty_list=unc_d, **flags)
        # the two solvers on the SAME graph input (the device's stage-1 boxes) ...
        ref_same = oracle.box_alignment_relative_sample_np(corners_d, noisy.copy(), unc_d, **flags)
        np.testing.assert_allclose(ref_d[:, :2], ref_same[:, :2], rtol=0, atol=1e-5, err_msg=f"sigma {s}")
        assert np.abs((ref_d[:, 2] - ref_same[:, 2] + 180) % 360 - 180).max() < 1e-4, s
        # ... and the oracle's own chain (its stage-1 boxes differ from the device's by float32 decode rounding, <= 1e-4 m: so do the optima)
        ref_o = oracle.box_alignment_relative_sample_np(corners_o, noisy.copy(), unc_o, **flags)
        np.testing.assert_allclose(ref_d[:, :2], ref_o[:, :2], rtol=0, atol=2e-3, err_msg=f"sigma {s}")
        assert np.abs((ref_d[:, 2] - ref_o[:, 2] + 180) % 360 - 180).max() < 2e-3, s
        fixed_d, fixed_o = noisy.copy(), noisy.copy()
        fixed_d[:, [0, 1, 4]], fixed_o[:, [0, 1, 4]] = ref_d, ref_o
        e_noisy, e_fixed = _relative_error(noisy, clean), _relative_error(fixed_d, clean)
        print(f"cfg4 sigma {s}: relative pose error noisy {e_noisy[0]:.3f} m / {e_noisy[1]:.3f} deg -> aligned {e_fixed[0]:.3f} m / {e_fixed[1]:.3f} deg")
        if s > 0:
            assert e_fixed[0] < 0.5 * e_noisy[0] + 0.02 and e_fixed[1] < 0.5 * e_noisy[1] + 0.02, (s, e_noisy, e_fixed)
        assert e_fixed[0] < 0.08 and e_fixed[1] < 0.08, (s, e_fixed)                    # left: the 5 cm detection noise of the planted boxes
        # both detection pipelines below take the DEVICE-corrected poses: the solvers were compared above, what follows compares fusion ->
        # post-process -> TP / FP (a 0.4 mm pose difference moves features by 1e-3 of a cell: not what the candidate-set equality is about)
        pw = torch.from_numpy(get_pairwise_transformation(fixed_d, 5)[None])
        runs.append((s, pw, pw))
    # a common bias shift that keeps every logit of every run away from the threshold's logit (a 1e-6 difference must not flip a candidate)
    lt = math.log(0.2 / 0.8)
    with torch.no_grad():
        logits = [model(dict(fd, pairwise_t_matrix=pw.to(DEV)))["cls_preds"].double().flatten() for _, pw, _ in runs]
        for shift in np.arange(0.0, 0.05, 0.0005):
            if all(float((l + shift - lt).abs().min()) > 2e-4 for l in logits):
                break
        else:
            raise AssertionError("no bias shift clears the threshold for all four runs")
        model.cls_head.bias += float(shift)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    for s, pw_d, pw_o in runs:
        with torch.no_grad():
            out = model(dict(fd, pairwise_t_matrix=pw_d.to(DEV)))
            ref = oracle.coalign_forward(sd, hd["model"]["args"], dict(frame, pairwise_t_matrix=pw_o))
        boxes, scores = pp.post_process(meta, {"ego": out})
        rb, rsc, info = oracle.post_process([ref], anchors, hd["postprocess"])
        assert pp.last_counts["candidates"] == len(info["cand_index"]) > 100, s
        assert boxes.shape == rb.shape and rb.shape[0] > 30, (s, boxes.shape, rb.shape)
        gt = plant_ground_truth(boxes, 9000 + int(10 * s))
        st_d, st_o = ev.new_result_stat(), ev.new_result_stat()
        for thr in ev.IOU_THRESHOLDS:
            ev.caluclate_tp_fp(boxes, scores, gt.to(DEV), st_d, thr)
            oracle.caluclate_tp_fp(rb.numpy(), rsc.numpy(), gt.numpy(), st_o, thr)
            assert st_d[thr]["gt"] == st_o[thr]["gt"] and len(st_d[thr]["tp"]) == len(st_o[thr]["tp"]), (s, thr)
            exact = st_d[thr]["tp"] == st_o[thr]["tp"] and st_d[thr]["fp"] == st_o[thr]["fp"]
            if not exact:
                # The lists are in score order; two detections whose scores differ by less than the float32 evaluation noise of the two pipelines
                # (~1e-6) may swap places.  Accept exactly that: equal TP / FP multisets inside every run of near-tied scores, nothing else.
                sc = np.asarray(st_o[thr]["score"], dtype=np.float64)
                brk = np.nonzero(np.abs(np.diff(sc)) > 5e-6)[0] + 1
                for lo, hi in zip(np.r_[0, brk], np.r_[brk, len(sc)]):
                    assert sorted(st_d[thr]["tp"][lo:hi]) == sorted(st_o[thr]["tp"][lo:hi]) and sorted(st_d[thr]["fp"][lo:hi]) == sorted(st_o[thr]["fp"][lo:hi]), (s, thr, lo, hi)
            assert abs(ev.calculate_ap(st_d, thr)[0] - oracle.calculate_ap(st_o, thr)[0]) <= (1e-12 if exact else 2e-3), (s, thr)      # same TP / FP lists; the two VOC sums add in another order
        assert sum(st_d[0.7]["tp"]) > 5, s


def test_attfusion_with_a_configured_feat_dim_that_differs_from_the_channels():
    """ScaledDotProductAttention divides by sqrt(feat_dim) of the CONFIG (att_fuse.py:36-47, fusion_in_one.py:96-136); the kernel by sqrt(C).
    A module built with another feat_dim now computes the reference's result (input rescaling, fusion.py) instead of raising."""
    import torch.nn.functional as F
    from coalign_amd.fusion import AttFusion
    g = torch.Generator().manual_seed(3)
    x = torch.randn(3, 64, 24, 40, generator=g)
    theta = torch.tensor([[[1, 0, 0], [0, 1, 0]], [[0.98, -0.1, 0.05], [0.1, 0.98, -0.02]], [[1.0, 0.05, -0.1], [-0.05, 1.0, 0.04]]], dtype=torch.float64)
    aff = torch.zeros(1, 5, 5, 2, 3, dtype=torch.float64)
    aff[0, 0, :3] = theta
    w = oracle.warp_affine_simple(x, theta, (24, 40))
    q = w.view(3, 64, -1).permute(2, 0, 1)
    want = torch.bmm(F.softmax(torch.bmm(q, q.transpose(1, 2)) / np.sqrt(16.0), -1), q).permute(1, 2, 0).reshape(3, 64, 24, 40)[0]
    got = AttFusion(16)(x.to(DEV), [3], aff.to(DEV))[0].cpu()
    assert float((got - want).abs().max()) <= 1e-5 * float(want.abs().max())
    plain = AttFusion(64)(x.to(DEV), [3], aff.to(DEV))[0].cpu()
    assert float((plain - want).abs().max()) > 1e-3 * float(want.abs().max())       # sqrt(64) and sqrt(16) really differ on this input


def test_pcdet_nms_beyond_the_device_walk_limit(monkeypatch):
    """nms_gpu / nms_normal_gpu with more boxes than coalign_pcdet_nms walks on the device (the reference has no size limit): the
    chunked host walk gives the oracle's keep list; forced at a small limit so that the test stays small."""
    from coalign_amd import pcdet
    rs = np.random.RandomState(4)
    n = 3000
    b = np.zeros((n, 7), np.float32)
    b[:, 0] = rs.uniform(-60, 60, n); b[:, 1] = rs.uniform(-30, 30, n); b[:, 3] = rs.uniform(3, 5, n); b[:, 4] = rs.uniform(1.5, 2.2, n); b[:, 5] = 1.6
    b[:, 6] = rs.uniform(-3.1, 3.1, n)
    sc = rs.uniform(0, 1, n).astype(np.float32)
    boxes, scores = torch.from_numpy(b).to(DEV), torch.from_numpy(sc).to(DEV)
    for normal, thr in ((False, 0.1), (True, 0.3)):
        fn = pcdet.nms_normal_gpu if normal else pcdet.nms_gpu
        ref = fn(boxes, scores, thr)[0]
        monkeypatch.setattr(pcdet, "PCDET_NMS_DEVICE_MAX", 1000)
        big = fn(boxes, scores, thr)[0]
        monkeypatch.setattr(pcdet, "PCDET_NMS_DEVICE_MAX", 16384)
        assert torch.equal(big, ref), normal


# ------------------------------------------------------------------------------------------------ stacked convolution tiles
_STACK_CHECK = r"""
import hashlib, os, sys, torch
import torch.nn.functional as F
from coalign_amd import ops
T = int(os.environ.get("STACK_CHECK_TERMS", "3"))
h = hashlib.sha256()
worst = 0.0
# (N, C, H, W): the detector's stage-2 / stage-3 maps at 5 and 2 agents, DAIR and LSS sizes, one image, ragged sizes, a tile height that divides H
for (N, C, H, W) in ((5, 256, 25, 88), (5, 128, 50, 176), (2, 256, 25, 88), (2, 128, 50, 176), (2, 256, 25, 63), (2, 128, 50, 126), (8, 256, 30, 30),
                     (1, 256, 25, 88), (1, 128, 50, 176), (3, 64, 24, 40), (7, 64, 6, 33), (4, 128, 48, 48), (3, 128, 27, 48)):
    g = torch.Generator().manual_seed(N * 1000 + C + H)
    x = torch.randn(N, C, H, W, generator=g).cuda(); w = (torch.randn(C, C, 3, 3, generator=g) / (C * 9) ** 0.5).cuda()
    b = torch.randn(C, generator=g).cuda(); r = torch.randn(N, C, H, W, generator=g).cuda()
    ws = ops.pack_conv3x3_emu_weight(w, T, True)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    for res, relu in ((None, False), (r, True)):
        want = ref if res is None else ref + res.double()
        want = torch.relu(want) if relu else want
        for cl in (False, True):
            got = ops.conv3x3_emu_bias_act(x, ws, b, C, res, relu, T, out_channels_last=cl)
            worst = max(worst, float((got.double() - want).abs().max() / want.abs().max()))
            h.update(got.contiguous().cpu().numpy().tobytes())
print("WORST", worst, "SHA", h.hexdigest())
sys.exit(0 if worst <= 5e-6 else 1)
"""


def test_stacked_convolution_tiles_equal_per_image_tiles_bit_for_bit():
    """conv3x3_emu.hip, VAR_STACK / VAR_NCO1 (the batch tiled as one image of N * H rows; 32-channel wavefronts): every output against the
    fp64 convolution (5e-6 of the output scale) AND bit-identical to the per-image tiles (COALIGN_EMU_STACK=0) -- the per-output sequence of
    products does not depend on the tile geometry.  Shapes with an image boundary inside a tile, at a tile edge, one image, ragged columns."""
    outs = {}
    for stack in ("5", "3", "1", "0"):     # bit 0: the 24 x 16 tiles, bit 1: the opt-in 6 x 32 / 32-channel tiles, bit 2: 4 x 8-pixel blocks on 88-wide maps (default: 5)
        r = subprocess.run([sys.executable, "-c", _STACK_CHECK], env=dict(os.environ, PYTHONPATH=ROOT, COALIGN_LAB="1", COALIGN_EMU_STACK=stack), capture_output=True, text=True,
                           timeout=900, cwd=ROOT)
        assert r.returncode == 0, (stack, r.stdout[-300:], r.stderr[-800:])
        outs[stack] = r.stdout.strip().split("SHA")[-1].strip()
    assert outs["5"] == outs["3"] == outs["1"] == outs["0"], outs
    # the fp16 2-way split (round 4): stacked tiles are its default too (COALIGN_EMU_STACK16 = 5: bits 0 and 2)
    outs = {}
    for stack in ("5", "1", "4", "0"):
        r = subprocess.run([sys.executable, "-c", _STACK_CHECK], env=dict(os.environ, PYTHONPATH=ROOT, COALIGN_LAB="1", COALIGN_EMU_STACK16=stack, STACK_CHECK_TERMS="16"),
                           capture_output=True, text=True, timeout=900, cwd=ROOT)
        assert r.returncode == 0, (stack, r.stdout[-300:], r.stderr[-800:])
        outs[stack] = r.stdout.strip().split("SHA")[-1].strip()
    assert outs["5"] == outs["1"] == outs["4"] == outs["0"], outs


_CORUN_CHECK = r"""
import sys, torch
from coalign_amd import ops
from coalign_amd.config import builtin_config
from coalign_amd.pose import normalize_pairwise_tfm
from coalign_amd.synthetic import make_frame
g = torch.Generator().manual_seed(3)
N = 5
fr = make_frame(builtin_config("opv2v_coalign"), N, pillars_per_agent=100, seed=303, noise=(0.2, 0.2))
theta = normalize_pairwise_tfm(fr["pairwise_t_matrix"].cuda(), 200, 704, 0.4)[0, 0, :N].contiguous()
xcl = [torch.randn(N, C, H, W, generator=g).cuda().contiguous(memory_format=torch.channels_last) for C, H, W in ((64, 100, 352), (128, 50, 176), (256, 25, 88))]
def conv(N_, C, H, W):
    x = torch.randn(N_, C, H, W, generator=g).cuda(); w = ops.pack_conv3x3_emu_weight((torch.randn(C, C, 3, 3, generator=g) / (C * 9) ** 0.5).cuda(), 3, True)
    b = torch.randn(C, generator=g).cuda(); r = torch.randn(N_, C, H, W, generator=g).cuda()
    return lambda: ops.conv3x3_emu_bias_act(x, w, b, C, r, True, 3)
side = torch.cuda.Stream()
ref = [t.clone() for t in ops.warp_fuse_nhwc(xcl, theta, ops.FUSE_ATT)]
bad = torch.zeros((), dtype=torch.int64, device="cuda")
for fn in (conv(5, 256, 25, 88), conv(5, 128, 50, 176), conv(5, 64, 100, 352)):
    for it in range(150):
        with torch.cuda.stream(side):
            fn()
        for a, b in zip(ops.warp_fuse_nhwc(xcl, theta, ops.FUSE_ATT), ref):
            bad += (a != b).any()
torch.cuda.synchronize()
print("DIFFERING", int(bad))
sys.exit(0 if int(bad) == 0 else 1)
"""


def test_fusion_is_not_disturbed_by_a_convolution_on_another_stream():
    """The hazard found in round 3 (profiles/round3/README.md): packed fp32 instructions of a wavefront that shares a SIMD with the matrix
    wavefronts of the 6 x 32 stacked convolution returned wrong lanes 48-63 -- one third of the fused maps differed.  The library is built
    without packed fp32 instructions; here the fusion kernel runs 450 times beside each convolution geometry (the opt-in variant included)
    and every fused map must equal the one computed alone."""
    for stack in ("3", "5"):          # the opt-in variant the finding was made with; the default set
        r = subprocess.run([sys.executable, "-c", _CORUN_CHECK], env=dict(os.environ, PYTHONPATH=ROOT, COALIGN_LAB="1", COALIGN_EMU_STACK=stack), capture_output=True, text=True,
                           timeout=600, cwd=ROOT)
        assert r.returncode == 0, (stack, r.stdout[-300:], r.stderr[-800:])


# ------------------------------------------------------------------------------------------------ pointwise layers on the split-bf16 matrix cores
@pytest.mark.parametrize("case", [(1, 64, 128, 100, 352, 1, 1), (2, 128, 128, 50, 176, 2, 1), (1, 256, 128, 25, 88, 4, 1), (5, 64, 64, 200, 704, 1, 2),
                                  (3, 64, 128, 101, 353, 1, 2), (2, 128, 256, 51, 177, 1, 2), (1, 16, 40, 7, 9, 1, 1), (1, 32, 8, 5, 3, 2, 1), (2, 48, 24, 9, 11, 4, 1)])
def test_pointwise_emu_against_float64(case):
    """coalign_pointwise_conv_emu (up-sampling heads and stride-2 skip convolutions as 3-way split bf16 products, fp32 accumulation) against
    the float64 layer: 5e-6 of the output scale -- the bound the split-bf16 3 x 3 layers are held to --, NCHW and channels-last input bit
    for bit the same, a channel slice of a larger tensor written without touching its neighbours."""
    import torch.nn.functional as F
    N, Ci, Co, H, W, up, st = case
    gen = torch.Generator(device="cpu").manual_seed(sum(case))
    x = torch.randn(N, Ci, H, W, generator=gen).to(DEV)
    b = torch.randn(Co, generator=gen).to(DEV)
    if st == 1:
        w = (torch.randn(Ci, Co, up, up, generator=gen) / Ci ** 0.5).to(DEV)
        ref = F.conv_transpose2d(x.double(), w.double(), b.double(), stride=up)
        if (Co * up * up) % 32:
            assert up == 1
            wp = ops.pack_pointwise_weight(w.reshape(Ci, Co).t().reshape(Co, Ci, 1, 1).contiguous(), False)
        else:
            wp = ops.pack_pointwise_weight(w, True)
    else:
        w = (torch.randn(Co, Ci, 1, 1, generator=gen) / Ci ** 0.5).to(DEV)
        ref = F.conv2d(x.double(), w.double(), b.double(), stride=st)
        wp = ops.pack_pointwise_weight(w, False)
    we = ops.pack_pointwise_emu_weight(wp)
    assert we.dtype == torch.int16 and we.numel() * 2 == ops.hip.lib().coalign_pointwise_emu_weight_bytes(Ci, wp.shape[1])
    scale = max(1.0, float(ref.abs().max()))
    for relu in (True, False):
        want = torch.relu(ref) if relu else ref
        got = ops.pointwise_conv(x, we, b, Co, up=up, in_stride=st, relu=relu)
        assert got.shape == want.shape
        assert float((got.double() - want).abs().max()) <= 5e-6 * scale, (case, relu)
        if Ci % 4 == 0:
            assert torch.equal(got, ops.pointwise_conv(x.contiguous(memory_format=torch.channels_last), we, b, Co, up=up, in_stride=st, relu=relu))
    f32 = ops.pointwise_conv(x, wp, b, Co, up=up, in_stride=st, relu=False)
    assert float((f32.double() - ref).abs().max()) <= 2e-5 * scale          # (the fp32 matrix-core kernel, same layer: its own bound)
    big = torch.full((N, Co + 40, ref.shape[2], ref.shape[3]), 7.0, device=DEV)
    ops.pointwise_conv(x, we, b, Co, up=up, in_stride=st, relu=True, out=big, c_off=24)
    assert float((big[:, 24:24 + Co].double() - torch.relu(ref)).abs().max()) <= 5e-6 * scale
    assert bool((big[:, :24] == 7.0).all()) and bool((big[:, 24 + Co:] == 7.0).all())
