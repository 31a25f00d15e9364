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
