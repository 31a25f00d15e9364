"""GPU parity tests: the gfx950 kernels (called through the C ABI via coalign_amd.ops) against
(a) golden vectors produced by the reference itself and (b) the CPU oracle on seeded inputs, plus
size-independent properties at the full OPV2V sizes.  Run on the MI355X box with ``pytest -m gpu``.

Tolerances (north_star: "within 1e-3 rel for fp32 BEV features, bit-exact for anchor indexing / NMS selection"):
feature tensors are compared with ``feat_close``: rtol 1e-4 plus an absolute floor of 1e-5 x the tensor's scale (measured
differences are ~1e-6), and a mean-error bound of 1e-6 x scale.
"""
import os

import numpy as np
import pytest
import torch

from oracle import coalign_oracle as oracle
from coalign_amd import ops
from coalign_amd.config import builtin_config
from coalign_amd.detector import build_model
from coalign_amd.postprocess import build_postprocessor, nms_rotated
from coalign_amd.synthetic import fill_parameters_, make_frame

pytestmark = pytest.mark.gpu
T = torch.from_numpy
DEV = "cuda:0"


def feat_close(got, ref, rtol=1e-4, what="", floor=1e-5):
    """|got - ref| <= rtol * |ref| + floor * max|ref| element-wise, and mean error <= 1e-6 * max|ref|.  north_star asks for 1e-3
    relative on fp32 BEV features; what is measured is ~1e-6, so the default is held an order of magnitude inside the requirement
    (rtol 1e-4, absolute floor 1e-5 of the tensor's scale for elements near zero, where fp32 cancellation makes a relative bound
    meaningless).  A test that needs the full 1e-3 passes it explicitly and says why."""
    got = got.detach().float().cpu()
    ref = ref if torch.is_tensor(ref) else T(np.asarray(ref))
    ref = ref.float()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    scale = float(ref.abs().max()) if ref.numel() else 1.0
    err = (got - ref).abs()
    tol = rtol * ref.abs() + floor * scale
    bad = err > tol
    assert not bool(bad.any()), f"{what}: {int(bad.sum())} / {err.numel()} outside tolerance, max err {float(err.max()):.3e} (scale {scale:.3e})"
    if err.numel():
        assert float(err.mean()) <= max(1e-6, 0.1 * floor) * max(scale, 1e-30), f"{what}: mean err {float(err.mean()):.3e} vs scale {scale:.3e}"


def pfn_state(g):
    p = "pillar_vfe.pfn_layers.0."
    return {p + "linear.weight": T(g["pfn_weight"]), p + "norm.weight": T(g["pfn_bn_weight"]), p + "norm.bias": T(g["pfn_bn_bias"]),
            p + "norm.running_mean": T(g["pfn_bn_mean"]), p + "norm.running_var": T(g["pfn_bn_var"])}


def run_pillar(vf, npts, coords, sd, margs, n_agents):
    p = "pillar_vfe.pfn_layers.0."
    nx, ny, _ = [int(v) for v in margs["point_pillar_scatter"]["grid_size"]]
    bn = tuple(sd[p + k].to(DEV) for k in ("norm.weight", "norm.bias", "norm.running_mean", "norm.running_var"))
    return ops.pillar_vfe_scatter(vf.to(DEV), npts.to(DEV), coords.to(DEV), sd[p + "linear.weight"].to(DEV), None, bn, 1e-3,
                                  True, False, margs["voxel_size"], margs["lidar_range"][:3], n_agents, ny, nx)


# ------------------------------------------------------------------------------------------------ pillar VFE + scatter
def test_pillar_golden_mini(golden):
    g = golden("model_mini.npz")
    margs = builtin_config("mini_coalign")["model"]["args"]
    feats, canvas = run_pillar(T(g["voxel_features"]), T(g["voxel_num_points"]), T(g["voxel_coords"]), pfn_state(g), margs, 5)
    feat_close(feats, g["pillar_features"], what="pillar_features vs reference")
    feat_close(canvas.reshape(-1)[::7], g["canvas_sample"], what="canvas sample vs reference")
    ref_canvas = oracle.scatter(T(g["pillar_features"]), T(g["voxel_coords"]), 5, 64, 32)
    empty = ref_canvas.abs().sum(dim=1) == 0                       # cells without a pillar: exactly zero
    assert bool((canvas.cpu().abs().sum(dim=1)[empty] == 0).all())
    feat_close(canvas, oracle.scatter(feats.cpu(), T(g["voxel_coords"]), 5, 64, 32), rtol=0, floor=0, what="scatter is a pure copy")


def test_pillar_fullsize_vs_oracle_and_reference(golden):
    g = golden("fullsize.npz")
    h = builtin_config("opv2v_coalign")
    margs = h["model"]["args"]
    model = build_model(h)
    fill_parameters_(model, seed=0, cls_bias=-1.0)
    sd = model.state_dict()
    fr = make_frame(h, 2, pillars_per_agent=int(g["pillars_per_agent"]), seed=int(g["frame_seed"]))
    pl = fr["processed_lidar"]
    feats, canvas = run_pillar(pl["voxel_features"], pl["voxel_num_points"], pl["voxel_coords"], sd, margs, 2)
    feat_close(feats[::50], g["pillar_rows"], what="pillar rows vs reference (full size)")
    assert list(canvas.shape) == list(g["canvas_shape"])
    flat = canvas.reshape(-1)
    feat_close(flat[T(g["canvas_sample_idx"]).to(DEV)], g["canvas_sample_val"], what="canvas samples vs reference (full size)")
    ref_feats = oracle.pillar_vfe(pl["voxel_features"], pl["voxel_num_points"], pl["voxel_coords"], sd, margs["voxel_size"], margs["lidar_range"])
    feat_close(feats, ref_feats, what="pillar_features vs oracle (full size)")
    # scatter property: canvas is exactly the features at occupied cells and exactly 0 elsewhere
    ref_canvas = oracle.scatter(feats.cpu(), pl["voxel_coords"], 2, 704, 200)
    assert torch.equal(canvas.cpu(), ref_canvas)


def test_pillar_edge_cases():
    h = builtin_config("mini_coalign")
    margs = h["model"]["args"]
    model = build_model(h)
    fill_parameters_(model, seed=3)
    sd = model.state_dict()
    # empty input: all-zero canvas
    feats, canvas = run_pillar(torch.zeros(0, 32, 4), torch.zeros(0, dtype=torch.int32), torch.zeros(0, 4, dtype=torch.int32), sd, margs, 2)
    assert feats.shape == (0, 64) and float(canvas.abs().sum()) == 0.0
    # ragged: 1 point, full 32 points, padded slots holding garbage-free zeros; duplicate cell -> larger row wins
    fr = make_frame(h, 2, pillars_per_agent=40, seed=5, num_points_mode="uniform")
    pl = fr["processed_lidar"]
    pl["voxel_num_points"][0] = 1
    pl["voxel_num_points"][1] = 32
    pl["voxel_features"][0, 1:] = 0
    pl["voxel_coords"][3] = pl["voxel_coords"][2]
    feats, canvas = run_pillar(pl["voxel_features"], pl["voxel_num_points"], pl["voxel_coords"], sd, margs, 2)
    ref = oracle.pillar_vfe(pl["voxel_features"], pl["voxel_num_points"], pl["voxel_coords"], sd, margs["voxel_size"], margs["lidar_range"])
    feat_close(feats, ref, what="ragged pillars")
    c = pl["voxel_coords"][3]
    assert torch.equal(canvas[int(c[0]), :, int(c[2]), int(c[3])].cpu(), feats[3].cpu())
    # scatter-only entry point gives the same canvas
    assert torch.equal(ops.scatter_to_bev(feats, pl["voxel_coords"].to(DEV), 2, 32, 64), canvas)


def test_pillar_dense_duplicates_and_unfused_route(monkeypatch):
    """Dense canvas (strips hold far more pillars than the LDS feature buffer), duplicate cells, out-of-canvas pillars:
    every pillar_features row is still produced, the canvas follows the 'larger row wins' rule, and the separate-kernel
    route gives bit-identical results."""
    h = builtin_config("mini_coalign")
    margs = h["model"]["args"]
    model = build_model(h)
    fill_parameters_(model, seed=4)
    sd = model.state_dict()
    fr = make_frame(h, 2, pillars_per_agent=1500, seed=6, num_points_mode="uniform")     # 1500 / 2048 cells occupied
    pl = fr["processed_lidar"]
    c = pl["voxel_coords"]
    c[10] = c[700]; c[11] = c[700]                      # three pillars in one cell: row 700 wins
    c[20, 3] = 64                                       # x == nx: outside the canvas
    c[21, 0] = 7                                        # agent index out of range
    feats, canvas = run_pillar(pl["voxel_features"], pl["voxel_num_points"], c, sd, margs, 2)
    ref = oracle.pillar_vfe(pl["voxel_features"], pl["voxel_num_points"], c, sd, margs["voxel_size"], margs["lidar_range"])
    feat_close(feats, ref, what="dense pillars, every row")
    keep = torch.ones(len(c), dtype=torch.bool)
    keep[[20, 21]] = False
    ref_canvas = oracle.scatter(feats.cpu()[keep], c[keep], 2, 64, 32)     # sequential indexing: last duplicate wins
    assert torch.equal(canvas.cpu(), ref_canvas)
    if os.environ.get("COALIGN_LAB") == "1" and os.environ.get("COALIGN_UNFUSED_PILLARS"):
        return        # (child process below: this WAS the separate-kernel route of the laboratory build, checked against the oracle above)
    # the separate-kernel route (laboratory build, COALIGN_UNFUSED_PILLARS=1): the same test once more in its own process
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider", "tests/test_hip_parity.py::test_pillar_dense_duplicates_and_unfused_route"],
                       env=dict(os.environ, PYTHONPATH=root, COALIGN_LAB="1", COALIGN_UNFUSED_PILLARS="1"), capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, (r.stdout[-1200:], r.stderr[-400:])


# ------------------------------------------------------------------------------------------------ warp + fusion
def test_warp_golden(golden):
    g = golden("warp.npz")
    out = ops.warp_fuse(T(g["src"]).to(DEV), T(g["theta"]).to(DEV), [6], ops.FUSE_NONE, out_hw=(16, 32))
    feat_close(out, g["warped"], what="warp_affine_simple vs reference")
    assert float(out[5].abs().max()) == 0.0


def test_fusion_golden(golden):
    g = golden("fusion.npz")
    rl, aff = [int(v) for v in g["record_len"]], T(g["affine"]).to(DEV)
    theta = torch.cat([aff[b, 0, :n] for b, n in enumerate(rl)])
    for s in range(3):
        x = T(g[f"x{s}"]).to(DEV)
        feat_close(ops.warp_fuse(x, theta, rl, ops.FUSE_ATT), g[f"att{s}"], what=f"AttFusion scale {s} vs reference")
        feat_close(ops.warp_fuse(x, theta, rl, ops.FUSE_MAX), g[f"max{s}"], what=f"MaxFusion scale {s} vs reference")


@pytest.mark.parametrize("n_agents,div", [(1, 2), (2, 1), (3, 2), (5, 1), (8, 2)])
def test_fusion_fullsize_vs_oracle(n_agents, div):
    """OPV2V feature-map sizes (div=1) for the benchmarked agent counts, half-size maps for the other kernel variants."""
    h = builtin_config("opv2v_coalign")
    fr = make_frame(h, n_agents, pillars_per_agent=10, seed=40 + n_agents, spread_yaw=180.0)
    aff = oracle.normalize_pairwise_tfm(fr["pairwise_t_matrix"], 200, 704, 0.4)
    gen = torch.Generator().manual_seed(n_agents)
    for C, H, W in ((64, 100 // div, 352 // div), (128, 50 // div, 176 // div), (256, 25 // div + 1, 88 // div)):
        x = torch.randn(n_agents, C, H, W, generator=gen)
        rl = torch.tensor([n_agents])
        theta = aff[0, 0, :n_agents].to(DEV)
        feat_close(ops.warp_fuse(x.to(DEV), theta, [n_agents], ops.FUSE_ATT), oracle.att_fuse(x, rl, aff), what=f"att N={n_agents} C={C}")
        if C == 64:
            feat_close(ops.warp_fuse(x.to(DEV), theta, [n_agents], ops.FUSE_MAX), oracle.max_fuse(x, rl, aff), what=f"max N={n_agents}")


def test_fusion_fullsize_reference_samples(golden):
    g = golden("fullsize.npz")
    gen = torch.Generator().manual_seed(int(g["fusion_gen_seed"]))
    aff = T(g["affine"]).to(DEV)
    for s, (C, H, W) in enumerate(((64, 100, 352), (128, 50, 176), (256, 25, 88))):
        x = torch.randn(2, C, H, W, generator=gen)
        y = ops.warp_fuse(x.to(DEV), aff[0, 0, :2], [2], ops.FUSE_ATT)
        assert list(y.shape) == list(g[f"att{s}_shape"])
        feat_close(y.reshape(-1)[::211], g[f"att{s}_sample"], what=f"att full-size scale {s} vs reference samples")


def test_fusion_properties():
    gen = torch.Generator().manual_seed(9)
    ego = torch.randn(1, 64, 100, 352, generator=gen).to(DEV)
    ident = torch.tensor([[1.0, 0, 0], [0, 1, 0]], dtype=torch.float64, device=DEV)
    # N identical copies at identity pose: attention returns the ego map itself
    x = ego.repeat(4, 1, 1, 1)
    y = ops.warp_fuse(x, ident.repeat(4, 1, 1), [4], ops.FUSE_ATT)
    # (the identity warp itself is only exact to an ulp of the pixel coordinate: ix = j +- 3e-5 mixes in ~1e-4 of a neighbour)
    # an "identity" theta still leaves sub-pixel residue in float32 (ix = j +- 1e-5), so the bilinear tap mixes ~1e-5 of the neighbour
    # in -- exactly like F.grid_sample does: hold the property to 1e-4 of the map's scale, not to rounding
    feat_close(y, ego.cpu(), what="identity pose, identical agents", floor=1e-4)
    # permuting the non-ego agents changes nothing but the fp summation order
    others = torch.randn(3, 64, 100, 352, generator=gen).to(DEV)
    th = torch.tensor([[[0.9, -0.1, 0.05], [0.3, 0.95, -0.02]], [[1.0, 0.0, 0.3], [0.0, 1.0, 0.1]], [[-1.0, 0.02, 0.0], [-0.1, -1.0, 0.0]]],
                      dtype=torch.float64, device=DEV)
    a = ops.warp_fuse(torch.cat([ego, others]), torch.cat([ident[None], th]), [4], ops.FUSE_ATT)
    perm = [2, 0, 1]
    b = ops.warp_fuse(torch.cat([ego, others[perm]]), torch.cat([ident[None], th[perm]]), [4], ops.FUSE_ATT)
    feat_close(a, b.cpu(), rtol=1e-5, what="agent permutation invariance")
    # batch of two frames == the two frames fused separately, bit for bit
    xb = torch.cat([ego, others[:1], ego, others[1:]])
    thb = torch.cat([ident[None], th[:1], ident[None], th[1:]])
    both = ops.warp_fuse(xb, thb, [2, 3], ops.FUSE_ATT)
    assert torch.equal(both[0], ops.warp_fuse(xb[:2], thb[:2], [2], ops.FUSE_ATT)[0])
    assert torch.equal(both[1], ops.warp_fuse(xb[2:], thb[2:], [3], ops.FUSE_ATT)[0])


# ------------------------------------------------------------------------------------------------ decode / NMS / post_process
def _margin_ok(cls, thr, ulps=16):
    p = torch.sigmoid(cls.double())
    return bool(((p - thr).abs() > ulps * 6e-8).all())


def test_decode_candidates_vs_oracle(golden):
    g = golden("postprocess.npz")
    cls, reg, dirp, anchors = T(g["i_cls"]), T(g["i_reg"]), T(g["i_dir"]), T(g["anchors"])
    assert _margin_ok(cls, 0.2)
    idx, boxes7, scores, corners = oracle.decode_candidates(cls, reg, dirp, anchors, 0.2, "hwl")
    A, H, W = cls.shape[1:]
    buf = ops.DecodeBuffers(A * H * W, A, H, W, 1000, DEV)
    ops.anchor_decode(buf, 0, cls.to(DEV), reg.to(DEV), dirp.to(DEV), anchors.reshape(-1, 7).float().to(DEV), 0.2, 0.7853, 2, "hwl", None)
    n = int(buf.counts[1].item())
    assert n == len(idx) and torch.equal(buf.cand_index[:n].cpu().long(), idx)          # selection + order bit exact
    np.testing.assert_allclose(buf.cand_score[:n].cpu().numpy(), scores.numpy(), rtol=3e-7, atol=0)
    np.testing.assert_allclose(buf.cand_box7[:n].cpu().numpy(), boxes7.numpy(), rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(buf.cand_corners[:n].cpu().numpy(), corners.numpy(), rtol=2e-6, atol=4e-6)
    keep = torch.logical_and(oracle.remove_large_pred_bbx(corners), oracle.remove_bbx_abnormal_z(corners))
    assert torch.equal(buf.cand_keep[:n].cpu().bool(), keep)


def _run_post(pp, agents, anchors):
    data = {f"a{i}": {"transformation_matrix": ag.get("transformation_matrix", torch.eye(4)), "anchor_box": anchors} for i, ag in enumerate(agents)}
    outd = {f"a{i}": {k: v.to(DEV) for k, v in ag.items() if k.endswith("_preds")} for i, ag in enumerate(agents)}
    return pp.post_process(data, outd)


def test_post_process_golden(golden):
    g = golden("postprocess.npz")
    pp = build_postprocessor(builtin_config("mini_coalign")["postprocess"], False)
    anchors = T(g["anchors"])
    boxes, scores = _run_post(pp, [dict(cls_preds=T(g["i_cls"]), reg_preds=T(g["i_reg"]), dir_preds=T(g["i_dir"]))], anchors)
    assert boxes.shape == g["i_boxes"].shape
    np.testing.assert_allclose(scores.cpu().numpy(), g["i_scores"], rtol=3e-7, atol=0)
    np.testing.assert_allclose(boxes.cpu().numpy(), g["i_boxes"], rtol=2e-6, atol=1e-5)
    late = [dict(cls_preds=T(g["l_cls0"]), reg_preds=T(g["l_reg0"]), dir_preds=T(g["l_dir0"]), transformation_matrix=torch.eye(4)),
            dict(cls_preds=T(g["l_cls1"]), reg_preds=T(g["l_reg1"]), dir_preds=T(g["l_dir1"]), transformation_matrix=T(g["l_T1"]))]
    boxes, scores = _run_post(pp, late, anchors)
    assert boxes.shape == g["l_boxes"].shape
    np.testing.assert_allclose(scores.cpu().numpy(), g["l_scores"], rtol=3e-7, atol=0)
    np.testing.assert_allclose(boxes.cpu().numpy(), g["l_boxes"], rtol=2e-6, atol=1e-5)
    b, s = _run_post(pp, [dict(cls_preds=torch.full((1, 2, 16, 32), -9.0), reg_preds=T(g["i_reg"]), dir_preds=T(g["i_dir"]))], anchors)
    assert b is None and s is None


@pytest.mark.parametrize("tag", ["small", "mid", "over1000"])
def test_nms_golden(golden, tag):
    g = golden("nms.npz")
    keep = nms_rotated(T(g[f"{tag}_corners"]).to(DEV), T(g[f"{tag}_scores"]).to(DEV), 0.15)
    assert keep.dtype == np.int32 and np.array_equal(keep, g[f"{tag}_keep"])


def test_nms_edge_cases_and_properties(golden):
    g = golden("nms.npz")
    assert nms_rotated(torch.zeros(0, 8, 3, device=DEV), torch.zeros(0, device=DEV), 0.15).shape == (0,)
    q = T(g["over1000_corners"][:50, :4, :2].copy()).to(DEV)
    assert np.array_equal(nms_rotated(q, T(g["over1000_scores"][:50]).to(DEV), 0.15), g["quad_keep"])
    # idempotence: NMS of the kept set keeps everything, in the same order
    c, s = T(g["mid_corners"]).to(DEV), T(g["mid_scores"]).to(DEV)
    k1 = nms_rotated(c, s, 0.15)
    k2 = nms_rotated(c[k1.astype(np.int64)], s[k1.astype(np.int64)], 0.15)
    assert np.array_equal(k2, np.arange(len(k1)))
    # ties: defined as larger index first (oracle.score_order), duplicates of one box collapse to one
    c4 = c[:1].repeat(4, 1, 1)
    assert np.array_equal(nms_rotated(c4, torch.full((4,), 0.5, device=DEV), 0.15), np.array([3], dtype=np.int32))
    # large random set against the oracle
    rs = np.random.RandomState(5)
    K = 3000
    b7 = np.zeros((K, 7), np.float32)
    b7[:, 0] = rs.uniform(-100, 100, K); b7[:, 1] = rs.uniform(-40, 40, K); b7[:, 2] = -1; b7[:, 3] = 1.5
    b7[:, 4] = rs.uniform(1.4, 2.2, K); b7[:, 5] = rs.uniform(3, 5.5, K); b7[:, 6] = rs.uniform(-3.2, 3.2, K)
    corners = oracle.boxes_to_corners_3d(T(b7), "hwl")
    sc = T(rs.uniform(0.2, 1, K).astype(np.float32))
    assert np.array_equal(nms_rotated(corners.to(DEV), sc.to(DEV), 0.15), oracle.nms_rotated(corners.numpy(), sc.numpy(), 0.15))


def test_pcdet_iou_bev_vs_oracle():
    rs = np.random.RandomState(2)
    a = np.zeros((64, 7), np.float32); b = np.zeros((48, 7), np.float32)
    for m in (a, b):
        m[:, 0] = rs.uniform(-6, 6, len(m)); m[:, 1] = rs.uniform(-6, 6, len(m)); m[:, 3] = rs.uniform(3, 5, len(m))
        m[:, 4] = rs.uniform(1.5, 2.2, len(m)); m[:, 5] = 1.5; m[:, 6] = rs.uniform(-3.2, 3.2, len(m))
    got = ops.boxes_iou_bev(T(a).to(DEV), T(b).to(DEV)).cpu().numpy()
    lib = oracle._nms_lib()
    import ctypes
    lib.oracle_pcdet_iou.restype = ctypes.c_float
    lib.oracle_pcdet_iou.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    ref = np.array([[lib.oracle_pcdet_iou(a[i].ctypes.data, b[j].ctypes.data) for j in range(len(b))] for i in range(len(a))], np.float32)
    # round 6: cos / sin / atan2 of this row are float64 evaluations rounded once to float32 on BOTH sides (csrc/nms.hip trig_*, oracle/rotated_nms.c trig_*), every
    # other operation is a single IEEE float32 operation in the same order (-ffp-contract=off): the matrices are compared in ULPS, not to a tolerance.  Bound 4 ulp
    # (VERDICT r05); expected and printed: bit-equal everywhere (a double-rounding case of the device's vs glibc's float64 functions, ~2^-29 per call, would show as 1 ulp).
    ulp = np.abs(got.view(np.int32).astype(np.int64) - ref.view(np.int32).astype(np.int64))
    print(f"\npcdet IoU matrix vs the restatement: {int((ulp == 0).sum())} / {ulp.size} bit-equal, max {int(ulp.max())} ulp, {int((ref > 0).sum())} overlapping pairs")
    assert (ref > 0).sum() > 200
    assert ulp.max() <= 4
    # ... and a larger, denser set through the overlap-area entry point (boxes_overlap_bev: iou3d_nms_kernel.cu:236-249)
    n = 400
    big = np.zeros((n, 7), np.float32)
    big[:, 0] = rs.uniform(-12, 12, n); big[:, 1] = rs.uniform(-8, 8, n); big[:, 3] = rs.uniform(2, 5.5, n)
    big[:, 4] = rs.uniform(1.2, 2.4, n); big[:, 5] = 1.5; big[:, 6] = rs.uniform(-7, 7, n)
    big[::7, 6] = np.float32(np.pi / 2); big[3::11, 6] = 0.0                     # axis-aligned pairs: the nearly-parallel branch of the edge intersection
    got2 = ops.boxes_iou_bev(T(big).to(DEV), T(big).to(DEV)).cpu().numpy()
    ref2 = np.array([[lib.oracle_pcdet_iou(big[i].ctypes.data, big[j].ctypes.data) for j in range(n)] for i in range(n)], np.float32)
    ulp2 = np.abs(got2.view(np.int32).astype(np.int64) - ref2.view(np.int32).astype(np.int64))
    print(f"400 x 400: {int((ulp2 == 0).sum())} / {ulp2.size} bit-equal, max {int(ulp2.max())} ulp, {int((ref2 > 0).sum())} overlapping pairs")
    assert ulp2.max() <= 4 and (ulp2 == 0).mean() > 0.9999


# ------------------------------------------------------------------------------------------------ whole model
def _batch_from(g):
    return {"processed_lidar": {"voxel_features": T(g["voxel_features"]), "voxel_coords": T(g["voxel_coords"]),
                                "voxel_num_points": T(g["voxel_num_points"])},
            "record_len": T(g["record_len"]), "pairwise_t_matrix": T(g["pairwise_t_matrix"])}


def test_model_mini_vs_reference(golden, conv_mode):
    from coalign_amd.detector import to_device
    g = golden("model_mini.npz")
    h = builtin_config("mini_coalign")
    model = build_model(h)
    fill_parameters_(model, seed=int(g["fill_seed"]), cls_bias=float(g["cls_bias"]))
    model = model.to(DEV).eval()
    with torch.no_grad():
        batch = to_device(_batch_from(g), DEV)
        feats, aff = model.encode(batch)
        out = model(batch)
    np.testing.assert_allclose(aff.cpu().numpy(), g["normalized_affine"], rtol=0, atol=1e-15)
    for s in range(3):
        feat_close(feats[s].reshape(-1)[::5], g[f"feat{s}_sample"], what=f"backbone scale {s} vs reference")
        feat_close(model.fusion_net[s](feats[s], batch["record_len"], aff), g[f"fused{s}"], what=f"fused scale {s} vs reference")
    for k in ("cls_preds", "reg_preds", "dir_preds"):
        feat_close(out[k], g[k], what=k + " vs reference")


def test_model_fullsize_vs_oracle_and_postprocess(conv_mode):
    """cfg 2 (OPV2V CoAlign, 2 agents, full 704x200 canvas): HIP path vs the CPU oracle end to end,
    then identical detections after post-processing."""
    h = builtin_config("opv2v_coalign")
    model = build_model(h)
    fill_parameters_(model, seed=0, cls_bias=-1.5)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    frame = make_frame(h, 2, pillars_per_agent=6000, seed=77, noise=(0.2, 0.2))
    with torch.no_grad():
        ref = oracle.coalign_forward(sd, h["model"]["args"], frame)
        from coalign_amd.detector import to_device
        out = model.to(DEV).eval()(to_device(frame, DEV))
    for k in ("cls_preds", "reg_preds", "dir_preds"):
        feat_close(out[k], ref[k], what=k + " vs oracle (full size)")
    pp = build_postprocessor(h["postprocess"], False)
    anchors = T(pp.generate_anchor_box())
    boxes, scores = pp.post_process({"ego": {"transformation_matrix": torch.eye(4), "anchor_box": anchors}}, {"ego": out})
    # feed the oracle the DEVICE logits so that the comparison isolates post-processing (selection must be identical)
    cpu_out = {k: v.cpu() for k, v in out.items()}
    if not _margin_ok(cpu_out["cls_preds"], 0.2):
        pytest.skip("a logit sits within a few ulp of the threshold for this seed")
    rb, rs_, info = oracle.post_process([cpu_out], anchors, h["postprocess"])
    assert pp.last_counts["candidates"] == len(info["cand_index"])
    assert boxes.shape == rb.shape
    np.testing.assert_allclose(scores.cpu().numpy(), rs_.numpy(), rtol=3e-7, atol=0)
    np.testing.assert_allclose(boxes.cpu().numpy(), rb.numpy(), rtol=2e-6, atol=2e-5)


# ------------------------------------------------------------------------------------------------ other configs / API level
def test_late_fusion_pointpillar_vs_reference_and_oracle(golden, conv_mode):
    """cfg 1: single-agent PointPillar (plain BaseBEVBackbone) per cav + one merged post-process."""
    from coalign_amd.detector import to_device
    from coalign_amd.inference import inference_late_fusion
    g = golden("late_mini.npz")
    h = builtin_config("mini_pointpillar_late")
    model = build_model(h)
    fill_parameters_(model, seed=0, cls_bias=-1.0)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    model = model.to(DEV).eval()
    batch = {"processed_lidar": {"voxel_features": T(g["voxel_features"]), "voxel_coords": T(g["voxel_coords"]),
                                 "voxel_num_points": T(g["voxel_num_points"])}}
    with torch.no_grad():
        out = model(to_device(batch, DEV))
    for k in ("cls_preds", "reg_preds", "dir_preds"):
        feat_close(out[k], g[k], what=f"PointPillar {k} vs reference")
    ref = oracle.pointpillar_forward(sd, h["model"]["args"], batch)
    for k in ("cls_preds", "reg_preds", "dir_preds"):
        feat_close(out[k], ref[k], what=f"PointPillar {k} vs oracle")
    # two cavs through the late-fusion driver == oracle post-process on the same (device) logits
    pp = build_postprocessor(h["postprocess"], False)
    anchors = T(pp.generate_anchor_box())
    fr2 = make_frame(h, 1, pillars_per_agent=150, seed=13)
    T1 = torch.tensor([[0.97, -0.24, 0, 2.0], [0.24, 0.97, 0, -1.0], [0, 0, 1, 0.1], [0, 0, 0, 1]])
    data = {"a": dict(to_device(batch, DEV), transformation_matrix=torch.eye(4), anchor_box=anchors),
            "b": dict(to_device({"processed_lidar": fr2["processed_lidar"]}, DEV), transformation_matrix=T1, anchor_box=anchors)}
    res = inference_late_fusion(data, model, pp)
    with torch.no_grad():
        outs = [{k: v.cpu() for k, v in model(data[c]).items()} for c in ("a", "b")]
    if not all(_margin_ok(o["cls_preds"], 0.2) for o in outs):
        pytest.skip("logit within a few ulp of the threshold")
    outs[0]["transformation_matrix"], outs[1]["transformation_matrix"] = torch.eye(4), T1
    rb, rs_, _ = oracle.post_process(outs, anchors, h["postprocess"])
    if rb is None:
        assert res["pred_box_tensor"] is None
    else:
        assert res["pred_box_tensor"].shape == rb.shape
        np.testing.assert_allclose(res["pred_box_tensor"].cpu().numpy(), rb.numpy(), rtol=2e-6, atol=2e-5)


def test_dairv2x_geometry_fusion_and_pillars_vs_oracle():
    """cfg 4: DAIR-V2X canvas 504x200 (z range -3.5..1.5, voxel height 5), vehicle + road-side unit at ~170 deg, pose noise."""
    h = builtin_config("dairv2x_coalign")
    margs = h["model"]["args"]
    model = build_model(h)
    fill_parameters_(model, seed=1)
    sd = model.state_dict()
    fr = make_frame(h, 2, pillars_per_agent=4000, seed=5, noise=(0.2, 0.2), infra_agent=True)
    pl = fr["processed_lidar"]
    feats, canvas = run_pillar(pl["voxel_features"], pl["voxel_num_points"], pl["voxel_coords"], sd, margs, 2)
    assert canvas.shape == (2, 64, 200, 504)
    ref = oracle.pillar_vfe(pl["voxel_features"], pl["voxel_num_points"], pl["voxel_coords"], sd, margs["voxel_size"], margs["lidar_range"])
    feat_close(feats, ref, what="DAIR pillar features")
    assert torch.equal(canvas.cpu(), oracle.scatter(feats.cpu(), pl["voxel_coords"], 2, 504, 200))
    aff = oracle.normalize_pairwise_tfm(fr["pairwise_t_matrix"], 200, 504, 0.4)
    gen = torch.Generator().manual_seed(3)
    rl = torch.tensor([2])
    for C, H, W in ((64, 100, 252), (128, 50, 126), (256, 25, 63)):        # W = 126, 63 are not multiples of 4: generic route
        x = torch.randn(2, C, H, W, generator=gen)
        feat_close(ops.warp_fuse(x.to(DEV), aff[0, 0, :2].to(DEV), [2], ops.FUSE_ATT), oracle.att_fuse(x, rl, aff), what=f"DAIR att C={C}")


def test_lss_fusion_shapes_vs_oracle():
    """cfg 5: only the fusion step of the LSS camera model is on the hot path: 8 agents, 64@120^2, 128@60^2, 256@30^2."""
    import yaml
    from coalign_amd.config import CONFIG_DIR
    cfg = yaml.safe_load(open(CONFIG_DIR + "/lss_coalign_fusion.yaml"))["fusion_only"]
    H0, W0 = cfg["bev_hw"]
    rs = np.random.RandomState(8)
    poses = [np.zeros(6)] + [np.array([rs.uniform(-15, 15), rs.uniform(-15, 15), 0, 0, rs.uniform(-180, 180), 0]) for _ in range(7)]
    pair = torch.from_numpy(oracle.pairwise_transformation(poses, 8))[None]
    aff = oracle.normalize_pairwise_tfm(pair, H0, W0, cfg["discrete_ratio"], cfg["downsample_rate"])
    gen = torch.Generator().manual_seed(4)
    rl = torch.tensor([8])
    from coalign_amd.fusion import AttFusion, MaxFusion
    for C, H, W in cfg["scales"]:
        x = torch.randn(8, C, H, W, generator=gen) * 0.5
        feat_close(AttFusion(C)(x.to(DEV), rl, aff.to(DEV)), oracle.att_fuse(x, rl, aff), what=f"LSS att C={C}")
    x = torch.randn(8, 64, 120, 120, generator=gen)
    feat_close(MaxFusion()(x.to(DEV), rl, aff.to(DEV)), oracle.max_fuse(x, rl, aff), what="LSS max")


def test_api_level_functions(golden):
    """The opencood-named entry points (not just the raw ops): warp_affine_simple, warp_feature, regroup, NaiveCompressor."""
    from coalign_amd import fusion
    from coalign_amd.backbone import NaiveCompressor
    import coalign_amd.backbone as bb
    g = golden("warp.npz")
    out = fusion.warp_affine_simple(T(g["src"]).to(DEV), T(g["theta"]).to(DEV), (16, 32))
    feat_close(out, g["warped"], what="warp_affine_simple")
    f = golden("fusion.npz")
    rl, aff = T(f["record_len"]), T(f["affine"]).to(DEV)
    x = T(f["x0"]).to(DEV)
    parts = fusion.regroup(x, rl)
    assert [p.shape[0] for p in parts] == [3, 2]
    w = fusion.warp_feature(x, rl, aff)
    ref = torch.cat([oracle.warp_affine_simple(T(f["x0"])[:3], T(f["affine"])[0, 0, :3], (8, 16)),
                     oracle.warp_affine_simple(T(f["x0"])[3:], T(f["affine"])[1, 0, :2], (8, 16))])
    feat_close(w, ref, what="warp_feature")
    # fused conv epilogue == the plain torch.nn sequence (BatchNorm folded)
    comp = NaiveCompressor(64, 4)
    fill_parameters_(comp, seed=2)
    comp = comp.to(DEV).eval()
    xin = torch.randn(2, 64, 40, 48, device=DEV)
    with torch.no_grad():
        fast = comp(xin)
        bb.FAST_INFERENCE = False
        try:
            plain = comp(xin)
        finally:
            bb.FAST_INFERENCE = True
    feat_close(fast, plain.cpu(), what="NaiveCompressor folded vs plain")


def test_evaluation_on_device_vs_reference(golden):
    """next-2: AP through the device IoU matrix == the reference's eval_utils numbers; IoU matrix bit-equal to the oracle's."""
    from coalign_amd import evaluation as ev
    g = golden("eval.npz")
    stat = ev.new_result_stat()
    for f in range(3):
        det, sc, gt = T(g[f"det{f}"]).to(DEV), T(g[f"score{f}"]).to(DEV), T(g[f"gt{f}"]).to(DEV)
        assert np.array_equal(ops.iou_rotated_matrix(det, gt).cpu().numpy(), oracle.iou_matrix(g[f"det{f}"], g[f"gt{f}"]))
        for thr in (0.3, 0.5, 0.7):
            ev.caluclate_tp_fp(det, sc, gt, stat, thr)
    ev.caluclate_tp_fp(None, None, T(g["gt2"]), stat, 0.7)
    for thr in (0.3, 0.5, 0.7):
        tag = str(int(thr * 100))
        assert stat[thr]["tp"] == list(g[f"tp{tag}"]) and stat[thr]["fp"] == list(g[f"fp{tag}"]) and stat[thr]["gt"] == int(g[f"gtn{tag}"])
        assert abs(ev.calculate_ap(stat, thr)[0] - float(g[f"ap{tag}"])) < 1e-12


def test_synthetic_inference_loop_ap_vs_oracle():
    """The reference's inference loop on seeded frames with planted ground truth (tests/inference_synthetic.py): the
    gfx950 path and the CPU oracle give the same TP/FP sequence and the same AP at every IoU threshold."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("inference_synthetic", os.path.join(os.path.dirname(__file__), "inference_synthetic.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rep = mod.run("mini_coalign", frames=4, agents=3, pillars=150, check_oracle=True)
    assert rep["detections"] > 8, rep
    assert rep["tp_fp_identical"], rep
    for k, v in rep["hip"].items():
        assert abs(v - rep["oracle"][k]) < 1e-12, rep
    assert 0.0 < rep["hip"]["ap30"] <= 1.0


def test_synthetic_inference_loop_ap_vs_oracle_full_size():
    """The same loop at the benchmarked geometry (opv2v_coalign: 704 x 200 canvas, three-scale backbone, two cavs with 3000 pillars each, ~800
    detections per frame): decode + NMS + range filter + matching + AP of the gfx950 path equal the CPU oracle's chain on the same head outputs,
    sequence for sequence, at every IoU threshold.  (The head outputs themselves against the oracle's model at this size:
    tests/test_pipeline_gpu.py::test_benchmarked_frame_end_to_end_vs_oracle, the cfg 1 / cfg 4 tests.)"""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("inference_synthetic", os.path.join(os.path.dirname(__file__), "inference_synthetic.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    # head scale / bias chosen for ~1 % of the anchors above the score threshold and no saturated sigmoid (at the mini harness's scale the
    # full-size maps give logits up to 85: 10 000 scores of exactly 1.0, whose order is the sort algorithm's -- in the reference too)
    rep = mod.run("opv2v_coalign", frames=2, agents=2, pillars=3000, check_oracle=True, oracle_heads="device", head_scale=(0.01, 0.005), cls_bias=-5.0)
    assert rep["detections"] > 60 and rep["candidates"] > 500, rep
    assert rep["counts_identical"], rep
    if rep["tied_candidate_scores"] == 0:
        assert rep["tp_fp_identical"], rep
        for k, v in rep["hip"].items():
            assert abs(v - rep["oracle"][k]) < 1e-12, rep
    else:
        for k, v in rep["hip"].items():
            assert abs(v - rep["oracle"][k]) < 1e-3, rep


# ------------------------------------------------------------------------------------------------ points -> pillars (next-1)
OPV2V_RANGE, OPV2V_VOXEL = [-140.8, -40, -3, 140.8, 40, 1], [0.4, 0.4, 4]


def _voxelize_ref(clouds, max_points, max_voxels, ego=False, frange=None, rng=OPV2V_RANGE, vs=OPV2V_VOXEL):
    per = []
    for c in clouds:
        if frange is not None:
            c = oracle.mask_points_by_range(c, frange)
        if ego:
            c = oracle.mask_ego_points(c)
        per.append(oracle.points_to_voxel(c, vs, rng, max_points, max_voxels))
    return oracle.collate_voxels(per), [len(p[0]) for p in per]


def _voxelize_hip(clouds, max_points, max_voxels, ego=False, frange=None, rng=OPV2V_RANGE, vs=OPV2V_VOXEL):
    off = np.concatenate([[0], np.cumsum([len(c) for c in clouds])]).tolist()
    pts = T(np.concatenate(clouds).astype(np.float32)).to(DEV)
    v, c, n, counts = ops.voxelize(pts, off, vs, rng, max_points, max_voxels, ego_filter=ego, filter_range=frange)
    counts = counts.cpu().tolist()
    m = counts[-1]
    return (v[:m].cpu().numpy(), c[:m].cpu().numpy(), n[:m].cpu().numpy()), counts[:-1]


def _same_voxels(got, ref):
    (gv, gc, gn), gcounts = got
    (rv, rc, rn), rcounts = ref
    assert gcounts == rcounts
    assert np.array_equal(gc, rc) and np.array_equal(gn, rn)
    assert np.array_equal(gv.view(np.uint32), rv.view(np.uint32))       # bit-exact copies, zero padding included


def test_voxelize_full_sweeps_vs_oracle():
    """Five synthetic 64-beam sweeps (~75 k points each, pillars holding from 1 to several hundred points) in one call:
    voxel numbering, the first-32-in-point-order selection, coords and counts are bit-identical to the sequential CPU loop;
    shuffled order (what the reference feeds) likewise; and the run is repeatable."""
    from coalign_amd.synthetic import make_point_cloud
    clouds = [make_point_cloud(40 + i) for i in range(5)]
    ref = _voxelize_ref(clouds, 32, 70000, ego=True)
    got = _voxelize_hip(clouds, 32, 70000, ego=True)
    _same_voxels(got, ref)
    assert max(ref[0][2]) == 32 and sum(ref[1]) > 20000
    _same_voxels(_voxelize_hip(clouds, 32, 70000, ego=True), got)
    rs = np.random.RandomState(0)
    shuffled = [c[rs.permutation(len(c))] for c in clouds[:2]]
    _same_voxels(_voxelize_hip(shuffled, 32, 70000, ego=True), _voxelize_ref(shuffled, 32, 70000, ego=True))
    # late-fusion order of filters (range mask, then ego mask; late_fusion_dataset.py:157-170) and a tighter filter range
    fr = [-70.4, -40, -3, 70.4, 40, 1]
    _same_voxels(_voxelize_hip(clouds[:2], 32, 70000, ego=True, frange=fr, rng=fr), _voxelize_ref(clouds[:2], 32, 70000, ego=True, frange=fr, rng=fr))


def test_voxelize_edge_cases(golden):
    g = golden("points.npz")
    cloud = g["cloud"]
    rs = np.random.RandomState(5)
    blob = rs.normal([5, 3, -1, 0.5], [0.5, 0.5, 0.3, 0.1], (6000, 4)).astype(np.float32)       # cells with 100s of points
    spike = np.tile(np.float32([[20.1, 10.1, -1, 0.5]]), (5000, 1)) + rs.uniform(0, 0.1, (5000, 4)).astype(np.float32)  # one cell, 5000 points
    # (a) boundary points, the reference's own filters pinned through golden masks
    (gv, gc, gn), _ = _voxelize_hip([cloud], 32, 70000, ego=True)
    (rv, rc, rn) = oracle.points_to_voxel(g["ego_masked"], OPV2V_VOXEL, OPV2V_RANGE, 32, 70000)
    assert np.array_equal(gv, rv) and np.array_equal(gc[:, 1:], rc) and np.array_equal(gn, rn)
    (gv, gc, gn), _ = _voxelize_hip([cloud], 32, 70000, frange=OPV2V_RANGE)
    (rv, rc, rn) = oracle.points_to_voxel(g["range_masked"], OPV2V_VOXEL, OPV2V_RANGE, 32, 70000)
    assert np.array_equal(gv, rv) and np.array_equal(gc[:, 1:], rc) and np.array_equal(gn, rn)
    # (b) max_voxels reached: later cells dropped, later points of open cells still taken; small max_points; ragged clouds incl. empty
    mixed = np.concatenate([cloud, blob, spike])[rs.permutation(len(cloud) + 11000)]
    for max_points, max_voxels in ((32, 70000), (5, 300), (1, 7), (64, 50)):
        clouds = [mixed, mixed[:0], spike, mixed[:1], blob[:1025]]
        _same_voxels(_voxelize_hip(clouds, max_points, max_voxels), _voxelize_ref(clouds, max_points, max_voxels))
    # (c) 3-D grid (SECOND-style voxels, nz > 1) and an all-outside cloud
    vs3, rng3 = [0.1, 0.1, 0.1], [-12.8, -6.4, -3, 12.8, 6.4, 1]
    near = np.concatenate([blob, cloud[:300]])
    _same_voxels(_voxelize_hip([near, blob], 5, 16000, rng=rng3, vs=vs3), _voxelize_ref([near, blob], 5, 16000, rng=rng3, vs=vs3))
    far = cloud.copy(); far[:, 0] += 1000
    (gv, gc, gn), counts = _voxelize_hip([far, far], 32, 70000)
    assert counts == [0, 0] and len(gv) == 0
    # (d) argument errors are reported, not executed
    from coalign_amd import hip
    with pytest.raises(hip.CoalignHipError):
        ops.voxelize(T(cloud).to(DEV), [0, len(cloud)], OPV2V_VOXEL, OPV2V_RANGE, 65, 70000)
    with pytest.raises(hip.CoalignHipError):
        ops.voxelize(T(cloud), [0, len(cloud)], OPV2V_VOXEL, OPV2V_RANGE, 32, 70000)


def test_voxelize_feeds_the_detector(conv_mode):
    """points -> pillars -> PillarVFE -> canvas on the device equals the oracle chain on the CPU."""
    from coalign_amd.preprocess import build_preprocessor
    from coalign_amd.synthetic import make_point_cloud
    h = builtin_config("opv2v_coalign")
    pre = build_preprocessor(h["preprocess"], train=False)
    clouds = [make_point_cloud(60 + i) for i in range(2)]
    out = pre.preprocess_clouds(clouds, ego_filter=True)
    (rv, rc, rn), counts = _voxelize_ref(clouds, 32, 70000, ego=True)
    assert out["voxel_counts"] == counts
    assert np.array_equal(out["voxel_features"].cpu().numpy(), rv) and np.array_equal(out["voxel_coords"].cpu().numpy(), rc)
    single = pre.preprocess(clouds[0])
    assert single["voxel_coords"].shape[1] == 3 and single["voxel_features"].shape[0] >= counts[0]   # no ego filter here
    coll = pre.collate_batch([single, single])
    assert coll["voxel_coords"].shape[1] == 4 and int(coll["voxel_coords"][:, 0].max()) == 1
    model = build_model(h)
    fill_parameters_(model, seed=0)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    model = model.to(DEV).eval()
    bd = {"voxel_features": out["voxel_features"], "voxel_coords": out["voxel_coords"], "voxel_num_points": out["voxel_num_points"],
          "record_len": torch.tensor([2])}
    with torch.no_grad():
        canvas = model.scatter(model.pillar_vfe(bd))["spatial_features"]
    margs = h["model"]["args"]
    pf = oracle.pillar_vfe(T(rv), T(rn), T(rc), sd, margs["voxel_size"], margs["lidar_range"])
    ref_canvas = oracle.scatter(pf, T(rc), 2, 704, 200)
    feat_close(canvas, ref_canvas, what="canvas from device voxels")


# ------------------------------------------------------------------------------------------------ box alignment (next-3)
def test_pose_graph_solver_vs_reference_graphs_and_oracle(golden):
    """coalign_pose_graph_optimize on the graphs the reference built (recorded from its g2o calls): agents end at the oracle
    LM's solution -- which the CPU suite checks against scipy's independent least-squares optimum -- for SE2 and point
    landmarks, with / without information matrices, dropped clusters, an agent without boxes; all nine in ONE launch."""
    from coalign_amd import box_align
    from tests.test_oracle_golden import BOX_ALIGN_CASES
    g = golden("box_align.npz")
    tags = [t for t in BOX_ALIGN_CASES if int(g[f"{t}_solved"])]
    graphs = [box_align.PoseGraph(g[f"{t}_vertices"], g[f"{t}_kinds"], g[f"{t}_edge_agent"], g[f"{t}_edge_landmark"], g[f"{t}_edge_meas"],
                                  g[f"{t}_edge_info"], int(g[f"{t}_len"].shape[0])) for t in tags]
    solved, stats = box_align.optimize_pose_graphs(graphs)
    again, _ = box_align.optimize_pose_graphs(graphs)
    for t, graph, x, y, st in zip(tags, graphs, solved, again, stats):
        ref = g[f"{t}_solution"]
        n = graph.n_agents
        edges = (graph.edge_agent, graph.edge_landmark, graph.edge_meas, graph.edge_info)
        assert np.array_equal(x, y), t                                              # deterministic
        assert np.array_equal(x[0], graph.vertices[0]), t                           # ego fixed
        np.testing.assert_allclose(x[:n, :2], ref[:n, :2], rtol=0, atol=1e-6, err_msg=t)
        assert np.abs(oracle._normalize_theta(x[:n, 2] - ref[:n, 2])).max() < 1e-7, t
        chi_ref = oracle.pose_graph_chi2(ref, graph.kinds, edges)
        assert abs(oracle.pose_graph_chi2(x, graph.kinds, edges) - chi_ref) <= 1e-9 * max(1.0, chi_ref), t
        assert abs(st[2] - chi_ref) <= 1e-9 * max(1.0, chi_ref) and 0 < st[0] < 200 and st[2] <= st[1], (t, st)
        untouched = np.ones(len(x), dtype=bool); untouched[graph.edge_agent] = False; untouched[graph.edge_landmark] = False
        assert np.array_equal(x[untouched], graph.vertices[untouched]), t           # vertices without edges do not move


def test_box_alignment_end_to_end_vs_reference(golden):
    """box_alignment_relative_sample_np (host graph construction + device solve) returns the reference's refined poses; the
    batched entry point and the g2o-shaped PoseGraphOptimization2D wrapper agree with it."""
    from coalign_amd import box_align
    from tests.test_oracle_golden import BOX_ALIGN_CASES, box_align_inputs
    g = golden("box_align.npz")
    samples = []
    for tag in BOX_ALIGN_CASES:
        corners, noisy, unc, flags = box_align_inputs(g, tag)
        refined = box_align.box_alignment_relative_sample_np(corners, noisy.copy(), uncertainty_list=unc, **flags)
        np.testing.assert_allclose(refined[:, :2], g[f"{tag}_refined"][:, :2], rtol=0, atol=2e-5, err_msg=tag)
        assert np.abs((refined[:, 2] - g[f"{tag}_refined"][:, 2] + 180) % 360 - 180).max() < 1e-4, tag
        if tag in ("default", "five_agents", "hard_boxes"):
            samples.append({"pred_corners_list": corners, "noisy_lidar_pose": noisy, "uncertainty_list": unc})
    batch = box_align.box_alignment_batch(samples, use_uncertainty=True, landmark_SE2=True, abandon_hard_cases=True, drop_hard_boxes=True)
    for tag, out in zip(("default", "five_agents", "hard_boxes"), batch):
        np.testing.assert_allclose(out, g[f"{tag}_refined"], rtol=0, atol=1e-4, err_msg=tag)
    # the reference's own call sequence against the g2o-shaped wrapper
    t = "points"
    pgo = box_align.PoseGraphOptimization2D()
    vert, kinds = g[f"{t}_vertices"], g[f"{t}_kinds"]
    for i in range(len(vert)):
        if kinds[i] == 2:
            pgo.add_vertex(i, vert[i, :2], fixed=False, SE2=False)
        else:
            pgo.add_vertex(i, box_align.SE2(vert[i]), fixed=kinds[i] == 0)
    for a, l, m, w in zip(g[f"{t}_edge_agent"], g[f"{t}_edge_landmark"], g[f"{t}_edge_meas"], g[f"{t}_edge_info"]):
        pgo.add_edge([int(a), int(l)], m[:2], np.diag(w[:2]), SE2=False)
    pgo.optimize(1000)
    n = int(g[f"{t}_len"].shape[0])
    got = np.array([pgo.get_pose(i).vector() for i in range(n)])
    np.testing.assert_allclose(got[:, :2], g[f"{t}_solution"][:n, :2], rtol=0, atol=1e-6)


def test_stage1_model_and_post_process_vs_reference(golden, conv_mode):
    """Stage 1 of box alignment on the device: PointPillarUncertainty (pillar kernel + MIOpen + merged 1x1 heads incl. unc_head)
    and UncertaintyVoxelPostprocessor.post_process_stage1 (decode with identity transform, per-agent NMS without the sanity
    mask, uncertainty gather) -- same kept anchors in the same order as the reference, boxes within float32 rounding."""
    g = golden("stage1_mini.npz")
    h = builtin_config("mini_pointpillar_uncertainty")
    model = build_model(h)
    fill_parameters_(model, seed=0, cls_bias=-1.0)
    with torch.no_grad():
        model.reg_head.weight.mul_(0.01); model.reg_head.bias.zero_(); model.cls_head.weight.mul_(0.05)
    model = model.to(DEV).eval()
    batch = {"processed_lidar": {"voxel_features": T(g["voxel_features"]).to(DEV), "voxel_coords": T(g["voxel_coords"]).to(DEV),
                                 "voxel_num_points": T(g["voxel_num_points"]).to(DEV)}}
    with torch.no_grad():
        out = model(batch)
    for k in ("cls_preds", "reg_preds", "unc_preds", "dir_preds"):
        feat_close(out[k], g[k], what=k)
    post = build_postprocessor(h["postprocess"], train=False)
    anchors = T(post.generate_anchor_box())
    # (a) the reference's own head outputs through the device post-process: selection is bit-exact
    ref_out = {k: T(g[k]).to(DEV) for k in ("cls_preds", "reg_preds", "unc_preds", "dir_preds")}
    corners, boxes, unc = post.post_process_stage1(ref_out, anchors)
    assert [len(c) for c in corners] == list(g["n_boxes"])
    for i in range(3):
        assert np.array_equal(unc[i].cpu().numpy(), g[f"unc{i}"])
        np.testing.assert_allclose(corners[i].cpu().numpy(), g[f"corners{i}"], rtol=1e-5, atol=1e-4)
        np.testing.assert_allclose(boxes[i].cpu().numpy(), g[f"boxes{i}"], rtol=1e-5, atol=1e-4)
    # (b) nothing above the threshold -> (None, None, None)
    quiet = dict(ref_out, cls_preds=torch.full_like(ref_out["cls_preds"], -9.0))
    assert post.post_process_stage1(quiet, anchors) == (None, None, None)
    # (c) stage 1 -> box alignment: the device detections of the three agents drive a pose-graph solve end to end
    from coalign_amd import box_align
    c2, _, u2 = post.post_process_stage1(out, anchors)
    poses = np.zeros((3, 6))
    refined = box_align.box_alignment_relative_sample_np([c.cpu().numpy().astype(np.float64) for c in c2], poses,
                                                         uncertainty_list=[u.cpu().numpy().astype(np.float64) for u in u2])
    assert refined.shape == (3, 3) and np.all(np.isfinite(refined)) and np.array_equal(refined[0], [0, 0, 0])


@pytest.mark.parametrize("emu_terms,tol", [(0, 1e-4), (3, 1e-4), (2, 1e-3)])
def test_full_frame_vs_stock_pytorch_ops_on_the_gpu(emu_terms, tol):
    """The whole path at the OPV2V size against the same path written with stock eager PyTorch ops on the GPU (nn.Linear /
    BatchNorm, index scatter, Conv2d+BN+ReLU modules, F.affine_grid + F.grid_sample, bmm + softmax attention, tensor-op box
    decode; tools/eager_torch_baseline.py): head outputs within 1e-4 relative, identical detections.  emu_terms 3 / 2 = the
    opt-in split-bf16 3x3 convolutions (COALIGN_CONV_EMU): 3-way split holds the same 1e-4, 2-way split the north star's 1e-3."""
    import importlib.util
    import os
    from coalign_amd import backbone as bb_mod
    from coalign_amd.detector import to_device
    spec = importlib.util.spec_from_file_location("eager_torch_baseline", os.path.join(os.path.dirname(__file__), "..", "tools", "eager_torch_baseline.py"))
    eb = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(eb)
    h = builtin_config("opv2v_coalign")
    model = build_model(h)
    fill_parameters_(model, seed=0, cls_bias=-2.0)
    with torch.no_grad():
        model.reg_head.weight.mul_(0.02); model.cls_head.weight.mul_(0.3)
    model = model.to(DEV).eval()
    post = build_postprocessor(h["postprocess"], False)
    anchors = T(post.generate_anchor_box()).to(DEV)
    frame = to_device(make_frame(h, 3, pillars_per_agent=6000, seed=77), DEV)
    with torch.no_grad():
        try:
            bb_mod.FAST_INFERENCE = False
            ref = eb.eager_forward(model, frame)
            ref_boxes, ref_scores = eb.eager_post_process(ref, anchors, h["postprocess"])
        finally:
            bb_mod.FAST_INFERENCE = True
        saved = bb_mod.CONV_EMU_TERMS
        try:
            bb_mod.CONV_EMU_TERMS = emu_terms
            out = model(frame)
        finally:
            bb_mod.CONV_EMU_TERMS = saved
        boxes, scores = post.post_process({"ego": {"transformation_matrix": torch.eye(4, device=DEV), "anchor_box": anchors}}, {"ego": out})
    for k in ref:
        err = float((out[k] - ref[k]).abs().max()) / float(ref[k].abs().max())
        print(f"emu_terms {emu_terms} {k}: max |diff| / max |ref| = {err:.2e}")
        assert err <= tol, k
    assert ref_boxes is not None and boxes is not None and boxes.shape == ref_boxes.shape and boxes.shape[0] > 5
    assert float((boxes - ref_boxes).abs().max()) < 10 * tol and float((scores - ref_scores).abs().max()) < tol


@pytest.mark.parametrize("shape", [(2, 64, 64, 100, 352), (3, 128, 128, 50, 176), (2, 256, 256, 25, 88), (1, 384, 256, 36, 96), (2, 64, 128, 37, 52), (1, 8, 64, 5, 4), (1, 256, 64, 100, 352)])
def test_conv3x3_bias_act_vs_torch(shape):
    """The fp32 matrix-core 3x3 convolution (entry point (9); the detector routes the 64-channel stage through it) against torch's
    convolution: with / without residual and ReLU, map sizes that do not divide the tile, one-chunk inputs, long-K shapes whose tiles
    are split between workgroups (stream-K hand-over)."""
    import torch.nn.functional as F
    N, Ci, Co, H, W = shape
    gen = torch.Generator(device="cpu").manual_seed(sum(shape))
    x = torch.randn(N, Ci, H, W, generator=gen).to(DEV)
    w = (torch.randn(Co, Ci, 3, 3, generator=gen) / (Ci * 9) ** 0.5).to(DEV)
    b = torch.randn(Co, generator=gen).to(DEV)
    r = torch.randn(N, Co, H, W, generator=gen).to(DEV)
    wp = ops.pack_conv3x3_weight(w)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    for res, relu in ((None, False), (r, True), (None, True)):
        want = ref if res is None else ref + res.double()
        want = torch.relu(want) if relu else want
        got = ops.conv3x3_bias_act(x, wp, b, res, relu)
        assert got.shape == want.shape
        assert float((got.double() - want).abs().max()) <= 2e-5 * float(want.abs().max()), (shape, res is not None, relu)
    assert torch.equal(ops.conv3x3_bias_act(x, wp, b, r, True), ops.conv3x3_bias_act(x, wp, b, r, True))     # deterministic
    got = ops.conv3x3_bias_act(x, wp, None, None, False)
    assert float((got.double() - (ref - b.double().view(1, -1, 1, 1))).abs().max()) <= 2e-5 * float(ref.abs().max())


@pytest.mark.parametrize("terms,tol", [(3, 5e-6), (2, 2e-5)])
@pytest.mark.parametrize("shape", [(2, 64, 64, 100, 352), (3, 128, 128, 50, 176), (2, 256, 256, 25, 88), (1, 384, 256, 36, 96), (2, 64, 128, 37, 52), (1, 8, 64, 5, 4),
                                   (1, 16, 64, 9, 63), (1, 256, 256, 100, 352)])
def test_conv3x3_emu_bias_act_vs_fp64(shape, terms, tol):
    """The opt-in split-bf16 3x3 convolution (entry point (9b)) against the fp64 convolution: the 3-way split must be as accurate
    as a native fp32 convolution (5e-6 of the output scale; torch's own fp32 kernels land at 2e-7 .. 4e-6 on these shapes), the
    2-way split within 2e-5; with / without residual and ReLU, ragged map sizes (odd widths: no alignment requirement), one-chunk
    inputs, the shrink-header shape whose 572 long tiles are split between workgroups (stream-K hand-over), inputs spanning 12
    orders of magnitude (the split is exponent-agnostic), determinism."""
    import torch.nn.functional as F
    N, Ci, Co, H, W = shape
    gen = torch.Generator(device="cpu").manual_seed(sum(shape) + terms)
    x = torch.randn(N, Ci, H, W, generator=gen).to(DEV)
    w = (torch.randn(Co, Ci, 3, 3, generator=gen) / (Ci * 9) ** 0.5).to(DEV)
    b = torch.randn(Co, generator=gen).to(DEV)
    r = torch.randn(N, Co, H, W, generator=gen).to(DEV)
    ws = ops.pack_conv3x3_emu_weight(w, terms)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    for res, relu in ((None, False), (r, True)):
        want = ref if res is None else ref + res.double()
        want = torch.relu(want) if relu else want
        got = ops.conv3x3_emu_bias_act(x, ws, b, Co, res, relu, terms)
        assert got.shape == want.shape
        assert float((got.double() - want).abs().max()) <= tol * float(want.abs().max()), (shape, res is not None, relu)
    assert torch.equal(ops.conv3x3_emu_bias_act(x, ws, b, Co, r, True, terms), ops.conv3x3_emu_bias_act(x, ws, b, Co, r, True, terms))
    # scale invariance: per-channel input scales from 1e-6 to 1e6 (no fixed-point assumptions in the split)
    scale = (10.0 ** torch.linspace(-6, 6, Ci)).view(1, -1, 1, 1).to(DEV)
    xs, wsc = x * scale, w / scale
    got = ops.conv3x3_emu_bias_act(xs, ops.pack_conv3x3_emu_weight(wsc, terms), b, Co, None, False, terms)
    want = F.conv2d(xs.double(), wsc.double(), b.double(), padding=1)
    assert float((got.double() - want).abs().max()) <= tol * float(want.abs().max())


@pytest.mark.parametrize("terms,tol", [(3, 5e-6), (2, 2e-5)])
@pytest.mark.parametrize("shape", [(2, 64, 64, 100, 352), (3, 128, 128, 50, 176), (2, 256, 256, 25, 88), (1, 384, 256, 36, 96), (2, 64, 128, 37, 52),
                                   (1, 16, 64, 9, 63), (1, 384, 256, 100, 352)])
def test_conv3x3_emu_tap_major_vs_fp64(shape, terms, tol):
    """The tap-major weight image of the split-bf16 convolution (COALIGN_LAYOUT_W_TAPMAJOR: 16-channel intervals of nine matrix steps,
    no zero tenth tap): same bounds against the fp64 convolution as the default image, NCHW and channels-last output, residual / ReLU,
    ragged maps, the shrink-header shape (stream-K hand-over), determinism, and agreement with the default image to rounding."""
    import torch.nn.functional as F
    N, Ci, Co, H, W = shape
    gen = torch.Generator(device="cpu").manual_seed(sum(shape) + terms + 7)
    x = torch.randn(N, Ci, H, W, generator=gen).to(DEV)
    w = (torch.randn(Co, Ci, 3, 3, generator=gen) / (Ci * 9) ** 0.5).to(DEV)
    b = torch.randn(Co, generator=gen).to(DEV)
    r = torch.randn(N, Co, H, W, generator=gen).to(DEV)
    wt = ops.pack_conv3x3_emu_weight(w, terms, tap_major=True)
    assert wt.numel() != ops.pack_conv3x3_emu_weight(w, terms).numel()
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    for res, relu in ((None, False), (r, True)):
        want = ref if res is None else ref + res.double()
        want = torch.relu(want) if relu else want
        for cl in (False, True):
            got = ops.conv3x3_emu_bias_act(x, wt, b, Co, res, relu, terms, out_channels_last=cl)
            assert got.shape == want.shape and ops.is_channels_last(got) == (cl and Co > 1)
            assert float((got.double() - want).abs().max()) <= tol * float(want.abs().max()), (shape, res is not None, relu, cl)
    assert torch.equal(ops.conv3x3_emu_bias_act(x, wt, b, Co, r, True, terms), ops.conv3x3_emu_bias_act(x, wt, b, Co, r, True, terms))
    pairs = ops.conv3x3_emu_bias_act(x, ops.pack_conv3x3_emu_weight(w, terms), b, Co, r, True, terms)
    assert float((pairs - ops.conv3x3_emu_bias_act(x, wt, b, Co, r, True, terms)).abs().max()) <= 2 * tol * float(pairs.abs().max())
    with pytest.raises(ValueError):
        ops.conv3x3_emu_bias_act(x, wt, b, Co, None, True, terms, stride=2)


def test_batch_dict_producer_on_device_vs_reference_dataset(golden, conv_mode):
    """next-4 end to end: raw per-cav records -> IntermediateFusionBatcher with the device voxeliser -> the batch the reference's
    dataset + collate produce (bit-identical pillars, poses, transforms, ground truth), and the detector + post-process +
    evaluation run straight off that batch."""
    from coalign_amd import evaluation as ev
    from coalign_amd.dataset import IntermediateFusionBatcher
    from coalign_amd.detector import to_device
    from tests.test_host_cpu import check_batch_against_reference, dataset_scenario
    g = golden("dataset.npz")
    h = builtin_config("opv2v_coalign")
    h.pop("box_align", None)
    batcher = IntermediateFusionBatcher(h, train=False, device=DEV)
    batches = {}
    for tag in ("a", "b"):
        np.random.seed(int(g[f"{tag}_np_seed"]))
        batches[tag] = batcher(dataset_scenario(g, tag))
        check_batch_against_reference(batches[tag], g, tag)
    model = build_model(h)
    fill_parameters_(model, seed=0, cls_bias=-2.0)
    with torch.no_grad():
        model.reg_head.weight.mul_(0.02); model.cls_head.weight.mul_(0.3)
    model = model.to(DEV).eval()
    batch = to_device(batches["a"], DEV)
    with torch.no_grad():
        out = model(batch["ego"])
    boxes, scores, gt = batcher.post_process(batch, {"ego": out})
    assert gt.shape == (int(g["a_object_bbx_mask"].sum()), 8, 3) or gt.shape[0] <= int(g["a_object_bbx_mask"].sum())
    stat = ev.new_result_stat()
    for thr in ev.IOU_THRESHOLDS:
        ev.caluclate_tp_fp(boxes, scores, gt.to(DEV), stat, thr)
    assert stat[0.7]["gt"] == gt.shape[0]


@pytest.mark.parametrize("case", [(1, 64, 128, 100, 352, 1, 1), (2, 128, 128, 50, 176, 2, 1), (1, 256, 128, 25, 88, 4, 1), (3, 64, 128, 101, 353, 1, 2),
                                  (2, 128, 256, 50, 176, 1, 2), (1, 6, 20, 7, 9, 1, 1), (1, 10, 8, 5, 3, 2, 1)])
def test_pointwise_conv_vs_torch(case):
    """coalign_pointwise_conv (non-overlapping transposed convolutions written into a channel slice of a larger tensor, 1x1 stride-2
    convolutions) against torch, incl. odd map sizes, channel counts that do not fill a 32-row tile, and an offset slice."""
    import torch.nn.functional as F
    N, Ci, Co, H, W, up, st = case
    gen = torch.Generator(device="cpu").manual_seed(sum(case))
    x = torch.randn(N, Ci, H, W, generator=gen).to(DEV)
    b = torch.randn(Co, generator=gen).to(DEV)
    if st == 1:
        w = (torch.randn(Ci, Co, up, up, generator=gen) / Ci ** 0.5).to(DEV)
        ref = F.conv_transpose2d(x.double(), w.double(), b.double(), stride=up)
        wp = ops.pack_pointwise_weight(w, True) if (Co * up * up) % 32 == 0 else None
        if wp is None:          # rows that do not fill a tile: the 1x1 form with zero-padded columns
            assert up == 1 or Co * up * up % 32, case
            if up != 1:
                with pytest.raises(Exception):
                    ops.pointwise_conv(x, w.reshape(Ci, -1).contiguous(), b, Co, up=up)
                return
            wp = ops.pack_pointwise_weight(w.reshape(Ci, Co).t().reshape(Co, Ci, 1, 1).contiguous(), False)
    else:
        w = (torch.randn(Co, Ci, 1, 1, generator=gen) / Ci ** 0.5).to(DEV)
        ref = F.conv2d(x.double(), w.double(), b.double(), stride=st)
        wp = ops.pack_pointwise_weight(w, False)
    for relu in (True, False):
        want = torch.relu(ref) if relu else ref
        got = ops.pointwise_conv(x, wp, b, Co, up=up, in_stride=st, relu=relu)
        assert got.shape == want.shape
        assert float((got.double() - want).abs().max()) <= 2e-5 * max(1.0, float(want.abs().max())), (case, relu)
    big = torch.full((N, Co + 40, ref.shape[2], ref.shape[3]), 7.0, device=DEV)
    ops.pointwise_conv(x, wp, b, Co, up=up, in_stride=st, relu=True, out=big, c_off=24)
    assert float((big[:, 24:24 + Co].double() - torch.relu(ref)).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max()))
    assert bool((big[:, :24] == 7.0).all()) and bool((big[:, 24 + Co:] == 7.0).all())        # neighbours of the slice untouched


def test_pcdet_iou3d_and_nms_vs_oracle():
    """Row N: the OpenPCDet-semantics API (boxes_iou3d_gpu, nms_gpu) against the oracle's restatement of the CUDA extension."""
    from coalign_amd import pcdet
    rs = np.random.RandomState(17)
    n = 300
    boxes = np.zeros((n, 7), dtype=np.float32)
    boxes[:, 0] = rs.uniform(-20, 20, n); boxes[:, 1] = rs.uniform(-10, 10, n); boxes[:, 2] = rs.uniform(-1.2, -0.8, n)
    boxes[:, 3] = rs.uniform(3.5, 5, n); boxes[:, 4] = rs.uniform(1.6, 2.1, n); boxes[:, 5] = rs.uniform(1.4, 1.8, n); boxes[:, 6] = rs.uniform(-3.1, 3.1, n)
    scores = rs.uniform(0, 1, n).astype(np.float32)
    bd, sd = T(boxes).to(DEV), T(scores).to(DEV)
    for thr, pre in ((0.01, None), (0.3, 200)):
        keep, _ = pcdet.nms_gpu(bd, sd, thr, pre_maxsize=pre)
        assert keep.cpu().numpy().tolist() == oracle.pcdet_nms(boxes, scores, thr, pre).tolist()
    a, b = boxes[:40], boxes[40:90]
    got = pcdet.boxes_iou3d_gpu(T(a).to(DEV), T(b).to(DEV)).cpu().numpy()
    ov = np.array([[oracle.pcdet_overlap(x, y) for y in b] for x in a], dtype=np.float32)        # fp32 BEV overlap areas
    np.testing.assert_allclose(ops.boxes_overlap_bev(T(a).to(DEV), T(b).to(DEV)).cpu().numpy(), ov, rtol=1e-5, atol=1e-6)
    top = np.minimum((a[:, 2] + a[:, 5] / 2)[:, None], (b[:, 2] + b[:, 5] / 2)[None])
    bottom = np.maximum((a[:, 2] - a[:, 5] / 2)[:, None], (b[:, 2] - b[:, 5] / 2)[None])
    h = np.clip(top - bottom, 0, None)
    o3 = ov * h
    want = o3 / np.clip((a[:, 3] * a[:, 4] * a[:, 5])[:, None] + (b[:, 3] * b[:, 4] * b[:, 5])[None] - o3, 1e-6, None)
    np.testing.assert_allclose(got, want, rtol=2e-4, atol=1e-5)
    assert (got > 0.05).sum() > 5
