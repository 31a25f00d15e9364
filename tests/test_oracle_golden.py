"""Pin the CPU oracle against golden vectors produced by the reference itself
(tests/golden/make_golden.py).  CPU only; runs in the build container and on the GPU box."""
import hashlib

import numpy as np
import pytest
import torch

from oracle import coalign_oracle as oracle
from coalign_amd.config import builtin_config

T = torch.from_numpy


def close(a, b, rtol=1e-5, atol=1e-6):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    np.testing.assert_allclose(a, np.asarray(b), rtol=rtol, atol=atol)


def test_pillar_vfe_and_scatter(golden):
    g = golden("model_mini.npz")
    sd = {"pillar_vfe.pfn_layers.0.linear.weight": T(g["pfn_weight"]), "pillar_vfe.pfn_layers.0.norm.weight": T(g["pfn_bn_weight"]),
          "pillar_vfe.pfn_layers.0.norm.bias": T(g["pfn_bn_bias"]), "pillar_vfe.pfn_layers.0.norm.running_mean": T(g["pfn_bn_mean"]),
          "pillar_vfe.pfn_layers.0.norm.running_var": T(g["pfn_bn_var"])}
    h = builtin_config("mini_coalign")["model"]["args"]
    pf = oracle.pillar_vfe(T(g["voxel_features"]), T(g["voxel_num_points"]), T(g["voxel_coords"]), sd, h["voxel_size"], h["lidar_range"])
    close(pf, g["pillar_features"], rtol=1e-5, atol=1e-5)
    canvas = oracle.scatter(pf, T(g["voxel_coords"]), int(g["record_len"].sum()), 64, 32)
    close(canvas.reshape(-1)[::7], g["canvas_sample"], rtol=1e-5, atol=1e-5)
    # integer indexing is bit exact: same set of occupied positions
    nz = torch.nonzero(oracle.scatter(T(g["pillar_features"]), T(g["voxel_coords"]), 5, 64, 32).reshape(-1)).view(-1).int().numpy()
    assert np.array_equal(nz, g["canvas_nonzero"])


def test_pose_algebra(golden):
    g = golden("pose.npz")
    for p, w in zip(g["poses"], g["x_to_world"]):
        close(oracle.x_to_world(p), w, rtol=0, atol=1e-15)
    pair = oracle.pairwise_transformation(g["poses"], 5)
    close(pair, g["pairwise"], rtol=0, atol=1e-13)
    pt = T(g["pairwise"])[None]
    close(oracle.normalize_pairwise_tfm(pt, 200, 704, 0.4), g["normalized_200x704"], rtol=0, atol=1e-15)
    close(oracle.normalize_pairwise_tfm(pt, 32, 64, 0.4), g["normalized_32x64"], rtol=0, atol=1e-15)
    close(pt, g["pairwise_after"][None] if g["pairwise_after"].ndim == 4 else g["pairwise_after"], rtol=0, atol=0)


def test_warp_affine_simple(golden):
    g = golden("warp.npz")
    out = oracle.warp_affine_simple(T(g["src"]), T(g["theta"]), (16, 32))
    close(out, g["warped"], rtol=1e-5, atol=2e-6)
    assert np.all(g["warped"][5] == 0) and torch.all(out[5] == 0)          # out-of-range case is exactly zero


def test_att_and_max_fusion(golden):
    g = golden("fusion.npz")
    rl, aff = T(g["record_len"]), T(g["affine"])
    for s in range(3):
        x = T(g[f"x{s}"])
        close(oracle.att_fuse(x, rl, aff), g[f"att{s}"], rtol=1e-5, atol=2e-6)
        close(oracle.max_fuse(x, rl, aff), g[f"max{s}"], rtol=1e-5, atol=2e-6)


def test_anchors(golden):
    g = golden("anchors.npz")
    for tag, cfg in (("opv2v_coalign", "opv2v_coalign"), ("opv2v_late", "opv2v_pointpillar_late"),
                     ("dairv2x_coalign", "dairv2x_coalign"), ("mini", "mini_coalign")):
        h = builtin_config(cfg)
        a = oracle.generate_anchor_box(h["postprocess"]["anchor_args"], h["postprocess"]["order"])
        assert list(a.shape) == list(g[tag + "_shape"]) and a.dtype == np.float64
        assert hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest() == g[tag + "_sha256"].tobytes()
        assert np.array_equal(a.reshape(-1, 7)[0], g[tag + "_first"]) and np.array_equal(a.reshape(-1, 7)[-1], g[tag + "_last"])
        assert np.array_equal(np.asarray(h["model"]["args"]["point_pillar_scatter"]["grid_size"]), g[tag + "_grid_size"])
        assert [h["postprocess"]["anchor_args"][k] for k in "WHD"] == list(g[tag + "_WHD"])
    assert np.array_equal(oracle.generate_anchor_box(builtin_config("mini_coalign")["postprocess"]["anchor_args"]), g["mini_full"])


def test_geometry_helpers(golden):
    g = golden("postprocess.npz")
    c8 = oracle.boxes_to_corners_3d(T(g["g_boxes7"]), "hwl")
    close(c8, g["g_corners"], rtol=1e-6, atol=1e-6)
    close(oracle.project_box3d(c8, T(g["l_T1"])), g["g_proj"], rtol=1e-6, atol=1e-6)
    assert np.array_equal(oracle.remove_large_pred_bbx(T(g["g_corners"])).numpy(), g["g_keep_large"])
    assert np.array_equal(oracle.remove_bbx_abnormal_z(T(g["g_corners"])).numpy(), g["g_keep_z"])
    close(oracle.delta_to_boxes3d(T(g["i_reg"]), T(g["anchors"])), g["i_delta_boxes"], rtol=1e-6, atol=1e-6)


def _pp_cfg():
    return builtin_config("mini_coalign")["postprocess"]


def test_post_process_intermediate(golden):
    g = golden("postprocess.npz")
    boxes, scores, info = oracle.post_process([dict(cls_preds=T(g["i_cls"]), reg_preds=T(g["i_reg"]), dir_preds=T(g["i_dir"]))],
                                              T(g["anchors"]), _pp_cfg())
    assert boxes.shape == g["i_boxes"].shape                               # identical selection
    close(scores, g["i_scores"], rtol=0, atol=0)
    close(boxes, g["i_boxes"], rtol=1e-6, atol=1e-5)


def test_post_process_late_and_empty(golden):
    g = golden("postprocess.npz")
    agents = [dict(cls_preds=T(g["l_cls0"]), reg_preds=T(g["l_reg0"]), dir_preds=T(g["l_dir0"]), transformation_matrix=torch.eye(4)),
              dict(cls_preds=T(g["l_cls1"]), reg_preds=T(g["l_reg1"]), dir_preds=T(g["l_dir1"]), transformation_matrix=T(g["l_T1"]))]
    boxes, scores, _ = oracle.post_process(agents, T(g["anchors"]), _pp_cfg())
    assert boxes.shape == g["l_boxes"].shape
    close(scores, g["l_scores"], rtol=0, atol=0)
    close(boxes, g["l_boxes"], rtol=1e-6, atol=1e-5)
    b, s, _ = oracle.post_process([dict(cls_preds=torch.full((1, 2, 16, 32), -9.0), reg_preds=T(g["i_reg"]), dir_preds=T(g["i_dir"]))],
                                  T(g["anchors"]), _pp_cfg())
    assert b is None and s is None and bool(g["none_result"].all())


@pytest.mark.parametrize("tag", ["small", "mid", "over1000"])
def test_nms_control_flow(golden, tag):
    g = golden("nms.npz")
    keep = oracle.nms_rotated(g[f"{tag}_corners"], g[f"{tag}_scores"], 0.15)
    assert keep.dtype == np.int32 and np.array_equal(keep, g[f"{tag}_keep"])
    if tag == "small":
        assert np.array_equal(oracle.nms_rotated_numpy(g["small_corners"], g["small_scores"], 0.15), g["small_keep"])


def test_nms_edge_cases(golden):
    g = golden("nms.npz")
    assert oracle.nms_rotated(np.zeros((0, 8, 3), np.float32), np.zeros(0, np.float32), 0.15).shape == (0,) == g["empty_keep"].shape
    q = g["over1000_corners"][:50, :4, :2]
    assert np.array_equal(oracle.nms_rotated(q, g["over1000_scores"][:50], 0.15), g["quad_keep"])


def test_iou_known_answers():
    def rect(cx, cy, l, w, th):
        c, s = np.cos(th), np.sin(th)
        t = np.array([[1, -1], [1, 1], [-1, 1], [-1, -1]]) * np.array([l, w]) / 2
        return t @ np.array([[c, s], [-s, c]]) + np.array([cx, cy])
    a = rect(0, 0, 3.9, 1.6, 0)
    assert oracle.quad_iou(a, a) == 1.0
    assert oracle.quad_iou(a, rect(10, 0, 3.9, 1.6, 0)) == 0.0
    assert abs(oracle.quad_iou(a, rect(0, 0, 3.9, 1.6, np.pi / 2)) - 1.6 ** 2 / (2 * 3.9 * 1.6 - 1.6 ** 2)) < 1e-14
    assert abs(oracle.quad_iou(a, rect(3.9 / 2, 0, 3.9, 1.6, 0)) - 1 / 3) < 1e-14
    assert abs(oracle.quad_iou(a, a[::-1].copy()) - 1.0) < 1e-15                  # orientation independent
    rng = np.random.default_rng(0)
    for _ in range(500):                                                          # two independent algorithms agree
        p = rect(*rng.uniform(-2, 2, 2), *rng.uniform(1, 5, 2), rng.uniform(-4, 4))
        q = rect(*rng.uniform(-2, 2, 2), *rng.uniform(1, 5, 2), rng.uniform(-4, 4))
        assert abs(oracle.quad_iou(p, q) - oracle.quad_iou_python(p, q)) < 1e-12
    assert np.isnan(oracle.quad_iou(np.zeros((4, 2)), np.zeros((4, 2))))           # zero-area union -> NaN, never suppresses


def _filled_state_dict(cfg_name):
    """state_dict with the deterministic name-keyed test weights (built from the product's module classes only to get
    names and shapes; values come from fill_parameters_)."""
    from coalign_amd.synthetic import fill_parameters_
    from coalign_amd.detector import build_model
    h = builtin_config(cfg_name)
    m = build_model(h)
    fill_parameters_(m, seed=0, cls_bias=-1.0)
    return h, {k: v.clone() for k, v in m.state_dict().items()}


def test_full_model_mini(golden):
    g = golden("model_mini.npz")
    h, sd = _filled_state_dict("mini_coalign")
    assert len(sd) == int(g["n_state"])
    batch = {"processed_lidar": {"voxel_features": T(g["voxel_features"]), "voxel_coords": T(g["voxel_coords"]),
                                 "voxel_num_points": T(g["voxel_num_points"])},
             "record_len": T(g["record_len"]), "pairwise_t_matrix": T(g["pairwise_t_matrix"])}
    out = oracle.coalign_forward(sd, h["model"]["args"], batch, return_intermediate=True)
    close(out["pillar_features"], g["pillar_features"], rtol=1e-5, atol=1e-5)
    close(out["affine"], g["normalized_affine"], rtol=0, atol=1e-15)
    for s in range(3):
        scale = float(np.abs(g[f"fused{s}"]).max())
        close(out["feats"][s].reshape(-1)[::5], g[f"feat{s}_sample"], rtol=1e-4, atol=1e-5 * scale)
        close(out["fused"][s], g[f"fused{s}"], rtol=1e-4, atol=1e-5 * scale)
    for k in ("cls_preds", "reg_preds", "dir_preds"):
        close(out[k], g[k], rtol=1e-4, atol=1e-4 * float(np.abs(g[k]).max()))


def test_evaluation_matches_reference(golden):
    """TP/FP matching + VOC AP (next-2): the oracle restatement reproduces the reference's eval_utils on 3 frames
    (IoU through the stand-in polygon, see make_golden.py) at IoU 0.3 / 0.5 / 0.7."""
    g = golden("eval.npz")
    stat = {t: {"tp": [], "fp": [], "gt": 0, "score": []} for t in (0.3, 0.5, 0.7)}
    for f in range(3):
        for thr in (0.3, 0.5, 0.7):
            oracle.caluclate_tp_fp(g[f"det{f}"], g[f"score{f}"], g[f"gt{f}"], stat, thr)
    oracle.caluclate_tp_fp(None, None, g["gt2"], stat, 0.7)
    for thr in (0.3, 0.5, 0.7):
        tag = str(int(thr * 100))
        assert stat[thr]["tp"] == list(g[f"tp{tag}"]) and stat[thr]["fp"] == list(g[f"fp{tag}"]) and stat[thr]["gt"] == int(g[f"gtn{tag}"])
        ap, mrec, mpre = oracle.calculate_ap(stat, thr)
        assert abs(ap - float(g[f"ap{tag}"])) < 1e-12
        np.testing.assert_allclose(mrec, g[f"mrec{tag}"], rtol=0, atol=1e-12)
        np.testing.assert_allclose(mpre, g[f"mpre{tag}"], rtol=0, atol=1e-12)


def test_evaluation_host_logic_matches_reference(golden):
    """The product's host-side matching / AP code (coalign_amd.evaluation) fed with the oracle's IoU matrix."""
    from coalign_amd import evaluation as ev
    g = golden("eval.npz")
    stat = ev.new_result_stat()
    for f in range(3):
        iou = oracle.iou_matrix(g[f"det{f}"], g[f"gt{f}"])
        for thr in (0.3, 0.5, 0.7):
            tp, fp, sc = ev.match_tp_fp(iou, g[f"score{f}"], thr)
            stat[thr]["tp"] += tp; stat[thr]["fp"] += fp; stat[thr]["score"] += sc.tolist(); stat[thr]["gt"] += len(g[f"gt{f}"])
    stat[0.7]["gt"] += len(g["gt2"])
    for thr in (0.3, 0.5, 0.7):
        tag = str(int(thr * 100))
        assert stat[thr]["tp"] == list(g[f"tp{tag}"]) and stat[thr]["fp"] == list(g[f"fp{tag}"])
        ap, mrec, mpre = ev.calculate_ap(stat, thr)
        assert abs(ap - float(g[f"ap{tag}"])) < 1e-12
        np.testing.assert_allclose(mrec, g[f"mrec{tag}"], rtol=0, atol=1e-12)
        np.testing.assert_allclose(mpre, g[f"mpre{tag}"], rtol=0, atol=1e-12)


def _gt_inputs(g):
    return {c: {"transformation_matrix_clean": T(g[f"gtgen_T{c}"]), "object_bbx_center": T(g[f"gtgen_centre{c}"]),
                "object_bbx_mask": T(g[f"gtgen_mask{c}"]), "object_ids": g[f"gtgen_ids{c}"].tolist()} for c in (0, 1)}


def test_generate_gt_bbx_matches_reference(golden):
    g = golden("eval.npz")
    rng = [-140.8, -40, -3, 140.8, 40, 1]
    got = oracle.generate_gt_bbx(_gt_inputs(g), "hwl", rng)
    assert np.array_equal(got.numpy(), g["gtgen_out"])
    assert 0 < len(got) < 9          # 9 distinct ids, some outside the range: both the de-dup and the filter fired
    from coalign_amd.config import builtin_config
    from coalign_amd.postprocess import build_postprocessor
    post = build_postprocessor(builtin_config("opv2v_coalign")["postprocess"], train=False)
    mine = post.generate_gt_bbx(_gt_inputs(g))
    assert mine.shape == got.shape
    np.testing.assert_allclose(mine.numpy(), g["gtgen_out"], rtol=0, atol=2e-5)


def test_point_filters_match_reference(golden):
    """mask_points_by_range / mask_ego_points (pcd_utils.py:41-88) incl. points exactly on and one ulp off every boundary."""
    g = golden("points.npz")
    assert np.array_equal(oracle.mask_points_by_range(g["cloud"], [-140.8, -40, -3, 140.8, 40, 1]), g["range_masked"])
    assert np.array_equal(oracle.mask_ego_points(g["cloud"]), g["ego_masked"])
    assert len(g["ego_masked"]) < len(g["cloud"]) and len(g["range_masked"]) < len(g["cloud"])


def test_voxel_generator_restatement_is_self_consistent(golden):
    """spconv is absent (parity unpinned): the C loop is cross-checked against the plain-Python transcription of the same
    published loop and against the invariants of the output contract."""
    cloud = golden("points.npz")["cloud"]
    rs = np.random.RandomState(3)
    dense = np.concatenate([cloud, rs.normal([5, 3, -1, 0.5], [0.6, 0.6, 0.3, 0.1], (1500, 4)).astype(np.float32)])
    for max_points, max_voxels in ((32, 70000), (5, 16000), (3, 40)):
        v, c, n = oracle.points_to_voxel(dense, [0.4, 0.4, 4], [-140.8, -40, -3, 140.8, 40, 1], max_points, max_voxels)
        pv, pc, pn = oracle.points_to_voxel_python(dense, [0.4, 0.4, 4], [-140.8, -40, -3, 140.8, 40, 1], max_points, max_voxels)
        assert np.array_equal(v, pv) and np.array_equal(c, pc) and np.array_equal(n, pn)
        assert len(v) <= max_voxels and n.min() >= 1 and n.max() <= max_points
        assert len(np.unique(c, axis=0)) == len(c)                                   # one voxel per cell
        assert c[:, 0].max() == 0 and c[:, 1].max() < 200 and c[:, 2].max() < 704    # (z, y, x)
        for i in (0, len(v) // 2, len(v) - 1):                                       # points sit in their cell, padding is zero
            cell = np.floor((v[i, : n[i], :3] - np.float32([-140.8, -40, -3])) / np.float32([0.4, 0.4, 4])).astype(int)
            assert np.all(cell[:, ::-1] == c[i]) and not v[i, n[i]:].any()
    assert n.max() == 3 and len(v) == 40
    feats, coords, num = oracle.collate_voxels([(v, c, n), (pv, pc, pn)])
    assert feats.shape[0] == 80 and coords.shape == (80, 4) and list(np.unique(coords[:, 0])) == [0, 1]


BOX_ALIGN_CASES = ("default", "five_agents", "no_uncertainty", "points", "adaptive", "hard_boxes", "abandoned_few", "abandoned_yaw", "empty_agent")


def box_align_inputs(g, tag):
    lens = g[f"{tag}_len"]
    cuts = np.cumsum(lens)[:-1]
    flags = {}
    for item in g[f"{tag}_flags"]:
        k, v = str(item).split("=")
        flags[k] = v == "True"
    return np.split(g[f"{tag}_corners"], cuts), g[f"{tag}_noisy"], np.split(g[f"{tag}_unc"], cuts), flags


@pytest.mark.parametrize("tag", BOX_ALIGN_CASES)
def test_box_alignment_matches_reference(golden, tag):
    """next-3: the pose graph the reference hands to g2o (vertices, kinds, edges, information) and the refined poses it returns
    (solver = the oracle's LM behind a recording g2o stand-in, see make_golden.py) are reproduced by the restatement."""
    g = golden("box_align.npz")
    corners, noisy, unc, flags = box_align_inputs(g, tag)
    graph = oracle.build_pose_graph(corners, noisy, unc, **flags)
    if int(g[f"{tag}_solved"]) == 0:
        assert graph is None
    else:
        assert np.array_equal(graph["kinds"], g[f"{tag}_kinds"])
        assert np.array_equal(graph["edges"][0], g[f"{tag}_edge_agent"]) and np.array_equal(graph["edges"][1], g[f"{tag}_edge_landmark"])
        np.testing.assert_allclose(graph["vertices"], g[f"{tag}_vertices"], rtol=0, atol=1e-9)
        np.testing.assert_allclose(graph["edges"][2], g[f"{tag}_edge_meas"], rtol=0, atol=1e-12)
        np.testing.assert_allclose(graph["edges"][3], g[f"{tag}_edge_info"], rtol=1e-12, atol=0)
    refined = oracle.box_alignment_relative_sample_np(corners, noisy, unc, **flags)
    np.testing.assert_allclose(refined, g[f"{tag}_refined"], rtol=0, atol=1e-7)


def test_pose_graph_lm_reaches_the_least_squares_optimum(golden):
    """g2o is absent (parity unpinned): the restated Levenberg-Marquardt is checked against an independent solver
    (scipy.optimize.least_squares, trust-region) on the same residuals, and against the defining properties of the problem."""
    from scipy.optimize import least_squares
    g = golden("box_align.npz")
    for tag in ("default", "five_agents", "points", "adaptive"):
        kinds = g[f"{tag}_kinds"]
        edges = (g[f"{tag}_edge_agent"], g[f"{tag}_edge_landmark"], g[f"{tag}_edge_meas"], g[f"{tag}_edge_info"])
        x, stats = oracle.pose_graph_lm(g[f"{tag}_vertices"], kinds, edges)
        assert stats["chi2"] < stats["chi2_initial"] and stats["iterations"] < 200
        assert np.array_equal(x[0], g[f"{tag}_vertices"][0])                      # the ego vertex is fixed
        active = np.zeros(len(kinds), dtype=bool); active[edges[0]] = True; active[edges[1]] = True
        free = np.nonzero(active & (kinds != 0))[0]
        assert np.array_equal(x[~active], g[f"{tag}_vertices"][~active])          # vertices without edges do not move
        cols = np.concatenate([[3 * i, 3 * i + 1] + ([3 * i + 2] if kinds[v] == 1 else []) for i, v in enumerate(free)]).astype(int)

        def fun(p):
            v = g[f"{tag}_vertices"].copy()
            flat = v[free].ravel(); flat[cols] = p; v[free] = flat.reshape(-1, 3)
            return (oracle.pose_graph_residuals(v, kinds, edges) * np.sqrt(edges[3])).ravel()
        sol = least_squares(fun, g[f"{tag}_vertices"][free].ravel()[cols], xtol=1e-15, ftol=1e-15, gtol=1e-15)
        assert abs(2 * sol.cost - stats["chi2"]) <= 1e-9 * max(1.0, stats["chi2"])
        n_agents = int(np.sum(edges[0].max() + 1))
        best = g[f"{tag}_vertices"].copy(); flat = best[free].ravel(); flat[cols] = sol.x; best[free] = flat.reshape(-1, 3)
        np.testing.assert_allclose(x[:n_agents, :2], best[:n_agents, :2], rtol=0, atol=1e-6)
        assert np.abs(oracle._normalize_theta(x[:n_agents, 2] - best[:n_agents, 2])).max() < 1e-7


def test_stage1_uncertainty_model_and_post_process_match_reference(golden):
    """next-3 stage 1: PointPillarUncertainty forward (incl. the unc head) and post_process_stage1 (per-agent boxes in the
    agent frame + the kept anchors' log-variances) against the reference on a 3-agent mini frame."""
    from coalign_amd.detector import build_model
    from coalign_amd.synthetic import fill_parameters_
    g = golden("stage1_mini.npz")
    h = builtin_config("mini_pointpillar_uncertainty")
    model = build_model(h)
    sd = model.state_dict()
    assert list(sd.keys()) == [str(k) for k in g["state_keys"]] and [v.numel() for v in sd.values()] == list(g["state_numel"])
    fill_parameters_(model, seed=0, cls_bias=-1.0)
    with torch.no_grad():
        model.reg_head.weight.mul_(0.01); model.reg_head.bias.zero_(); model.cls_head.weight.mul_(0.05)
    sd = model.state_dict()
    batch = {"processed_lidar": {"voxel_features": T(g["voxel_features"]), "voxel_coords": T(g["voxel_coords"]), "voxel_num_points": T(g["voxel_num_points"])}}
    with torch.no_grad():
        out = oracle.pointpillar_forward(sd, h["model"]["args"], batch)
    for k in ("cls_preds", "reg_preds", "unc_preds", "dir_preds"):
        ref = T(g[k])
        assert float((out[k] - ref).abs().max()) <= 1e-4 * max(1.0, float(ref.abs().max())), k
    anchors = T(oracle.generate_anchor_box(h["postprocess"]["anchor_args"], h["postprocess"]["order"]))
    ref_out = {k: T(g[k]) for k in ("cls_preds", "reg_preds", "unc_preds", "dir_preds")}
    corners, boxes, unc = oracle.post_process_stage1(ref_out, anchors, h["postprocess"])
    assert [len(c) for c in corners] == list(g["n_boxes"]) and min(g["n_boxes"]) > 10
    for i in range(3):
        assert np.array_equal(unc[i].numpy(), g[f"unc{i}"])                      # same anchors kept, same order
        np.testing.assert_allclose(corners[i].numpy(), g[f"corners{i}"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(boxes[i].numpy(), g[f"boxes{i}"], rtol=1e-5, atol=1e-5)


def test_naive_compressor_matches_reference(golden):
    """Row D: the oracle's restatement of NaiveCompressor (naive_compress.py:5-31) against the reference module (randomised BN
    statistics, ratios 2 and 8) and against the reference model built from a yaml with ``compression: 4``."""
    import copy
    from coalign_amd.backbone import NaiveCompressor
    from coalign_amd.detector import build_model
    from coalign_amd.synthetic import fill_parameters_
    g = golden("naive_compress.npz")
    x = T(g["x"])
    for ratio in (2, 8):
        m = NaiveCompressor(64, ratio).eval()
        fill_parameters_(m, seed=40 + ratio)
        assert list(m.state_dict().keys()) == list(g[f"keys_r{ratio}"])          # checkpoint names of the reference module
        sd = {"naive_compressor." + k: v for k, v in m.state_dict().items()}
        y = oracle.naive_compressor(x, sd)
        close(y, g[f"y_r{ratio}"], rtol=1e-5, atol=1e-6 * float(np.abs(g[f"y_r{ratio}"]).max()))
        with torch.no_grad():
            close(m(x), g[f"y_r{ratio}"], rtol=1e-5, atol=1e-6 * float(np.abs(g[f"y_r{ratio}"]).max()))   # the host mirror's plain-torch route
    h = copy.deepcopy(builtin_config("mini_coalign"))
    h["model"]["args"]["compression"] = int(g["model_ratio"])
    model = build_model(h).eval()
    fill_parameters_(model, seed=0, cls_bias=-1.0)
    assert list(model.state_dict().keys()) == list(g["state_keys"])
    gm = golden("model_mini.npz")
    batch = {"processed_lidar": {"voxel_features": T(gm["voxel_features"]), "voxel_coords": T(gm["voxel_coords"]),
                                 "voxel_num_points": T(gm["voxel_num_points"])},
             "record_len": T(gm["record_len"]), "pairwise_t_matrix": T(gm["pairwise_t_matrix"])}
    out = oracle.coalign_forward({k: v.clone() for k, v in model.state_dict().items()}, h["model"]["args"], batch)
    for k in ("cls_preds", "reg_preds", "dir_preds"):
        close(out[k], g[k], rtol=1e-4, atol=1e-4 * float(np.abs(g[k]).max()))
        assert float(np.abs(g[k] - gm[k]).max()) > 1e-3                            # the compressor really changes the outputs
