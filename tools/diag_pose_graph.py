"""Diagnostic: the cfg-4 test's pose graph through the host graph builder (vs the oracle's) and, on a GPU, through the device solver."""
import math, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_round3_gpu import _dair_stage1_hypes, _plant_stage1_heads
from oracle import coalign_oracle as oracle
from coalign_amd import box_align
from coalign_amd.config import builtin_config
from coalign_amd.postprocess import build_postprocessor
h1 = _dair_stage1_hypes(); hd = builtin_config("dairv2x_coalign")
rs = np.random.RandomState(42)
clean = [np.zeros(6), np.array([30.0, 5.0, 0, 0, 170.0, 0])]
gx, gy = np.meshgrid(np.arange(-24, 72, 12.0), np.arange(-30, 31, 10.0))
world = np.stack([gx.ravel() + rs.uniform(-2, 2, gx.size), gy.ravel() + rs.uniform(-2, 2, gx.size)], 1)
yaw_w = rs.uniform(-2.5, 2.5, len(world))
pp1 = build_postprocessor(h1["postprocess"], False); anchors1 = pp1.generate_anchor_box(); rngd = hd["preprocess"]["cav_lidar_range"]
heads = {"cls_preds": [], "reg_preds": [], "unc_preds": []}
for pose in clean:
    th = math.radians(pose[4]); R = np.array([[math.cos(th), math.sin(th)], [-math.sin(th), math.cos(th)]])
    xy = (world - pose[:2]) @ R.T + rs.normal(0, 0.05, world.shape)
    inside = (xy[:, 0] > rngd[0] + 6) & (xy[:, 0] < rngd[3] - 6) & (xy[:, 1] > rngd[1] + 6) & (xy[:, 1] < rngd[4] - 6)
    obj = np.zeros((int(inside.sum()), 7)); obj[:, :2], obj[:, 2], obj[:, 3:6], obj[:, 6] = xy[inside], -1.0, [1.56, 2.0, 4.5], yaw_w[inside] - th
    c, r, u = _plant_stage1_heads(obj, anchors1, rs); heads["cls_preds"].append(c); heads["reg_preds"].append(r); heads["unc_preds"].append(u)
heads = {k: torch.from_numpy(np.concatenate(v)) for k, v in heads.items()}
co, bo, uo = oracle.post_process_stage1(heads, torch.from_numpy(anchors1), h1["postprocess"])
corners = [c.numpy().astype(np.float64) for c in co]; unc = [u.numpy().astype(np.float64) for u in uo]
flags = dict(use_uncertainty=True, landmark_SE2=True, adaptive_landmark=False, normalize_uncertainty=False, abandon_hard_cases=True, drop_hard_boxes=True)
noisy = np.array(clean)
go = oracle.build_pose_graph(corners, noisy, unc, **flags)
gh = box_align.build_pose_graph(corners, noisy, unc, True, False, False, True, True, False, True, 1.5, 0.2)
print("vertices equal", np.array_equal(go["vertices"], gh.vertices), "kinds", np.array_equal(go["kinds"], gh.kinds), "edges", [np.array_equal(a, b) for a, b in zip(go["edges"], (gh.edge_agent, gh.edge_landmark, gh.edge_meas, gh.edge_info))])
print("clusters oracle", len(go["clusters"]), "host", len(gh.clusters))
so = [tuple(c) for c in go["clusters"]]; sh = [tuple(c) for c in gh.clusters]
print("only oracle", [c for c in so if c not in sh][:5], "only host", [c for c in sh if c not in so][:5])
if len(go["vertices"]) == len(gh.vertices):
    print("max |dv|", np.abs(go["vertices"] - gh.vertices).max(), "V", len(gh.vertices), "E", len(gh.edge_agent))
x, st = oracle.pose_graph_lm(go["vertices"], go["kinds"], go["edges"], 1000)
print("oracle:", x[1], st)
if torch.cuda.is_available():
    sol, stats = box_align.optimize_pose_graphs([gh])
    print("device:", sol[0][1], "stats (it, chi0, chi, lambda)", stats[0].tolist(), "oracle chi2 at device solution", oracle.pose_graph_chi2(sol[0], gh.kinds, go["edges"]))
    gh2 = gh; gh2.vertices = sol[0].copy()
    sol2, stats2 = box_align.optimize_pose_graphs([gh2])
    print("device restart:", sol2[0][1], stats2[0].tolist())
    print("max |landmark diff| device vs oracle", np.abs(sol[0][2:] - x[2:]).max())
print("clusters oracle", len(go["clusters"]), "host", len(gh.clusters))
so = [tuple(c) for c in go["clusters"]]; sh = [tuple(c) for c in gh.clusters]
print("only oracle", [c for c in so if c not in sh][:5], "only host", [c for c in sh if c not in so][:5])
