"""Pose correction by box alignment -- CoAlign's agent-object pose graph.

Mirrors ``opencood/models/sub_modules/box_align_v2.py`` (``box_alignment_relative_sample_np`` :101-396,
``box_alignment_relative_np`` :398-435) and ``pose_graph_optim.py`` (``PoseGraphOptimization2D`` :11-60).  The reference builds
one small graph per frame inside ``__getitem__`` and solves it with g2o on the CPU; here the graph construction stays host
code (a few hundred boxes, set logic) and the Levenberg-Marquardt solve is ``coalign_pose_graph_optimize`` on the device --
one workgroup per graph, so the graphs of a whole split (``tools/pose_graph_pre_calc.py`` style) go out in one launch
(``box_alignment_batch``).

Numerics kept from the reference: numpy inputs pass through ``pose_to_tfm`` / ``project_box3d`` in **float32**
(common_utils.check_numpy_to_torch), so world-frame centres, the 1.5 m clustering threshold and the landmark
initialisation are float32 quantities; measurements (boxes in each agent's own frame) and the information matrices stay
float64.  The cluster growth loop of the reference re-reads the seed's row of the distance matrix (:247), i.e. a cluster is
the seed box plus its still unassigned neighbours -- reproduced as such.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import ops

ANCHOR_DIAG_SQ = 1.6 ** 2 + 3.9 ** 2      # box_align_v2.py:187-189, hard-coded anchor width / length
MAX_AGENTS, MAX_LANDMARKS = 8, 256         # limits of coalign_pose_graph_optimize


def pose_to_tfm(pose: np.ndarray) -> np.ndarray:
    """[N, 3] (x, y, yaw) or [N, 6] (x, y, z, roll, yaw, pitch), degrees -> [N, 4, 4] float32 (transformation_utils.py:93-160).
    Evaluated with torch float32 operations like the reference (numpy input goes through ``check_numpy_to_torch(...).float()`` there):
    numpy's and torch's float32 cos / sin / deg2rad differ in the last bit now and then, and the clustering below thresholds float32
    distances that cancel to rounding noise for near-coincident detections."""
    p = torch.from_numpy(np.ascontiguousarray(np.asarray(pose))).float()
    n = p.shape[0]
    tfm = torch.eye(4).view(1, 4, 4).repeat(n, 1, 1)
    ang = lambda col: torch.deg2rad(p[:, col])
    if p.shape[1] == 3:
        c, s = torch.cos(ang(2)), torch.sin(ang(2))
        tfm[:, 0, 0], tfm[:, 0, 1], tfm[:, 1, 0], tfm[:, 1, 1] = c, -s, s, c
        tfm[:, 0, 3], tfm[:, 1, 3] = p[:, 0], p[:, 1]
        return tfm.numpy()
    cy, sy = torch.cos(ang(4)), torch.sin(ang(4))
    cr, sr = torch.cos(ang(3)), torch.sin(ang(3))
    cp, sp = torch.cos(ang(5)), torch.sin(ang(5))
    tfm[:, 0, 3], tfm[:, 1, 3], tfm[:, 2, 3] = p[:, 0], p[:, 1], p[:, 2]
    tfm[:, 0, 0], tfm[:, 0, 1], tfm[:, 0, 2] = cp * cy, cy * sp * sr - sy * cr, -cy * sp * cr - sy * sr
    tfm[:, 1, 0], tfm[:, 1, 1], tfm[:, 1, 2] = sy * cp, sy * sp * sr + cy * cr, -sy * sp * cr + cy * sr
    tfm[:, 2, 0], tfm[:, 2, 1], tfm[:, 2, 2] = sp, -cp * sr, cp * cr
    return tfm.numpy()


def corner_to_center(corner3d: np.ndarray, order: str = "lwh") -> np.ndarray:
    """[K, 8, 3] corners -> [K, 7] (x, y, z, l, w, h, yaw) or (.., h, w, l, yaw) (box_utils.py:25-85)."""
    c = np.asarray(corner3d)
    xy = c[:, :, :2]
    edge = lambda i, j: np.sqrt(((xy[:, i] - xy[:, j]) ** 2).sum(axis=1))
    heading = lambda i, j: np.arctan2(xy[:, i, 1] - xy[:, j, 1], xy[:, i, 0] - xy[:, j, 0])
    centre = c[:, [0, 3, 5, 6], :].mean(axis=1)
    h = np.abs((c[:, 4:, 2] - c[:, :4, 2]).mean(axis=1))
    l = (edge(0, 3) + edge(2, 1) + edge(4, 7) + edge(5, 6)) / 4
    w = (edge(0, 1) + edge(2, 3) + edge(4, 5) + edge(6, 7)) / 4
    yaw = (heading(1, 2) + heading(0, 3) + heading(5, 6) + heading(4, 7)) / 4
    dims = (l, w, h) if order == "lwh" else (h, w, l)
    return np.column_stack([centre, *dims, yaw])


def _project_f32(corners: np.ndarray, tfm: np.ndarray) -> np.ndarray:
    """box_utils.project_box3d on numpy input (:278-316): float32, the 4 x 4 applied to HOMOGENEOUS corners in one torch.matmul (the
    translation is the fourth product of each dot product, not a separate addition: other rounding)."""
    c = torch.from_numpy(np.ascontiguousarray(np.asarray(corners))).float().transpose(1, 2)
    c = torch.cat((c, torch.ones(c.shape[0], 1, 8)), dim=1)
    return torch.matmul(torch.from_numpy(np.asarray(tfm)).float(), c)[:, :3, :].transpose(1, 2).numpy()


class PoseGraph:
    """Flat description of one agent-object graph (what the reference adds to g2o vertex by vertex, edge by edge)."""

    def __init__(self, vertices, kinds, edge_agent, edge_landmark, edge_meas, edge_info, n_agents, clusters=None):
        self.vertices = np.asarray(vertices, dtype=np.float64).reshape(-1, 3)
        self.kinds = np.asarray(kinds, dtype=np.int32)
        self.edge_agent = np.asarray(edge_agent, dtype=np.int32)
        self.edge_landmark = np.asarray(edge_landmark, dtype=np.int32)
        self.edge_meas = np.asarray(edge_meas, dtype=np.float64).reshape(-1, 3)
        self.edge_info = np.asarray(edge_info, dtype=np.float64).reshape(-1, 3)
        self.n_agents = int(n_agents)
        self.clusters = clusters

    def check(self) -> None:
        n_lm = len(self.vertices) - self.n_agents
        if not (1 <= self.n_agents <= MAX_AGENTS and 0 <= n_lm <= MAX_LANDMARKS):
            raise ValueError(f"pose graph with {self.n_agents} agents / {n_lm} landmarks exceeds the kernel limits ({MAX_AGENTS} / {MAX_LANDMARKS})")
        if len(self.edge_landmark) and np.any(np.diff(self.edge_landmark) < 0):
            raise ValueError("edges must be grouped by landmark (ascending landmark id)")


def build_pose_graph(pred_corners_list: Sequence[np.ndarray], noisy_lidar_pose: np.ndarray, uncertainty_list=None, landmark_SE2=True,
                     adaptive_landmark=False, normalize_uncertainty=False, abandon_hard_cases=False, drop_hard_boxes=False,
                     drop_unsure_edge=False, use_uncertainty=True, thres=1.5, yaw_var_thres=0.2) -> Optional[PoseGraph]:
    """Everything of box_alignment_relative_sample_np in front of the solver (box_align_v2.py:150-372).  ``None`` means the
    hard-case rules decided to keep the noisy poses."""
    noisy_lidar_pose = np.asarray(noisy_lidar_pose)
    n_agents = noisy_lidar_pose.shape[0]
    counts = np.array([len(c) for c in pred_corners_list])
    tfm = pose_to_tfm(noisy_lidar_pose)
    local = np.concatenate([corner_to_center(c) for c in pred_corners_list if len(c)], axis=0)                       # float64, agent frames
    world = np.concatenate([corner_to_center(_project_f32(c, tfm[i])) for i, c in enumerate(pred_corners_list) if len(c)], axis=0)
    # float32, world frame.  CONTIGUOUS copies like the reference's np.concatenate of slices (:166-180): the all-pair distance below cancels
    # ~1e3-sized squares in float32, two detections of one object a centimetre apart land within rounding of zero, and whether sqrt sees a
    # tiny positive or a tiny negative number (NaN: "not near", the pair is never clustered) depends on the matmul's summation order --
    # which BLAS chooses by memory layout (found by the cfg-4 sweep test: a strided view paired two boxes the reference's layout leaves apart)
    centre, yaw = np.ascontiguousarray(world[:, :3]), np.ascontiguousarray(world[:, 6])
    owner = np.repeat(np.arange(n_agents), counts)
    certainty = None
    if use_uncertainty and uncertainty_list is not None:
        certainty = np.exp(-np.concatenate([np.asarray(u) for u in uncertainty_list if len(u)], axis=0))
        certainty[:, :2] /= ANCHOR_DIAG_SQ              # var(x) = d_a^2 var(x_t) for the anchor-normalised regression target
        if normalize_uncertainty:
            certainty = np.sqrt(certainty)
    sq = (centre * centre).sum(axis=1, keepdims=True)
    with np.errstate(invalid="ignore"):
        dist = np.sqrt(sq + sq.T - 2 * centre @ centre.T)                   # all_pair_l2 (:79-96), float32
    near = (dist < np.float32(thres)) & (owner[:, None] != owner[None, :])  # boxes of one agent never pair up
    free = np.ones(len(owner), dtype=bool)
    clusters: List[List[int]] = []
    landmarks, varies = [], []
    for seed in range(len(owner)):
        if not free[seed] or not near[seed].any():
            continue
        members = [seed] + [int(j) for j in np.nonzero(near[seed] & free)[0]]
        if len(members) == 1:                           # all neighbours already belong to earlier clusters
            free[seed] = False
            continue
        yaw_var = np.var(yaw[members])
        if landmark_SE2 and not (adaptive_landmark and yaw_var > yaw_var_thres):
            lm = np.array([centre[seed, 0], centre[seed, 1], yaw[seed]])
        else:
            lm = np.array([centre[seed, 0], centre[seed, 1]])
            if landmark_SE2:
                certainty[members] *= 2                 # adaptive landmark: the point constraint counts double (:276-279)
        clusters.append(members)
        landmarks.append(lm)
        varies.append(bool(yaw_var > yaw_var_thres))
        free[members] = False
    n_lm = len(clusters)
    if abandon_hard_cases and (n_lm <= 3 or sum(varies) >= 0.5 * n_lm):
        return None
    vertices = np.zeros((n_agents + n_lm, 3))
    kinds = np.ones(n_agents + n_lm, dtype=np.int32)
    vertices[:n_agents] = noisy_lidar_pose[:, [0, 1, 4]]
    vertices[:n_agents, 2] = np.deg2rad(vertices[:n_agents, 2])
    kinds[0] = 0                                        # the ego pose is the gauge
    e_agent, e_lm, e_meas, e_info = [], [], [], []
    for k, (members, lm) in enumerate(zip(clusters, landmarks)):
        se2 = len(lm) == 3
        vertices[n_agents + k, : len(lm)] = lm
        kinds[n_agents + k] = 1 if se2 else 2
        if drop_hard_boxes and varies[k]:
            continue
        for j in members:
            info = np.ones(3)
            if certainty is not None:
                if drop_unsure_edge and certainty[j].sum() < 100:
                    continue
                info[: len(lm)] = certainty[j][: len(lm)]
            meas = local[j][[0, 1, 6]].astype(np.float64)
            if not se2:
                meas[2] = info[2] = 0.0
            e_agent.append(owner[j]); e_lm.append(n_agents + k); e_meas.append(meas); e_info.append(info)
    return PoseGraph(vertices, kinds, e_agent, e_lm, e_meas, e_info, n_agents, clusters)


def optimize_pose_graphs(graphs: Sequence[PoseGraph], max_iterations: int = 1000, device="cuda:0") -> Tuple[List[np.ndarray], np.ndarray]:
    """Solve a batch of graphs in one launch -> (optimised vertices per graph, stats [G, 4] = iterations, chi2 before, chi2
    after, final lambda)."""
    if not graphs:
        return [], np.zeros((0, 4))
    for g in graphs:
        g.check()
    voff = np.concatenate([[0], np.cumsum([len(g.vertices) for g in graphs])]).astype(np.int32)
    eoff = np.concatenate([[0], np.cumsum([len(g.edge_agent) for g in graphs])]).astype(np.int32)
    dev = torch.device(device)
    cat = lambda xs, dt: torch.from_numpy(np.ascontiguousarray(np.concatenate(xs), dtype=dt)).to(dev)
    out, stats = ops.pose_graph_optimize(
        torch.from_numpy(voff).to(dev), torch.from_numpy(eoff).to(dev),
        torch.tensor([g.n_agents for g in graphs], dtype=torch.int32, device=dev),
        cat([g.vertices for g in graphs], np.float64), cat([g.kinds for g in graphs], np.int32),
        cat([g.edge_agent for g in graphs], np.int32), cat([g.edge_landmark for g in graphs], np.int32),
        cat([g.edge_meas for g in graphs], np.float64), cat([g.edge_info for g in graphs], np.float64), max_iterations)
    out, stats = out.cpu().numpy(), stats.cpu().numpy()
    if np.any(stats[:, 0] < 0):
        raise RuntimeError("coalign_pose_graph_optimize rejected a graph (structure outside the kernel limits)")
    return [out[voff[i]: voff[i + 1]] for i in range(len(graphs))], stats


def _refined(graph: Optional[PoseGraph], solved: Optional[np.ndarray], noisy_lidar_pose: np.ndarray) -> np.ndarray:
    if graph is None:
        return np.asarray(noisy_lidar_pose)[:, [0, 1, 4]]
    refined = solved[: graph.n_agents].copy()
    refined[:, 2] = np.rad2deg(refined[:, 2])           # radians -> degrees, like the input poses
    return refined


def box_alignment_relative_sample_np(pred_corners_list, noisy_lidar_pose, uncertainty_list=None, landmark_SE2=True,
                                     adaptive_landmark=False, normalize_uncertainty=False, abandon_hard_cases=False,
                                     drop_hard_boxes=False, drop_unsure_edge=False, use_uncertainty=True, thres=1.5,
                                     yaw_var_thres=0.2, max_iterations=1000, device="cuda:0") -> np.ndarray:
    """One sample: per-agent stage-1 boxes ``[[N_i, 8, 3]]`` (each in its agent's frame), noisy poses [N, 6] (degrees),
    log-variances ``[[N_i, 3]]`` -> refined (x, y, yaw) [N, 3] in degrees.  Same signature / defaults as the reference."""
    graph = build_pose_graph(pred_corners_list, noisy_lidar_pose, uncertainty_list, landmark_SE2, adaptive_landmark, normalize_uncertainty,
                             abandon_hard_cases, drop_hard_boxes, drop_unsure_edge, use_uncertainty, thres, yaw_var_thres)
    solved = optimize_pose_graphs([graph], max_iterations, device)[0][0] if graph is not None else None
    return _refined(graph, solved, noisy_lidar_pose)


def box_alignment_batch(samples: Sequence[dict], max_iterations: int = 1000, device="cuda:0", **flags) -> List[np.ndarray]:
    """Many samples (``{'pred_corners_list', 'noisy_lidar_pose', 'uncertainty_list'}`` each) -> refined poses, all graphs
    solved in ONE launch."""
    graphs = [build_pose_graph(s["pred_corners_list"], s["noisy_lidar_pose"], s.get("uncertainty_list"), **flags) for s in samples]
    live = [g for g in graphs if g is not None]
    solved = iter(optimize_pose_graphs(live, max_iterations, device)[0])
    return [_refined(g, next(solved) if g is not None else None, s["noisy_lidar_pose"]) for g, s in zip(graphs, samples)]


def box_alignment_relative_np(pred_corner3d_list, uncertainty_list, lidar_poses, record_len, **kwargs) -> np.ndarray:
    """box_align_v2.py:398-435: the samples of a collated batch, split by ``record_len``.  (The reference ends with
    ``np.cat``, which does not exist; the evident intent -- concatenation along axis 0 -- is what is returned.)"""
    samples, start = [], 0
    for b in [int(v) for v in record_len]:
        samples.append({"pred_corners_list": pred_corner3d_list[start: start + b], "noisy_lidar_pose": np.asarray(lidar_poses[start: start + b]),
                        "uncertainty_list": None if uncertainty_list is None else uncertainty_list[start: start + b]})
        start += b
    return np.concatenate(box_alignment_batch(samples, **kwargs), axis=0)


class SE2:
    """Stand-in for ``g2o.SE2`` at the interface ``PoseGraphOptimization2D`` exposes: built from (x, y, theta), ``vector()``."""

    def __init__(self, *args):
        self._v = np.asarray(args[0] if len(args) == 1 else args, dtype=np.float64).reshape(3).copy()

    def vector(self) -> np.ndarray:
        return self._v


class PoseGraphOptimization2D:
    """pose_graph_optim.py:11-60 with the same methods; ``optimize`` runs on the device."""

    def __init__(self, verbose: bool = False, device="cuda:0"):
        self._vertices: Dict[int, Tuple[np.ndarray, bool, bool]] = {}
        self._edges: List[Tuple[int, int, np.ndarray, np.ndarray, bool]] = []
        self._device = device
        self.verbose = verbose
        self.stats = None

    def add_vertex(self, id, pose, fixed=False, SE2=True):
        est = pose.vector() if hasattr(pose, "vector") else np.asarray(pose, dtype=np.float64)
        self._vertices[int(id)] = (np.asarray(est, dtype=np.float64), bool(fixed), bool(SE2))

    def add_edge(self, vertices, measurement, information=np.identity(3), robust_kernel=None, SE2=True):
        if robust_kernel is not None:
            raise NotImplementedError("robust kernels are not used on this path")
        info = np.asarray(information, dtype=np.float64)
        if not np.array_equal(info, np.diag(np.diag(info))):
            raise NotImplementedError("only diagonal information matrices (all the reference builds)")
        meas = measurement.vector() if hasattr(measurement, "vector") else np.asarray(measurement, dtype=np.float64)
        self._edges.append((int(vertices[0]), int(vertices[1]), np.asarray(meas, dtype=np.float64), np.diag(info).copy(), bool(SE2)))

    def optimize(self, max_iterations=1000):
        ids = sorted(self._vertices)
        if ids != list(range(len(ids))):
            raise ValueError("vertex ids must be 0..V-1")
        agents = sorted({a for a, *_ in self._edges} | {0})
        n_agents = max(agents) + 1
        vert = np.zeros((len(ids), 3)); kinds = np.zeros(len(ids), dtype=np.int32)
        for i in ids:
            est, fixed, se2 = self._vertices[i]
            vert[i, : len(est)] = est
            kinds[i] = 0 if fixed else (1 if se2 else 2)
        edges = sorted(self._edges, key=lambda e: e[1])         # stable: grouped by landmark, insertion order inside
        meas = np.zeros((len(edges), 3)); info = np.zeros((len(edges), 3))
        for k, (_, _, m, w, _) in enumerate(edges):
            meas[k, : len(m)] = m
            info[k, : len(w)] = w
        graph = PoseGraph(vert, kinds, [e[0] for e in edges], [e[1] for e in edges], meas, info, n_agents)
        (solved,), self.stats = optimize_pose_graphs([graph], max_iterations, self._device)
        for i in ids:
            est, fixed, se2 = self._vertices[i]
            self._vertices[i] = (solved[i, : len(est)].copy(), fixed, se2)

    def get_pose(self, id):
        est, _, se2 = self._vertices[int(id)]
        return SE2(est) if se2 else est
