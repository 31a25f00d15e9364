#!/usr/bin/env python
"""Generate the golden fixtures in this directory by IMPORTING THE REFERENCE ITSELF.

Runs only in the build container (needs /root/reference; it never travels to the GPU box, the
``.npz`` files do).  Usage:  python tests/golden/make_golden.py

What is pinned: every function of the hot path that the reference can execute here is
called unmodified -- ``create_model`` on the unchanged OPV2V CoAlign yaml (only ``cav_lidar_range`` is
shrunk for the mini cases), ``PillarVFE``, ``PointPillarScatter``, ``normalize_pairwise_tfm``,
``warp_affine_simple``, ``AttFusion`` / ``MaxFusion``, the ResNet backbone + heads,
``VoxelPostprocessor.generate_anchor_box`` / ``post_process`` and ``box_utils.nms_rotated``.

Optional third-party modules the reference imports at module scope but that are absent here are
replaced by inert stubs (icecream, pyquaternion, turtle, cv2, open3d, the un-built Cython
``box_overlaps``, pypcd).  ``g2o`` (pose-graph solver, absent) is a recording stand-in, see ``_G2O``;
torchvision / pypcd are inert and ``opencood.data_utils.datasets`` is registered as a bare package so that only the
intermediate-fusion dataset module is imported.  ``shapely.geometry.Polygon`` -- the one stub that *computes* something -- is backed
by the oracle's fp64 clipping routine: this pins the reference's NMS control flow (argsort, top-1000,
float32 IoU array, strict ``>``) but NOT the GEOS area arithmetic (parity unpinned).
"""
import hashlib
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)

from oracle import coalign_oracle as oracle  # noqa: E402
from coalign_amd.synthetic import fill_parameters_, make_frame  # noqa: E402  (input generators only)

REF = "/root/reference"
YAML_COALIGN = REF + "/opencood/hypes_yaml/opv2v/lidar_only_with_noise/coalign/pointpillar_coalign.yaml"
YAML_SINGLE = REF + "/opencood/hypes_yaml/opv2v/lidar_only_with_noise/pointpillar_single.yaml"
YAML_UNC = REF + "/opencood/hypes_yaml/opv2v/lidar_only_with_noise/coalign/pointpillar_uncertainty.yaml"
YAML_DAIR = REF + "/opencood/hypes_yaml/dairv2x/lidar_only_with_noise/coalign/pointpillar_coalign.yaml"
MINI_RANGE = [-12.8, -6.4, -3, 12.8, 6.4, 1]


class _OraclePolygon:
    """Minimal stand-in for shapely.geometry.Polygon (area / intersection / union of convex quads)."""

    def __init__(self, pts=None, area=None):
        self.pts = None if pts is None else np.asarray(pts, dtype=np.float64)
        self._area = area

    @property
    def area(self):
        if self._area is not None:
            return self._area
        x, y = self.pts[:, 0], self.pts[:, 1]
        return abs(0.5 * float(np.sum(x * np.roll(y, -1) - np.roll(x, -1) * y)))

    def intersection(self, other):
        return _OraclePolygon(area=oracle.quad_intersection_area(self.pts, other.pts))

    def union(self, other):
        return _OraclePolygon(area=self.area + other.area - oracle.quad_intersection_area(self.pts, other.pts))


class _G2O:
    """Recording stand-in for the python `g2o` binding (absent here): keeps the vertices / edges the reference adds and, in
    optimize(), hands them to the oracle's restatement of g2o's Levenberg-Marquardt.  Pins everything of box_align_v2 except
    the optimiser's own arithmetic."""
    recorded = []

    class SE2:
        def __init__(self, *a):
            self.v = np.asarray(a[0] if len(a) == 1 else a, dtype=np.float64).reshape(3).copy()

        def vector(self):
            return self.v

    class _Vertex:
        kind = 1

        def set_estimate(self, e): self.e = e
        def set_id(self, i): self.i = i
        def set_fixed(self, f): self.f = f
        def estimate(self): return self.e

    class VertexSE2(_Vertex):
        kind = 1

    class VertexPointXY(_Vertex):
        kind = 2

    class _Edge:
        def __init__(self): self.vs = [None, None]; self.rk = None
        def set_vertex(self, i, v): self.vs[i] = v
        def set_measurement(self, m): self.m = m
        def set_information(self, w): self.w = np.asarray(w, dtype=np.float64)
        def set_robust_kernel(self, k): self.rk = k

    class EdgeSE2(_Edge):
        pass

    class EdgeSE2PointXY(_Edge):
        pass

    class SparseOptimizer:
        def __init__(self): self.V, self.E = {}, []
        def set_algorithm(self, a): pass
        def set_verbose(self, v): pass
        def add_vertex(self, v): self.V[v.i] = v
        def add_edge(self, e): self.E.append(e)
        def vertex(self, i): return self.V[i]
        def initialize_optimization(self): pass

        def optimize(self, n):
            ids = sorted(self.V)
            assert ids == list(range(len(ids)))
            vert = np.zeros((len(ids), 3)); kinds = np.zeros(len(ids), dtype=np.int32)
            for i in ids:
                v = self.V[i]
                est = v.e.vector() if isinstance(v.e, _G2O.SE2) else np.asarray(v.e, dtype=np.float64)
                vert[i, : len(est)] = est
                kinds[i] = 0 if v.f else v.kind
            ea = np.array([e.vs[0].i for e in self.E], dtype=np.int32); el = np.array([e.vs[1].i for e in self.E], dtype=np.int32)
            meas = np.zeros((len(self.E), 3)); info = np.zeros((len(self.E), 3))
            for k, e in enumerate(self.E):
                m = e.m.vector() if isinstance(e.m, _G2O.SE2) else np.asarray(e.m, dtype=np.float64)
                meas[k, : len(m)] = m
                assert np.array_equal(e.w, np.diag(np.diag(e.w))) and e.rk is None
                info[k, : len(m)] = np.diag(e.w)
            edges = (ea, el, meas, info)
            x, stats = oracle.pose_graph_lm(vert, kinds, edges, n)
            _G2O.recorded.append({"vertices": vert, "kinds": kinds, "edges": edges, "solution": x, "iterations": stats["iterations"]})
            for i in ids:
                v = self.V[i]
                v.e = _G2O.SE2(x[i]) if isinstance(v.e, _G2O.SE2) else x[i, :2].copy()

    BlockSolverSE2 = LinearSolverDenseSE2 = OptimizationAlgorithmLevenberg = staticmethod(lambda *a: None)
    BlockSolverSE3 = LinearSolverCholmodSE3 = staticmethod(lambda *a: None)


def install_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    mod("icecream", ic=lambda *a, **k: None)
    mod("pyquaternion", Quaternion=object)
    sh = mod("shapely")
    sh.geometry = mod("shapely.geometry", Polygon=_OraclePolygon, Point=object, MultiPoint=object)
    tv = mod("torchvision")
    _inert = type("Inert", (), {"__init__": lambda self, *a, **k: None, "__call__": lambda self, x: x})
    tv.transforms = mod("torchvision.transforms", Normalize=_inert, Compose=_inert, ToTensor=_inert, ToPILImage=_inert)
    # the datasets package __init__ pulls every dataset family (and their dependencies): expose the directory only
    dsets = mod("opencood.data_utils.datasets")
    dsets.__path__ = [REF + "/opencood/data_utils/datasets"]
    mod("turtle", update=None)
    mod("cv2")
    mod("open3d")
    mod("pypcd").pypcd = mod("pypcd.pypcd")
    mod("g2o", **{k: getattr(_G2O, k) for k in dir(_G2O) if not k.startswith("__")})
    mod("opencood.utils.box_overlaps", bbox_overlaps=None)
    sys.path.insert(0, REF)


def load_hypes(path, lidar_range=None):
    from opencood.hypes_yaml import yaml_utils
    h = yaml_utils.load_yaml(path)
    if lidar_range is not None:
        h["preprocess"]["cav_lidar_range"][:] = lidar_range      # shared by the yaml anchors
        h = yaml_utils.load_point_pillar_params(h)
    return h


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrs.items()})
    print(f"wrote {name}: {os.path.getsize(path) / 1024:.0f} KiB")


def main():
    install_stubs()
    from opencood.tools import train_utils
    from opencood.data_utils.post_processor import build_postprocessor
    from opencood.models.fuse_modules.fusion_in_one import AttFusion, MaxFusion
    from opencood.models.sub_modules.torch_transformation_utils import warp_affine_simple
    from opencood.utils.transformation_utils import normalize_pairwise_tfm, x_to_world, get_pairwise_transformation
    from opencood.utils import box_utils

    torch.manual_seed(0)
    np.random.seed(303)
    torch.set_grad_enabled(False)

    # ------------------------------------------------------------------ mini CoAlign model, end to end
    h = load_hypes(YAML_COALIGN, MINI_RANGE)
    model = train_utils.create_model(h).eval()
    fill_parameters_(model, seed=0, cls_bias=-1.0)
    frame = make_frame(h, [3, 2], pillars_per_agent=120, seed=11, spread_xy=(4.0, 2.0), spread_yaw=40.0)
    bd = {"voxel_features": frame["processed_lidar"]["voxel_features"],
          "voxel_coords": frame["processed_lidar"]["voxel_coords"],
          "voxel_num_points": frame["processed_lidar"]["voxel_num_points"],
          "record_len": frame["record_len"]}
    bd = model.pillar_vfe(bd)
    bd = model.scatter(bd)
    canvas = bd["spatial_features"]
    H0, W0 = canvas.shape[2:]
    aff = normalize_pairwise_tfm(frame["pairwise_t_matrix"], H0, W0, model.voxel_size[0])
    feats = model.backbone.get_multiscale_feature(canvas)
    fused = [m(f, frame["record_len"], aff) for m, f in zip(model.fusion_net, feats)]
    out = model(frame)
    sd = model.state_dict()
    pfx = "pillar_vfe.pfn_layers.0."
    save("model_mini.npz",
         voxel_features=bd["voxel_features"], voxel_coords=bd["voxel_coords"], voxel_num_points=bd["voxel_num_points"],
         record_len=frame["record_len"], pairwise_t_matrix=frame["pairwise_t_matrix"],
         pfn_weight=sd[pfx + "linear.weight"], pfn_bn_weight=sd[pfx + "norm.weight"], pfn_bn_bias=sd[pfx + "norm.bias"],
         pfn_bn_mean=sd[pfx + "norm.running_mean"], pfn_bn_var=sd[pfx + "norm.running_var"],
         pillar_features=bd["pillar_features"], canvas_sample=canvas.reshape(-1)[::7],
         canvas_nonzero=torch.nonzero(canvas.reshape(-1)).view(-1).int(),
         normalized_affine=aff,
         feat0_sample=feats[0].reshape(-1)[::5], feat1_sample=feats[1].reshape(-1)[::5], feat2_sample=feats[2].reshape(-1)[::5],
         fused0=fused[0], fused1=fused[1], fused2=fused[2],
         cls_preds=out["cls_preds"], reg_preds=out["reg_preds"], dir_preds=out["dir_preds"],
         fill_seed=0, cls_bias=-1.0, n_state=len(sd), n_params=sum(p.numel() for p in model.parameters()),
         state_keys=np.array(list(sd.keys())), state_numel=np.array([v.numel() for v in sd.values()]))
    # single-agent PointPillar (late fusion, cfg 1) on the mini canvas: state_dict names + one forward
    hl = load_hypes(YAML_SINGLE, MINI_RANGE)
    ml = train_utils.create_model(hl).eval()
    fill_parameters_(ml, seed=0, cls_bias=-1.0)
    fl = make_frame(hl, 1, pillars_per_agent=150, seed=12)
    ol = ml(fl)
    sdl = ml.state_dict()
    save("late_mini.npz", voxel_features=fl["processed_lidar"]["voxel_features"], voxel_coords=fl["processed_lidar"]["voxel_coords"],
         voxel_num_points=fl["processed_lidar"]["voxel_num_points"], cls_preds=ol["cls_preds"], reg_preds=ol["reg_preds"],
         dir_preds=ol["dir_preds"], state_keys=np.array(list(sdl.keys())), state_numel=np.array([v.numel() for v in sdl.values()]))

    # ------------------------------------------------------------------ pose algebra
    poses = [[0, 0, 0, 0, 0, 0], [12.5, -3.25, 0.4, 1.0, 33.0, -2.0], [-7.0, 8.0, 0.1, 0.0, -121.0, 0.5]]
    worlds = np.stack([x_to_world(p) for p in poses])
    pair = get_pairwise_transformation({i: {"params": {"lidar_pose": p}} for i, p in enumerate(poses)}, 5, False)
    pt = torch.from_numpy(pair[None])
    save("pose.npz", poses=np.array(poses), x_to_world=worlds, pairwise=pair,
         normalized_200x704=normalize_pairwise_tfm(pt, 200, 704, 0.4), normalized_32x64=normalize_pairwise_tfm(pt, 32, 64, 0.4),
         pairwise_after=pt)   # pairwise_after == pairwise proves the input is not modified

    # ------------------------------------------------------------------ warp cases
    g = torch.Generator().manual_seed(5)
    src = torch.randn(6, 8, 16, 32, generator=g)
    deg = np.radians
    def rot(th, tx=0.0, ty=0.0, H=16, W=32):
        c, s = np.cos(th), np.sin(th)
        return [[c, -s * H / W, tx], [s * W / H, c, ty]]
    theta = torch.tensor([[[1, 0, 0], [0, 1, 0]],                     # identity
                          [[1, 0, 2 * 2.3 / 32], [0, 1, -2 * 1.7 / 16]],  # translation by non-integer pixels
                          rot(deg(90)), rot(deg(37), 0.11, -0.07), rot(deg(-170), 0.4, 0.3),
                          [[1, 0, 3.0], [0, 1, 0]]], dtype=torch.float64)   # fully out of range -> zeros
    save("warp.npz", src=src, theta=theta, warped=warp_affine_simple(src, theta, (16, 32)))

    # ------------------------------------------------------------------ fusion at the real channel widths
    rl = torch.tensor([3, 2])
    L = 5
    g = torch.Generator().manual_seed(7)
    pair = np.stack([get_pairwise_transformation({i: {"params": {"lidar_pose": p}} for i, p in enumerate(ps)}, L, False)
                     for ps in ([[0, 0, 0, 0, 0, 0], [2.0, -1.0, 0, 0, 25.0, 0], [-3.0, 1.5, 0, 0, -100.0, 0]],
                                [[0, 0, 0, 0, 0, 0], [1.0, 0.5, 0, 0, 5.0, 0]])])
    aff_f = normalize_pairwise_tfm(torch.from_numpy(pair), 16, 32, 0.4)
    fus = {"record_len": rl, "affine": aff_f}
    for s, (C, H, W) in enumerate(((64, 8, 16), (128, 4, 8), (256, 2, 4))):
        x = torch.randn(5, C, H, W, generator=g)
        fus[f"x{s}"] = x
        fus[f"att{s}"] = AttFusion(C)(x, rl, aff_f)
        fus[f"max{s}"] = MaxFusion()(x, rl, aff_f)
    save("fusion.npz", **fus)

    # ------------------------------------------------------------------ anchors for every config
    anc = {}
    for tag, path, rng in (("opv2v_coalign", YAML_COALIGN, None), ("opv2v_late", YAML_SINGLE, None),
                           ("dairv2x_coalign", YAML_DAIR, None), ("mini", YAML_COALIGN, MINI_RANGE)):
        hh = load_hypes(path, rng)
        pp = build_postprocessor(hh["postprocess"], False)
        a = pp.generate_anchor_box()
        anc[tag + "_shape"] = np.array(a.shape)
        anc[tag + "_first"] = a.reshape(-1, 7)[0]
        anc[tag + "_last"] = a.reshape(-1, 7)[-1]
        anc[tag + "_sha256"] = np.frombuffer(bytes.fromhex(sha(a)), dtype=np.uint8)
        anc[tag + "_grid_size"] = np.asarray(hh["model"]["args"]["point_pillar_scatter"]["grid_size"])
        anc[tag + "_WHD"] = np.array([hh["postprocess"]["anchor_args"][k] for k in "WHD"])
        if tag == "mini":
            anc["mini_full"] = a
    save("anchors.npz", **anc)

    # ------------------------------------------------------------------ post-processing (decode + filters + NMS + range)
    h = load_hypes(YAML_COALIGN, MINI_RANGE)
    pp = build_postprocessor(h["postprocess"], False)
    anchors = torch.from_numpy(pp.generate_anchor_box())
    g = torch.Generator().manual_seed(21)
    Hh, Ww = anchors.shape[:2]

    def rand_heads(bias):
        cls = torch.randn(1, 2, Hh, Ww, generator=g) * 1.5 + bias
        reg = torch.randn(1, 14, Hh, Ww, generator=g) * 0.35
        reg[:, [2, 9]] = reg[:, [2, 9]] * 0.2                      # keep z inside [-3, 1] for most boxes
        dirp = torch.randn(1, 4, Hh, Ww, generator=g)
        return cls, reg, dirp

    post = {"anchors": anchors}
    # intermediate fusion: one 'ego' entry, identity transform
    cls, reg, dirp = rand_heads(-1.2)
    data = {"ego": {"transformation_matrix": torch.eye(4), "anchor_box": anchors}}
    outd = {"ego": {"cls_preds": cls, "reg_preds": reg, "dir_preds": dirp}}
    boxes, scores = pp.post_process(data, outd)
    post.update(i_cls=cls, i_reg=reg, i_dir=dirp, i_boxes=boxes, i_scores=scores,
                i_delta_boxes=pp.delta_to_boxes3d(reg, anchors))
    # late fusion: two agents, second one projected with a non-trivial T (float32 like late_fusion_dataset.py:464-466)
    T1 = torch.from_numpy(np.linalg.solve(x_to_world([0, 0, 0, 0, 0, 0]), x_to_world([3.0, -1.0, 0.0, 0, 12.0, 0]))).float()
    c0, r0, d0 = rand_heads(-1.6)
    c1, r1, d1 = rand_heads(-1.6)
    data = {"a0": {"transformation_matrix": torch.eye(4), "anchor_box": anchors},
            "a1": {"transformation_matrix": T1, "anchor_box": anchors}}
    outd = {"a0": {"cls_preds": c0, "reg_preds": r0, "dir_preds": d0}, "a1": {"cls_preds": c1, "reg_preds": r1, "dir_preds": d1}}
    boxes, scores = pp.post_process(data, outd)
    post.update(l_cls0=c0, l_reg0=r0, l_dir0=d0, l_cls1=c1, l_reg1=r1, l_dir1=d1, l_T1=T1, l_boxes=boxes, l_scores=scores)
    # nothing above threshold -> (None, None)
    cN = torch.full((1, 2, Hh, Ww), -9.0)
    outd = {"ego": {"cls_preds": cN, "reg_preds": reg, "dir_preds": dirp}}
    bN, sN = pp.post_process({"ego": {"transformation_matrix": torch.eye(4), "anchor_box": anchors}}, outd)
    post.update(none_result=np.array([bN is None, sN is None]),
                score_threshold=h["postprocess"]["target_args"]["score_threshold"], nms_thresh=h["postprocess"]["nms_thresh"],
                gt_range=np.array(h["postprocess"]["gt_range"], dtype=np.float64))
    # geometry helpers on a handful of boxes
    b7 = torch.tensor([[1.0, 2.0, -1.0, 1.56, 1.6, 3.9, 0.3], [-4.0, 0.5, -0.8, 1.4, 1.9, 4.4, -2.0],
                       [0.0, 0.0, -1.0, 1.5, 7.0, 3.0, 0.0], [2.0, 2.0, -2.9, 2.5, 1.6, 3.9, 1.0]])
    c8 = box_utils.boxes_to_corners_3d(b7, "hwl")
    post.update(g_boxes7=b7, g_corners=c8, g_proj=box_utils.project_box3d(c8, T1),
                g_standup=box_utils.corner_to_standup_box_torch(c8),
                g_keep_large=box_utils.remove_large_pred_bbx(c8), g_keep_z=box_utils.remove_bbx_abnormal_z(c8))
    save("postprocess.npz", **post)

    # ------------------------------------------------------------------ rotated NMS through the reference's control flow
    rs = np.random.RandomState(99)
    nms = {}
    for tag, K, spread in (("small", 60, 12.0), ("mid", 400, 40.0), ("over1000", 1500, 60.0)):
        b7 = np.zeros((K, 7), dtype=np.float32)
        b7[:, 0] = rs.uniform(-spread, spread, K); b7[:, 1] = rs.uniform(-spread / 3, spread / 3, K); b7[:, 2] = -1
        b7[:, 3] = 1.56; b7[:, 4] = rs.uniform(1.4, 2.2, K); b7[:, 5] = rs.uniform(3.0, 5.5, K); b7[:, 6] = rs.uniform(-3.2, 3.2, K)
        corners = box_utils.boxes_to_corners_3d(torch.from_numpy(b7), "hwl")
        sc = torch.from_numpy(rs.uniform(0.2, 1.0, K).astype(np.float32))
        keep = box_utils.nms_rotated(corners, sc, 0.15)
        nms.update({f"{tag}_corners": corners, f"{tag}_scores": sc, f"{tag}_keep": keep})
    nms["empty_keep"] = box_utils.nms_rotated(torch.zeros(0, 8, 3), torch.zeros(0), 0.15)
    nms["quad_keep"] = box_utils.nms_rotated(corners[:50, :4, :2], sc[:50], 0.15)   # (N,4,2) input form
    save("nms.npz", **nms)

    # ------------------------------------------------------------------ evaluation: the reference's TP/FP matching + AP
    from opencood.utils import eval_utils
    rs = np.random.RandomState(31)
    ev = {}
    stat = {0.3: {"tp": [], "fp": [], "gt": 0, "score": []}, 0.5: {"tp": [], "fp": [], "gt": 0, "score": []},
            0.7: {"tp": [], "fp": [], "gt": 0, "score": []}}
    for f in range(3):
        G = 25 + 5 * f
        g7 = np.zeros((G, 7), dtype=np.float32)
        g7[:, 0] = rs.uniform(-60, 60, G); g7[:, 1] = rs.uniform(-30, 30, G); g7[:, 2] = -1; g7[:, 3] = 1.56
        g7[:, 4] = rs.uniform(1.5, 2.1, G); g7[:, 5] = rs.uniform(3.5, 5.0, G); g7[:, 6] = rs.uniform(-3.1, 3.1, G)
        d7 = np.concatenate([g7[: G - 5] + rs.normal(0, [0.35, 0.25, 0, 0, 0.1, 0.25, 0.08], (G - 5, 7)).astype(np.float32),
                             g7[:8] + np.array([9.0, 7.0, 0, 0, 0, 0, 0.5], dtype=np.float32)])        # jittered hits + false alarms
        gt_c = box_utils.boxes_to_corners_3d(torch.from_numpy(g7), "hwl")
        det_c = box_utils.boxes_to_corners_3d(torch.from_numpy(d7), "hwl")
        det_s = torch.from_numpy(rs.uniform(0.2, 1.0, len(d7)).astype(np.float32))
        for thr in (0.3, 0.5, 0.7):
            eval_utils.caluclate_tp_fp(det_c, det_s, gt_c, stat, thr)
        ev.update({f"gt{f}": gt_c, f"det{f}": det_c, f"score{f}": det_s})
    eval_utils.caluclate_tp_fp(None, None, gt_c, stat, 0.7)          # a frame without detections only adds ground truths
    for thr in (0.3, 0.5, 0.7):
        ap, mrec, mpre = eval_utils.calculate_ap(stat, thr)
        tag = str(int(thr * 100))
        ev.update({f"tp{tag}": np.array(stat[thr]["tp"]), f"fp{tag}": np.array(stat[thr]["fp"]), f"gtn{tag}": stat[thr]["gt"],
                   f"scores{tag}": np.array(stat[thr]["score"]), f"ap{tag}": ap, f"mrec{tag}": np.array(mrec), f"mpre{tag}": np.array(mpre)})
    # ground-truth boxes the way the inference loop builds them (base_postprocessor.py:46-106): two agents, shared ids
    from opencood.data_utils.post_processor.voxel_postprocessor import VoxelPostprocessor as RefPost
    hg = load_hypes(YAML_COALIGN)
    rp = RefPost(hg["postprocess"], train=False)
    gt_in = {}
    for c, (ids, T) in enumerate((([3, 7, 11, 20, 21, 40], np.eye(4)),
                                  ([7, 50, 3, 60, 61], x_to_world([6.0, -2.0, 0.1, 0.0, 35.0, 0.0])))):
        M = 100
        centre = np.zeros((M, 7), dtype=np.float32)
        mask = np.zeros(M, dtype=np.float32)
        mask[: len(ids)] = 1
        centre[: len(ids), 0] = rs.uniform(-150, 150, len(ids)); centre[: len(ids), 1] = rs.uniform(-45, 45, len(ids))
        centre[: len(ids), 2] = rs.uniform(-2.5, 0.5, len(ids)); centre[: len(ids), 3] = 1.56
        centre[: len(ids), 4] = 2.0; centre[: len(ids), 5] = 4.5; centre[: len(ids), 6] = rs.uniform(-3.1, 3.1, len(ids))
        gt_in[c] = {"transformation_matrix_clean": torch.from_numpy(T.astype(np.float32)), "object_bbx_center": torch.from_numpy(centre),
                    "object_bbx_mask": torch.from_numpy(mask), "object_ids": list(ids)}
        ev.update({f"gtgen_centre{c}": centre, f"gtgen_mask{c}": mask, f"gtgen_ids{c}": np.array(ids), f"gtgen_T{c}": T.astype(np.float32)})
    ev["gtgen_out"] = rp.generate_gt_bbx(gt_in)
    save("eval.npz", **ev)

    # ------------------------------------------------------------------ point filters in front of the voxeliser (next-1)
    from opencood.utils import pcd_utils
    rs = np.random.RandomState(41)
    cloud = rs.uniform([-150, -45, -4, 0], [150, 45, 2, 1], (1500, 4)).astype(np.float32)
    cloud[:200, :2] = rs.uniform([-3, -2], [4, 2], (200, 2))                                   # around the ego body
    edge = np.float32([-140.8, -40, -3, 140.8, 40, 1, -1.95, 2.95, -1.1, 1.1])
    cloud[200:210, 0] = [edge[0], edge[3], np.nextafter(edge[0], np.float32(0)), np.nextafter(edge[3], np.float32(0)), edge[6], edge[7],
                         np.nextafter(edge[6], np.float32(-9)), np.nextafter(edge[7], np.float32(9)), 0.5, 0.5]
    cloud[200:210, 1] = [0, 0, 0, 0, 0.3, 0.3, 0.3, 0.3, edge[8], np.nextafter(edge[9], np.float32(9))]
    cloud[200:210, 2] = -1
    save("points.npz", cloud=cloud, range_masked=pcd_utils.mask_points_by_range(cloud, [-140.8, -40, -3, 140.8, 40, 1]),
         ego_masked=pcd_utils.mask_ego_points(cloud))

    # ------------------------------------------------------------------ batch-dict producer (next-4): the reference's dataset class
    import copy
    import json
    from collections import OrderedDict
    from opencood.data_utils.datasets.intermediate_fusion_dataset import getIntermediateFusionDataset
    from opencood.data_utils.post_processor import build_postprocessor as ref_build_post
    from opencood.data_utils.pre_processor.sp_voxel_preprocessor import SpVoxelPreprocessor as RefSpVox
    from coalign_amd.synthetic import make_point_cloud

    class _VoxStub(RefSpVox):
        """The reference's pre-processor with spconv's generator (absent) replaced by the oracle's restatement of it; the
        collate functions are the reference's own."""
        def __init__(self, params, train):                         # noqa: super().__init__ imports spconv
            self.params, self.train = params, train
            self.lidar_range, self.voxel_size = params["cav_lidar_range"], params["args"]["voxel_size"]
            self.max_points, self.max_voxels = params["args"]["max_points_per_voxel"], params["args"]["max_voxel_test"]

        def preprocess(self, pcd_np):
            v, c, n = oracle.points_to_voxel(pcd_np, self.voxel_size, self.lidar_range, self.max_points, self.max_voxels)
            return {"voxel_features": v, "voxel_coords": c, "voxel_num_points": n}

    class _MemoryBase:
        """Stands in for basedataset (disk readers): serves one in-memory scenario."""
        def __init__(self, params, visualize, train=True):
            self.params, self.visualize, self.train = params, visualize, train
            self.pre_processor = _VoxStub(params["preprocess"], train)
            self.post_processor = ref_build_post(params["postprocess"], train)
            self.post_processor.generate_label = lambda **kw: {}       # anchor targets: training only (needs the Cython overlaps)
            self.post_processor.collate_batch = lambda lst: {}
            self.max_cav = params["train_params"]["max_cav"]
            self.load_lidar_file, self.load_camera_file, self.load_depth_file = True, False, False
            self.scenario = None

        def retrieve_base_data(self, idx):
            return copy.deepcopy(self.scenario)

        def generate_object_center(self, cav_contents, reference_lidar_pose):
            return self.post_processor.generate_object_center(cav_contents, reference_lidar_pose)

    def memory_scenario(seed, n_cav=3, n_obj=40, far=True):
        rs = np.random.RandomState(seed)
        vehicles = OrderedDict()
        for k in range(n_obj):
            vehicles[100 + k] = {"location": [float(rs.uniform(-120, 120)), float(rs.uniform(-38, 38)), float(rs.uniform(-0.2, 0.2))],
                                 "angle": [0.0, float(rs.uniform(-180, 180)), 0.0], "extent": [float(rs.uniform(1.8, 2.6)), float(rs.uniform(0.8, 1.1)), float(rs.uniform(0.7, 0.9))],
                                 "center": [0.0, 0.0, float(rs.uniform(0.6, 0.9))]}
        sc = OrderedDict()
        for c in range(n_cav):
            pose = [float(rs.uniform(-25, 25)), float(rs.uniform(-8, 8)), 1.9, 0.0, float(rs.uniform(-180, 180)), 0.0]
            if far and c == n_cav - 1:
                pose[0] += 200.0                                       # beyond comm_range: must be dropped
            seen = OrderedDict((k, v) for k, v in vehicles.items() if rs.uniform() < 0.7)
            sc[str(c * 7 + 1)] = {"ego": c == 0, "params": {"lidar_pose": pose, "vehicles": seen},
                                  "lidar_np": make_point_cloud(seed * 10 + c, beams=16, azimuth_steps=450)}
        return sc

    IFD = getIntermediateFusionDataset(_MemoryBase)
    hd = load_hypes(YAML_COALIGN)
    hd.pop("box_align", None)
    ds = IFD(hd, visualize=False, train=False)
    dsg = {}
    for tag, seed, n_cav in (("a", 3, 4), ("b", 4, 2)):
        ds.scenario = memory_scenario(seed, n_cav)
        np.random.seed(1000 + seed)
        batch = ds.collate_batch_test([ds[0]])["ego"]
        for c, (cid, cav) in enumerate(ds.scenario.items()):
            dsg.update({f"{tag}_lidar{c}": cav["lidar_np"], f"{tag}_pose{c}": np.array(cav["params"]["lidar_pose"]),
                        f"{tag}_veh_ids{c}": np.array(list(cav["params"]["vehicles"].keys())),
                        f"{tag}_veh{c}": np.array([v["location"] + v["angle"] + v["extent"] + v["center"] for v in cav["params"]["vehicles"].values()]).reshape(-1, 12)})
        dsg.update({f"{tag}_n_cav": n_cav, f"{tag}_cav_ids": np.array(list(ds.scenario.keys())), f"{tag}_np_seed": 1000 + seed,
                    f"{tag}_voxel_features": batch["processed_lidar"]["voxel_features"], f"{tag}_voxel_coords": batch["processed_lidar"]["voxel_coords"],
                    f"{tag}_voxel_num_points": batch["processed_lidar"]["voxel_num_points"], f"{tag}_record_len": batch["record_len"],
                    f"{tag}_pairwise_t_matrix": batch["pairwise_t_matrix"], f"{tag}_object_bbx_center": batch["object_bbx_center"],
                    f"{tag}_object_bbx_mask": batch["object_bbx_mask"], f"{tag}_object_ids": np.array(batch["object_ids"]),
                    f"{tag}_lidar_pose": batch["lidar_pose"], f"{tag}_lidar_pose_clean": batch["lidar_pose_clean"],
                    f"{tag}_cav_id_list": np.array(batch["cav_id_list"]), f"{tag}_transformation_matrix": batch["transformation_matrix"]})
        print(f"  dataset {tag}: cavs kept {batch['cav_id_list']}, voxels {tuple(batch['processed_lidar']['voxel_features'].shape)}, objects {int(batch['object_bbx_mask'].sum())}")
    np.savez_compressed(os.path.join(HERE, "dataset.npz"), **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in dsg.items()})
    print(f"wrote dataset.npz: {os.path.getsize(os.path.join(HERE, 'dataset.npz')) // 1024} KiB")

    # ------------------------------------------------------------------ the dataset's side branches (round 3): Laplace pose noise
    # (pose_utils.py:19-21, 77-105) and proj_first = True (intermediate_fusion_dataset.py:43-44, 104; transformation_utils.py:43-49),
    # and iou_preds rescoring in the post-processor (voxel_postprocessor.py:335-339)
    hb = copy.deepcopy(hd)
    hb["fusion"]["args"]["proj_first"] = True
    hb["noise_setting"] = {"add_noise": True, "args": {"pos_std": 0.3, "rot_std": 0.4, "pos_mean": 0.0, "rot_mean": 0.0, "laplace": True}}
    dsb = IFD(hb, visualize=False, train=False)
    brg = {}
    dsb.scenario = memory_scenario(6, 3, far=False)
    np.random.seed(2024)
    batch = dsb.collate_batch_test([dsb[0]])["ego"]
    for c, (cid, cav) in enumerate(dsb.scenario.items()):
        brg.update({f"lidar{c}": cav["lidar_np"], f"pose{c}": np.array(cav["params"]["lidar_pose"]),
                    f"veh_ids{c}": np.array(list(cav["params"]["vehicles"].keys())),
                    f"veh{c}": np.array([v["location"] + v["angle"] + v["extent"] + v["center"] for v in cav["params"]["vehicles"].values()]).reshape(-1, 12)})
    brg.update({"n_cav": 3, "cav_ids": np.array(list(dsb.scenario.keys())), "np_seed": 2024,
                "voxel_features": batch["processed_lidar"]["voxel_features"], "voxel_coords": batch["processed_lidar"]["voxel_coords"],
                "voxel_num_points": batch["processed_lidar"]["voxel_num_points"], "record_len": batch["record_len"],
                "pairwise_t_matrix": batch["pairwise_t_matrix"], "object_bbx_center": batch["object_bbx_center"],
                "object_bbx_mask": batch["object_bbx_mask"], "object_ids": np.array(batch["object_ids"]),
                "lidar_pose": batch["lidar_pose"], "lidar_pose_clean": batch["lidar_pose_clean"]})
    print(f"  dataset branches: voxels {tuple(batch['processed_lidar']['voxel_features'].shape)}, poses {batch['lidar_pose'][:, [0, 1, 4]].tolist()}")
    # iou_preds: the mini model's head outputs of postprocess.npz + an IoU head, through the reference's post_process
    gp = np.load(os.path.join(HERE, "postprocess.npz"))
    ppr = ref_build_post(load_hypes(YAML_COALIGN, lidar_range=MINI_RANGE)["postprocess"], False)
    gen = torch.Generator().manual_seed(77)
    iou_map = torch.randn(1, 2, gp["i_cls"].shape[2], gp["i_cls"].shape[3], generator=gen)
    anchors_t = torch.from_numpy(gp["anchors"])
    outd = {"ego": {"cls_preds": torch.from_numpy(gp["i_cls"]), "reg_preds": torch.from_numpy(gp["i_reg"]), "dir_preds": torch.from_numpy(gp["i_dir"]), "iou_preds": iou_map}}
    bx, sc = ppr.post_process({"ego": {"transformation_matrix": torch.eye(4), "anchor_box": anchors_t}}, outd)
    brg.update({"iou_preds": iou_map.numpy(), "iou_boxes": bx.numpy(), "iou_scores": sc.numpy()})
    print(f"  iou_preds rescoring: {bx.shape[0]} boxes")
    np.savez_compressed(os.path.join(HERE, "dataset_branches.npz"), **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in brg.items()})
    print(f"wrote dataset_branches.npz: {os.path.getsize(os.path.join(HERE, 'dataset_branches.npz')) // 1024} KiB")

    # ------------------------------------------------------------------ stage-1 detector with uncertainty head + its post-process (next-3)
    from opencood.data_utils.post_processor.uncertainty_voxel_postprocessor import UncertaintyVoxelPostprocessor as RefUncPost
    hu = load_hypes(YAML_UNC, MINI_RANGE)
    mu = train_utils.create_model(hu).eval()
    fill_parameters_(mu, seed=0, cls_bias=-1.0)
    with torch.no_grad():                      # small box deltas / moderate logits so that real detections come out
        mu.reg_head.weight.mul_(0.01); mu.reg_head.bias.zero_(); mu.cls_head.weight.mul_(0.05)
    fu = make_frame(hu, 3, pillars_per_agent=150, seed=21)
    with torch.no_grad():
        ou = mu(fu)
    pu = RefUncPost(hu["postprocess"], train=False)
    anchors_u = torch.from_numpy(pu.generate_anchor_box())
    c_list, b_list, u_list = pu.post_process_stage1({k: v.clone() for k, v in ou.items()}, anchors_u)
    sdu = mu.state_dict()
    st1 = {"voxel_features": fu["processed_lidar"]["voxel_features"], "voxel_coords": fu["processed_lidar"]["voxel_coords"],
           "voxel_num_points": fu["processed_lidar"]["voxel_num_points"], "cls_preds": ou["cls_preds"], "reg_preds": ou["reg_preds"],
           "unc_preds": ou["unc_preds"], "dir_preds": ou["dir_preds"], "state_keys": np.array(list(sdu.keys())),
           "state_numel": np.array([v.numel() for v in sdu.values()]), "n_boxes": np.array([len(c) for c in c_list])}
    for i, (c, b, u) in enumerate(zip(c_list, b_list, u_list)):
        st1.update({f"corners{i}": c, f"boxes{i}": b, f"unc{i}": u})
    print("  stage1 boxes per agent:", [len(c) for c in c_list])
    save("stage1_mini.npz", **st1)

    # ------------------------------------------------------------------ pose correction by box alignment (next-3)
    from opencood.models.sub_modules.box_align_v2 import box_alignment_relative_sample_np

    def align_scene(seed, N, K, pos_noise, rot_noise, detect=0.6, jitter=0.05, yaw_flip=0.0):
        """K vehicles in the world, N agents each detecting a random subset in its own (true) frame with box noise; the
        agents' reported poses carry (pos_noise m, rot_noise deg) errors -- what box alignment is there to correct."""
        rs = np.random.RandomState(seed)
        obj = np.zeros((K, 7)); obj[:, 0] = rs.uniform(-60, 60, K); obj[:, 1] = rs.uniform(-30, 30, K); obj[:, 2] = -1
        obj[:, 3:6] = [4.5, 2.0, 1.6]; obj[:, 6] = rs.uniform(-3.1, 3.1, K)
        pose = np.zeros((N, 6)); pose[:, 0] = rs.uniform(-20, 20, N); pose[:, 1] = rs.uniform(-10, 10, N); pose[:, 4] = rs.uniform(-180, 180, N)
        noisy = pose.copy()
        noisy[1:, 0] += rs.normal(0, pos_noise, N - 1); noisy[1:, 1] += rs.normal(0, pos_noise, N - 1); noisy[1:, 4] += rs.normal(0, rot_noise, N - 1)
        corners, unc = [], []
        for i in range(N):
            Ti = np.linalg.inv(x_to_world(pose[i].tolist()))
            b = obj[rs.uniform(size=K) < detect].copy()
            b[rs.uniform(size=len(b)) < yaw_flip, 6] += np.pi / 2          # heading disagreements -> "yaw varies" clusters
            c = box_utils.boxes_to_corners_3d(torch.from_numpy(b), "lwh").numpy()
            c = (Ti[:3, :3] @ c.transpose(0, 2, 1)).transpose(0, 2, 1) + Ti[:3, 3]
            c[:, :, :2] += rs.normal(0, jitter, (len(c), 1, 2))
            corners.append(c.astype(np.float64)); unc.append(rs.uniform(-3, 0, (len(c), 3)))
        return corners, noisy, unc

    ba = {}
    cases = [("default", dict(seed=1, N=4, K=30, pos_noise=0.3, rot_noise=0.4), dict(use_uncertainty=True, landmark_SE2=True, adaptive_landmark=False, normalize_uncertainty=False, abandon_hard_cases=True, drop_hard_boxes=True)),
             ("five_agents", dict(seed=2, N=5, K=60, pos_noise=0.4, rot_noise=0.8), dict(use_uncertainty=True, landmark_SE2=True, adaptive_landmark=False, normalize_uncertainty=False, abandon_hard_cases=True, drop_hard_boxes=True)),
             ("no_uncertainty", dict(seed=3, N=3, K=40, pos_noise=0.2, rot_noise=0.2), dict(use_uncertainty=False)),
             ("points", dict(seed=4, N=3, K=40, pos_noise=0.2, rot_noise=0.2), dict(landmark_SE2=False, normalize_uncertainty=True)),
             ("adaptive", dict(seed=5, N=4, K=50, pos_noise=0.3, rot_noise=0.3, yaw_flip=0.3), dict(adaptive_landmark=True, drop_unsure_edge=True)),
             ("hard_boxes", dict(seed=6, N=4, K=50, pos_noise=0.3, rot_noise=0.3, yaw_flip=0.15), dict(abandon_hard_cases=True, drop_hard_boxes=True)),
             ("abandoned_few", dict(seed=7, N=2, K=4, pos_noise=0.3, rot_noise=0.3, detect=0.9), dict(abandon_hard_cases=True)),
             ("abandoned_yaw", dict(seed=8, N=3, K=30, pos_noise=0.3, rot_noise=0.3, yaw_flip=0.8), dict(abandon_hard_cases=True)),
             ("empty_agent", dict(seed=9, N=3, K=30, pos_noise=0.3, rot_noise=0.3), dict())]
    for tag, sc, flags in cases:
        corners, noisy, unc = align_scene(**sc)
        if tag == "empty_agent":
            corners[1], unc[1] = np.zeros((0, 8, 3)), np.zeros((0, 3))
        if tag == "adaptive":
            unc = [u - 3.5 for u in unc]                 # certain enough that drop_unsure_edge keeps most edges, not all
        _G2O.recorded.clear()
        refined = box_alignment_relative_sample_np([c.copy() for c in corners], noisy.copy(), uncertainty_list=[u.copy() for u in unc], **flags)
        ba.update({f"{tag}_corners": np.concatenate(corners), f"{tag}_len": np.array([len(c) for c in corners]), f"{tag}_unc": np.concatenate(unc),
                   f"{tag}_noisy": noisy, f"{tag}_refined": refined, f"{tag}_flags": np.array(sorted(f"{k}={v}" for k, v in flags.items())),
                   f"{tag}_solved": len(_G2O.recorded)})
        if _G2O.recorded:
            g = _G2O.recorded[0]
            ba.update({f"{tag}_vertices": g["vertices"], f"{tag}_kinds": g["kinds"], f"{tag}_edge_agent": g["edges"][0], f"{tag}_edge_landmark": g["edges"][1],
                       f"{tag}_edge_meas": g["edges"][2], f"{tag}_edge_info": g["edges"][3], f"{tag}_solution": g["solution"]})
        print(f"  box_align {tag}: solved={len(_G2O.recorded)} moved={np.abs(refined - noisy[:, [0, 1, 4]]).max():.3f}")
    save("box_align.npz", **ba)

    # ------------------------------------------------------------------ full-size pillar path + fusion (samples only)
    hf = load_hypes(YAML_COALIGN)
    mf = train_utils.create_model(hf).eval()
    fill_parameters_(mf, seed=0, cls_bias=-1.0)
    fr = make_frame(hf, 2, pillars_per_agent=8000, seed=303)
    bd = {k: fr["processed_lidar"][k] for k in ("voxel_features", "voxel_coords", "voxel_num_points")}
    bd = mf.scatter(mf.pillar_vfe(bd))
    cv = bd["spatial_features"]
    nz = torch.nonzero(cv.reshape(-1)).view(-1)
    g = torch.Generator().manual_seed(13)
    full = {"pillar_rows": bd["pillar_features"][::50], "canvas_shape": np.array(cv.shape),
            "canvas_nonzero_count": nz.numel(), "canvas_nonzero_sha256": np.frombuffer(bytes.fromhex(sha(nz.numpy().astype(np.int64))), dtype=np.uint8),
            "canvas_sample_idx": nz[::997], "canvas_sample_val": cv.reshape(-1)[nz[::997]],
            "frame_seed": 303, "pillars_per_agent": 8000}
    aff = normalize_pairwise_tfm(fr["pairwise_t_matrix"], 200, 704, 0.4)
    full["affine"] = aff
    for s, (C, H, W) in enumerate(((64, 100, 352), (128, 50, 176), (256, 25, 88))):
        x = torch.randn(2, C, H, W, generator=g)
        y = AttFusion(C)(x, fr["record_len"], aff)
        full[f"att{s}_sample"] = y.reshape(-1)[::211]
        full[f"att{s}_shape"] = np.array(y.shape)
    full["fusion_gen_seed"] = 13
    save("fullsize.npz", **full)

    # ------------------------------------------------------------------ row D: NaiveCompressor (naive_compress.py:5-31)
    # (a) the module alone, randomised BN statistics, ratios 2 and 8; (b) wired into the mini CoAlign model through the
    # yaml's ``compression`` key (point_pillar_baseline_multiscale.py:50-53,113-114), end to end
    from opencood.models.sub_modules.naive_compress import NaiveCompressor
    comp = {}
    gc = torch.Generator().manual_seed(21)
    xin = torch.randn(1, 64, 16, 24, generator=gc)
    comp["x"] = xin
    for ratio in (2, 8):
        nc = NaiveCompressor(64, ratio).eval()
        fill_parameters_(nc, seed=40 + ratio)
        comp[f"y_r{ratio}"] = nc(xin)
        comp[f"keys_r{ratio}"] = np.array(list(nc.state_dict().keys()))
    hc = load_hypes(YAML_COALIGN, MINI_RANGE)
    hc["model"]["args"]["compression"] = 4
    mc = train_utils.create_model(hc).eval()
    fill_parameters_(mc, seed=0, cls_bias=-1.0)
    oc = mc(frame)
    comp.update(cls_preds=oc["cls_preds"], reg_preds=oc["reg_preds"], dir_preds=oc["dir_preds"], model_ratio=4,
                state_keys=np.array(list(mc.state_dict().keys())))
    save("naive_compress.npz", **comp)


if __name__ == "__main__":
    main()
