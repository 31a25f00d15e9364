#!/usr/bin/env python
"""Test infrastructure: wall-clock of the CPU oracle on the workloads of tools/bench_voxelize.py and tools/bench_box_align.py
(the host-side figures quoted in DESIGN.md).  Lives under tests/ because only tests/, smoke() and bench.py's cpu_baseline leg
may execute oracle/.      python tests/cpu_reference_timings.py"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import coalign_oracle as oracle                    # noqa: E402
from coalign_amd.synthetic import make_point_cloud             # noqa: E402

RANGE, VOXEL = [-140.8, -40, -3, 140.8, 40, 1], [0.4, 0.4, 4]
clouds = [make_point_cloud(40 + i) for i in range(5)]
oracle.points_to_voxel(oracle.mask_ego_points(clouds[0]), VOXEL, RANGE, 32, 70000)
t = time.perf_counter()
for c in clouds:
    oracle.points_to_voxel(oracle.mask_ego_points(c), VOXEL, RANGE, 32, 70000)
rep = {"voxelise_5_sweeps_us": round((time.perf_counter() - t) * 1e6, 1), "points": int(sum(len(c) for c in clouds))}
g = np.load(os.path.join(os.path.dirname(__file__), "golden", "box_align.npz"), allow_pickle=True)
t, n = time.perf_counter(), 0
for tag in ("default", "five_agents", "hard_boxes", "no_uncertainty") * 2:
    oracle.pose_graph_lm(g[f"{tag}_vertices"], g[f"{tag}_kinds"], (g[f"{tag}_edge_agent"], g[f"{tag}_edge_landmark"], g[f"{tag}_edge_meas"], g[f"{tag}_edge_info"]))
    n += 1
rep["pose_graph_lm_ms_per_graph"] = round((time.perf_counter() - t) * 1e3 / n, 2)
print(json.dumps(rep))
