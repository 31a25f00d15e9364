#!/usr/bin/env python
"""Time coalign_voxelize on a frame of synthetic 64-beam sweeps (inputs resident in HBM).
python tools/bench_voxelize.py [--clouds 5] [--iters 50] [--shuffle]   (host-side reference timing: tests/cpu_reference_timings.py)"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coalign_amd import ops                                   # noqa: E402
from coalign_amd.synthetic import make_point_cloud            # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--clouds", type=int, default=5)
ap.add_argument("--iters", type=int, default=50)
ap.add_argument("--shuffle", action="store_true")
a = ap.parse_args()
rs = np.random.RandomState(0)
clouds = [make_point_cloud(40 + i) for i in range(a.clouds)]
if a.shuffle:
    clouds = [c[rs.permutation(len(c))] for c in clouds]
off = np.concatenate([[0], np.cumsum([len(c) for c in clouds])]).tolist()
pts = torch.from_numpy(np.concatenate(clouds)).cuda()
RANGE, VOXEL = [-140.8, -40, -3, 140.8, 40, 1], [0.4, 0.4, 4]
for _ in range(5):
    out = ops.voxelize(pts, off, VOXEL, RANGE, 32, 70000, ego_filter=True)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(a.iters):
    out = ops.voxelize(pts, off, VOXEL, RANGE, 32, 70000, ego_filter=True)
e.record()
torch.cuda.synchronize()
us = s.elapsed_time(e) * 1e3 / a.iters
m = int(out[3][-1])
alg = pts.numel() * 4 + m * (32 * 16 + 16 + 4)
rep = {"clouds": a.clouds, "points": int(pts.shape[0]), "voxels": m, "us_per_call": round(us, 2), "algorithmic_bytes": alg,
       "GBps": round(alg / us / 1e3, 1), "shuffled": a.shuffle}
print(json.dumps(rep))
