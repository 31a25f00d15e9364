#!/usr/bin/env python
"""Occupancy of the GPU's timeline from a rocprofv3 --kernel-trace CSV: over the busiest window of the run (the timed steps of bench.py), the share of wall time
with >= 1 / >= 2 kernels in flight, the kernel time per frame and per kernel family.  usage: timeline_busy.py <kernel_trace.csv> <steps>"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2])
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows]
ev.sort()
# the timed region: the last `steps` pillar launches (one per frame) and what follows them
pill = [e for e in ev if "pillar_sparse_kernel" in e[2]]
import bisect
starts = [e[0] for e in ev]
def n_kernels(i):
    return bisect.bisect_left(starts, pill[i + steps][0]) - bisect.bisect_left(starts, pill[i][0])
cand = [i for i in range(len(pill) - steps) if n_kernels(i) >= 45 * steps]      # whole frames (>= 45 launches each), not the isolated-kernel timing loops
best = min(cand, key=lambda i: pill[i + steps][0] - pill[i][0])                # the densest such run = the timed region
t0, t1 = pill[best][0], pill[best + steps][0]
win = [(max(s, t0), min(e, t1), n) for s, e, n in ev if e > t0 and s < t1]
pts = []
for s, e, _ in win:
    pts.append((s, 1)); pts.append((e, -1))
pts.sort()
busy1 = busy2 = 0
depth, last = 0, t0
for t, d in pts:
    if depth >= 1: busy1 += t - last
    if depth >= 2: busy2 += t - last
    depth += d; last = t
tot = t1 - t0
fam = defaultdict(lambda: [0, 0])
for s, e, n in win:
    k = n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:70]
    fam[k][0] += 1; fam[k][1] += e - s
print(f"window {tot / 1e6:.2f} ms for {steps} frames = {tot / steps / 1e3:.1f} us/frame; >=1 kernel in flight {busy1 / tot:.3f}, >=2 {busy2 / tot:.3f}; kernel time per frame {sum(v[1] for v in fam.values()) / steps / 1e3:.0f} us")
for k, v in sorted(fam.items(), key=lambda kv: -kv[1][1])[:24]:
    print(f"  {v[0] / steps:6.2f} x {v[1] / v[0] / 1e3:7.1f} us = {v[1] / steps / 1e3:7.1f} us/frame  {k}")
