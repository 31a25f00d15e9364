#!/usr/bin/env python
"""Probe (round 6): does ONE forward over a batch of TWO frames (B = 2: 10 agents through the encoder, two fusion groups, heads on [2, ...]) cost less per frame than two
forwards of one frame each?  HIP-graph replays of model.forward (no post-processing), one stream and two streams side by side.
Measured (round 6, ms per frame): B=1 one stream 1.894, B=1 two streams 1.520, B=2 one stream 1.846, B=2 two streams 1.821 -- two frames side by side on two streams
(what FramePipeline does) beat one launch over two frames; batching is not pursued."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from coalign_amd.config import builtin_config
from coalign_amd.detector import build_model, to_device
from coalign_amd.synthetic import fill_parameters_, make_frame

dev = torch.device("cuda:0")
h = builtin_config("opv2v_coalign")
model = build_model(h); fill_parameters_(model, seed=0); model = model.to(dev).eval()
N = 5
frames = [to_device(make_frame(h, N, pillars_per_agent=8000, seed=303 + i, noise=(0.2, 0.2)), dev) for i in range(4)]


def batch_of(fs):
    pls = [f["processed_lidar"] for f in fs]
    coords = []
    for k, pl in enumerate(pls):
        c = pl["voxel_coords"].clone(); c[:, 0] += k * N; coords.append(c)
    return {"processed_lidar": {"voxel_features": torch.cat([p["voxel_features"] for p in pls]), "voxel_coords": torch.cat(coords),
                                "voxel_num_points": torch.cat([p["voxel_num_points"] for p in pls])},
            "record_len": [N] * len(fs), "pairwise_t_matrix": torch.cat([f["pairwise_t_matrix"] for f in fs])}


def graph_of(batch, stream):
    with torch.no_grad(), torch.cuda.stream(stream):
        for _ in range(2):
            out = model(batch)
        stream.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            out = model(batch)
    return g, out


def timed(graphs, reps=30):
    for g, s in graphs:
        with torch.cuda.stream(s):
            g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        for g, s in graphs:
            with torch.cuda.stream(s):
                g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for f in frames:
    f["record_len"] = [N]
b1a, b1b = frames[0], frames[1]
b2a, b2b = batch_of(frames[0:2]), batch_of(frames[2:4])
g1a, o1 = graph_of(b1a, s1); g1b, _ = graph_of(b1b, s2)
g2a, o2 = graph_of(b2a, s1); g2b, _ = graph_of(b2b, s2)
with torch.no_grad():
    ref0 = model(frames[0])
with torch.cuda.stream(s1):
    g2a.replay()                                    # (a captured graph's output buffers hold nothing until its first replay)
torch.cuda.synchronize()
print("B=2 output frame 0 equals the B=1 output:", {k: bool(torch.equal(o2[k][0:1], ref0[k])) for k in ref0}, {k: float((o2[k][0:1] - ref0[k]).abs().max()) for k in ref0})
t11 = timed([(g1a, s1)]); t12 = timed([(g1a, s1), (g1b, s2)])
t21 = timed([(g2a, s1)]); t22 = timed([(g2a, s1), (g2b, s2)])
print(f"forward only (no decode / NMS), ms per FRAME: B=1 one stream {t11:.3f}; B=1 two streams {t12 / 2:.3f}; B=2 one stream {t21 / 2:.3f}; B=2 two streams {t22 / 4:.3f}")
