#!/usr/bin/env python
"""Producer-split convolution (csrc/conv3x3_sp.hip) against the consumer-split fp16 kernel (csrc/conv3x3_emu.hip), per backbone layer shape, each alone on
the GPU: HIP-graph replays of 20 launches timed with events.  Prints one JSON line per shape and a frame-weighted sum; GEO=81,121,124,148 adds fixed geometries."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from coalign_amd import ops  # noqa: E402

SHAPES = ((5, 64, 64, 100, 352), (5, 128, 128, 50, 176), (5, 256, 256, 25, 88), (1, 256, 256, 100, 352), (2, 64, 64, 100, 252), (2, 256, 256, 25, 63))
if os.environ.get("SHAPES"):          # e.g. SHAPES="2x64x64x100x352,2x128x128x50x176": other shapes (frame weights zero)
    SHAPES = tuple(tuple(int(v) for v in item.split("x")) for item in os.environ["SHAPES"].split(","))
WEIGHT = (5, 9, 15, 1, 0, 0) if not os.environ.get("SHAPES") else (0,) * len(SHAPES)            # stride-1 SplitMap-input 3x3 layers of each shape in one 5-agent OPV2V frame (layer_nums 3 / 5 / 8 + the shrink header's second)


def timed(fn, n=20, reps=5):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        g.replay()
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / n * 1e3)
    return round(best, 2)


def main():
    geos = [0] + [int(v) for v in os.environ.get("GEO", "").split(",") if v]
    total = {}
    for (N, Ci, Co, H, W), wt in zip(SHAPES, WEIGHT):
        g = torch.Generator(device="cuda").manual_seed(N + Ci + H)
        x = torch.relu(torch.randn((N, Ci, H, W), generator=g, device="cuda"))
        w = torch.randn((Co, Ci, 3, 3), generator=g, device="cuda") / (Ci * 9) ** 0.5
        b = torch.randn(Co, generator=g, device="cuda")
        r = torch.randn((N, Co, H, W), generator=g, device="cuda")
        w16 = ops.pack_conv3x3_emu_weight(w, 16, True)
        xs, rs = ops.SplitMap.pack(x), ops.SplitMap.pack(r)
        row = {"shape": [N, Ci, Co, H, W], "gflop_x3": round(3 * 2 * 9 * Ci * Co * H * W * N / 1e9, 2)}
        row["consumer_split_us"] = timed(lambda: ops.conv3x3_emu_bias_act(x, w16, b, Co, r, True, 16))
        for geo in geos:
            try:
                row[f"sp_geo{geo}_us"] = timed(lambda: ops.conv3x3_sp(xs, w16, b, Co, rs, True, out_split=True, geometry=geo))
                row[f"sp_geo{geo}_nhwc_out_us"] = timed(lambda: ops.conv3x3_sp(xs, w16, b, Co, rs, True, out_split=False, geometry=geo))
            except Exception as exc:  # noqa: BLE001
                row[f"sp_geo{geo}_us"] = f"fail: {exc}"
        best = min(v for k, v in row.items() if k.startswith("sp_geo") and k.endswith("_us") and not k.endswith("out_us") and isinstance(v, float))
        row["frac_of_fp16_peak"] = round(row["gflop_x3"] * 1e9 / (best * 1e-6) / 2.5e15, 3)
        print(json.dumps(row), flush=True)
        for k, v in row.items():
            if k.endswith("_us") and isinstance(v, float):
                total[k] = round(total.get(k, 0.0) + wt * v, 1)
    print(json.dumps({"frame_weighted_us": total}))


if __name__ == "__main__":
    main()
