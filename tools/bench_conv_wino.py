#!/usr/bin/env python
"""Winograd F(2x2, 3x3) split-bf16 convolution (csrc/conv3x3_wino.hip) against the direct tap-major split-bf16 kernel on the five stride-1 3x3
shapes of the OPV2V frame: microseconds per layer (HIP events, 20 launches), error against the float64 convolution, and the per-frame sum
weighted by how often each shape occurs (5 / 9 / 15 / 1 / 1).  Output: gpurun_out/wino/conv_wino.json."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from coalign_amd import ops

SHAPES = ((5, 64, 64, 100, 352), (5, 128, 128, 50, 176), (5, 256, 256, 25, 88), (1, 384, 256, 100, 352), (1, 256, 256, 100, 352))
WEIGHT = (5, 9, 15, 1, 1)


def timed(fn, n=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


out = {}
for (N, Ci, Co, H, W) in SHAPES:
    g = torch.Generator().manual_seed(N + Ci + H)
    x = torch.randn(N, Ci, H, W, generator=g).cuda(); w = (torch.randn(Co, Ci, 3, 3, generator=g) / (Ci * 9) ** 0.5).cuda()
    b = torch.randn(Co, generator=g).cuda(); r = torch.randn(N, Co, H, W, generator=g).cuda()
    xl, rl = x.contiguous(memory_format=torch.channels_last), r.contiguous(memory_format=torch.channels_last)
    wd, wu = ops.pack_conv3x3_emu_weight(w, 3, True), ops.pack_conv3x3_wino_weight(w)
    want = torch.relu(F.conv2d(x.double(), w.double(), b.double(), padding=1) + r.double())
    sc = float(want.abs().max())
    key = f"{N}x{Ci}x{Co}x{H}x{W}"
    row = {}
    d = ops.conv3x3_emu_bias_act(x, wd, b, Co, r, True, 3)
    row["direct_err"] = float(f"{float((d.double() - want).abs().max()) / sc:.2e}")
    row["direct_us"] = round(timed(lambda: ops.conv3x3_emu_bias_act(x, wd, b, Co, r, True, 3)), 1)
    for tbw in (0, 8, 16):
        try:
            y = ops.conv3x3_wino(xl, wu, b, Co, rl, True, tile_block_w=tbw)
            row[f"wino{tbw}_err"] = float(f"{float((y.double() - want).abs().max()) / sc:.2e}")
            row[f"wino{tbw}_us"] = round(timed(lambda: ops.conv3x3_wino(xl, wu, b, Co, rl, True, tile_block_w=tbw)), 1)
        except Exception as e:      # noqa: BLE001
            row[f"wino{tbw}"] = f"fail: {e}"
    out[key] = row
    print(key, row, flush=True)
tot = {k: sum(wt * out["x".join(map(str, s))].get(k, float("nan")) for s, wt in zip(SHAPES, WEIGHT)) for k in ("direct_us", "wino0_us", "wino8_us", "wino16_us")}
print("weighted us per frame:", {k: round(v) for k, v in tot.items()})
out["weighted_us_per_frame"] = tot
os.makedirs(os.path.join(ROOT, "gpurun_out", "wino"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "wino", "conv_wino.json"), "w"), indent=1)
