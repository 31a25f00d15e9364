#!/usr/bin/env python
"""The three strided first convolutions of the 5-agent OPV2V backbone, each alone on the GPU (graph replays of 20 launches, HIP events): the consumer-split
kernel of rounds 4-5 (csrc/conv3x3_emu.hip: float32 channels-last / sparse canvas in) against round 6's coalign_conv3x3_sp_s2 (SplitMap / sp16 rows in), plus what
the new route adds in front (coalign_sp_pack / coalign_sp_pack_rows).  Prints one JSON line."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from coalign_amd import ops  # noqa: E402


def timed(fn, n=20, reps=5):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    best = 1e9
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); g.replay(); e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / n * 1e3)
    return round(best, 2)


out = {}
for (N, Ci, Co, H, W) in ((5, 64, 128, 100, 352), (5, 128, 256, 50, 176), (2, 64, 128, 100, 252)):
    g = torch.Generator(device="cuda").manual_seed(H + Co)
    x = torch.relu(torch.randn((N, Ci, H, W), generator=g, device="cuda")).contiguous(memory_format=torch.channels_last)
    wt = torch.randn((Co, Ci, 3, 3), generator=g, device="cuda") / (9 * Ci) ** 0.5
    w_pair, w_tap = ops.pack_conv3x3_emu_weight(wt, 16, False), ops.pack_conv3x3_emu_weight(wt, 16, True)
    b = torch.randn(Co, generator=g, device="cuda")
    xs = ops.SplitMap.pack(x)
    key = f"{N}x{Ci}x{Co}x{H}x{W}"
    wd = torch.randn((Co, Ci, 1, 1), generator=g, device="cuda") / Ci ** 0.5
    w_skip = ops.pack_conv1x1_sp_weight(wd)
    from coalign_amd.backbone import PointwisePack
    pw, zb = PointwisePack(wd, False).get(), torch.zeros(Co, device="cuda")
    out[key] = {"emu_us": timed(lambda: ops.conv3x3_emu_bias_act(x, w_pair, b, Co, None, True, 16, stride=2, out_split=True)),
                "sp_s2_us": timed(lambda: ops.conv3x3_sp_s2(xs, w_tap, b, Co, True)),
                "sp_s2_with_skip_tap_us": timed(lambda: ops.conv3x3_sp_s2(xs, w_tap, b, Co, True, w_skip=w_skip)),
                "skip_pointwise_launch_us": timed(lambda: ops.pointwise_conv(x, pw, zb, Co, in_stride=2, relu=False, out_channels_last=True)),
                "sp_pack_us": timed(lambda: ops.SplitMap.pack(x))}
from test_s2_gpu import _sparse_canvas  # noqa: E402
for pillars in (8000,):
    sc = _sparse_canvas(5, 200, 704, pillars, 11)
    g = torch.Generator(device="cuda").manual_seed(1)
    wt = torch.randn((64, 64, 3, 3), generator=g, device="cuda") / 24.0
    w_pair, w_tap = ops.pack_conv3x3_emu_weight(wt, 16, False), ops.pack_conv3x3_emu_weight(wt, 16, True)
    b = torch.randn(64, generator=g, device="cuda")
    rows = ops.sp_pack_rows(sc)
    L = ops.hip.lib()
    y = ops.SplitMap.empty(5, 64, 100, 352, "cuda")

    def s2_only():
        ops.hip.check(L.coalign_conv3x3_sp_s2_sparse(rows.data_ptr(), rows.shape[0], sc.stamps.data_ptr(), sc.state.data_ptr(), w_tap.data_ptr(), b.data_ptr(), y.data.data_ptr(),
                                                     5, 64, 64, 200, 704, 1, None, ops._stream()), "s2")
    out[f"sparse_5x{pillars}"] = {"emu_us": timed(lambda: ops.conv3x3_emu_sparse(sc, w_pair, b, 64, True, 16, False, out_split=True)),
                                 "sp_s2_us": timed(s2_only), "pack_rows_us": timed(lambda: ops.sp_pack_rows(sc)),
                                 "sp_s2_with_pack_us": timed(lambda: ops.conv3x3_sp_s2(sc, w_tap, b, 64, True))}
print(json.dumps(out))
# the three up-sampling heads on ONE fused map: a launch per scale against one launch (round 6)
from coalign_amd.backbone import PointwisePack  # noqa: E402
g = torch.Generator(device="cuda").manual_seed(3)
layers, c_off = [], 0
for cin, up in ((64, 1), (128, 2), (256, 4)):
    x = torch.relu(torch.randn((1, cin, 100 // up, 352 // up), generator=g, device="cuda")).contiguous(memory_format=torch.channels_last)
    wt = torch.randn((cin, 128, up, up), generator=g, device="cuda") / cin ** 0.5
    layers.append((x, PointwisePack(wt, True).get(), torch.randn(128, generator=g, device="cuda"), 128, up, c_off))
    c_off += 128
cat = ops.SplitMap.empty(1, c_off, 100, 352, "cuda")


def per_scale():
    for (x, im, b, cout, up, off) in layers:
        ops.pointwise_conv(x, im, b, cout, up=up, relu=True, out=cat, c_off=off)
print(json.dumps({"heads_three_launches_us": timed(per_scale, n=10), "heads_one_launch_us": timed(lambda: ops.pointwise_heads_split(layers, cat), n=10)}))
# the merged 1 x 1 heads on the shrink header's map: pointwise kernel on channels-last float32 against coalign_heads_sp on the SplitMap (round 6)
g = torch.Generator(device="cuda").manual_seed(9)
xm = torch.relu(torch.randn((1, 256, 100, 352), generator=g, device="cuda"))
wh = torch.randn((20, 256, 1, 1), generator=g, device="cuda") / 16.0
bh = torch.randn(20, generator=g, device="cuda")
xcl, xsp = xm.contiguous(memory_format=torch.channels_last), ops.SplitMap.pack(xm)
pk, img = PointwisePack(wh, False).get(), ops.pack_heads_sp_weight(wh)
print(json.dumps({"heads_1x1_pointwise_us": timed(lambda: ops.pointwise_conv(xcl, pk, bh, 20, relu=False), n=10), "heads_1x1_on_split_map_us": timed(lambda: ops.heads_sp(xsp, img, bh, 20), n=10)}))
