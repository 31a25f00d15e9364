#!/usr/bin/env python
"""coalign_pointwise_conv vs the MIOpen route (conv / conv_transpose + fused epilogue) at the backbone's pointwise shapes."""
import json, os, sys
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coalign_amd import ops

def timed(fn, n=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3

torch.manual_seed(0)
for (N, Ci, Co, H, W, up, st) in ((1, 64, 128, 100, 352, 1, 1), (1, 128, 128, 50, 176, 2, 1), (1, 256, 128, 25, 88, 4, 1),
                                  (5, 64, 64, 200, 704, 1, 2), (5, 64, 128, 100, 352, 1, 2), (5, 128, 256, 50, 176, 1, 2), (1, 256, 20, 100, 352, 1, 1)):
    x = torch.randn(N, Ci, H, W, device="cuda"); b = torch.randn(Co, device="cuda")
    if up > 1 or (st == 1 and Co == 128):
        w = torch.randn(Ci, Co, up, up, device="cuda") / Ci ** 0.5
        ref = torch.relu(F.conv_transpose2d(x, w, b, stride=up))
        wp = ops.pack_pointwise_weight(w, True)
        old = lambda: ops.bias_act_(F.conv_transpose2d(x, w, None, stride=up).contiguous(), b, None, True)
    else:
        w = torch.randn(Co, Ci, 1, 1, device="cuda") / Ci ** 0.5
        ref = torch.relu(F.conv2d(x, w, b, stride=st))
        wp = ops.pack_pointwise_weight(w, False)
        old = lambda: ops.bias_act_(F.conv2d(x, w, None, stride=st), b, None, True)
    got = ops.pointwise_conv(x, wp, b, Co, up=up, in_stride=st, relu=True)
    err = float((got - ref).abs().max() / ref.abs().max())
    t_new, t_old = timed(lambda: ops.pointwise_conv(x, wp, b, Co, up=up, in_stride=st, relu=True)), timed(old)
    print(json.dumps({"shape": [N, Ci, Co, H, W], "up": up, "in_stride": st, "rel_err": err, "us_hip": round(t_new, 1), "us_miopen_plus_epilogue": round(t_old, 1)}), flush=True)
