#!/usr/bin/env python
"""coalign_conv3x3_bias_act vs MIOpen (F.conv2d) + the separate fused epilogue, at the backbone / shrink-header shapes."""
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
rows = []
for (N, Ci, Co, H, W, res) in ((5, 64, 64, 100, 352, True), (5, 128, 128, 50, 176, True), (5, 256, 256, 25, 88, True),
                               (1, 384, 256, 100, 352, False), (1, 256, 256, 100, 352, False), (2, 64, 64, 37, 52, True)):
    x = torch.randn(N, Ci, H, W, device="cuda"); w = torch.randn(Co, Ci, 3, 3, device="cuda") / (Ci * 9) ** 0.5
    b = torch.randn(Co, device="cuda"); r = torch.randn(N, Co, H, W, device="cuda") if res else None
    wp = ops.pack_conv3x3_weight(w)
    ref = F.conv2d(x, w, b, padding=1)
    if r is not None: ref = ref + r
    ref = torch.relu(ref)
    got = ops.conv3x3_bias_act(x, wp, b, r, True)
    err = float((got - ref).abs().max() / ref.abs().max())
    # fp64 truth: the error of every fp32 path (MIOpen's Winograd included) is measured against it
    ref64 = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    if r is not None: ref64 = ref64 + r.double()
    ref64 = torch.relu(ref64)
    e64 = lambda t: float((t.double() - ref64).abs().max() / ref64.abs().max())
    emu = {}
    for terms in (3, 2):
        ws = ops.pack_conv3x3_emu_weight(w, terms)
        g = ops.conv3x3_emu_bias_act(x, ws, b, Co, r, True, terms)
        emu[terms] = (e64(g), timed(lambda: ops.conv3x3_emu_bias_act(x, ws, b, Co, r, True, terms)))
    t_new = timed(lambda: ops.conv3x3_bias_act(x, wp, b, r, True))
    t_old = timed(lambda: ops.bias_act_(F.conv2d(x, w, None, padding=1), b, r, True))
    gf = 2 * N * Co * H * W * Ci * 9 / 1e9
    rows.append({"shape": [N, Ci, Co, H, W], "rel_err": err, "us_hip": round(t_new, 1), "us_miopen_plus_epilogue": round(t_old, 1),
                 "TFLOPs_hip": round(gf / t_new * 1e3, 1), "TFLOPs_miopen": round(gf / t_old * 1e3, 1),
                 "err_vs_fp64": {"hip_f32": e64(got), "miopen_f32": e64(ref), "emu_bf16x3": emu[3][0], "emu_bf16x2": emu[2][0]},
                 "us_emu_bf16x3": round(emu[3][1], 1), "us_emu_bf16x2": round(emu[2][1], 1),
                 "TFLOPs_emu_bf16x3": round(gf / emu[3][1] * 1e3, 1), "TFLOPs_emu_bf16x2": round(gf / emu[2][1] * 1e3, 1)})
    print(json.dumps(rows[-1]), flush=True)
