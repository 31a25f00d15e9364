#!/usr/bin/env python
"""The strided first convolution of a ResNet stage (consumer-split fp16 kernel, channels-last float32 in, SplitMap out): product rule against laboratory
settings, one process per setting (COALIGN_LAB=1), each alone on the GPU (graph replays of 20 launches); prints per-shape times and an output checksum
(settings that only reschedule the same arithmetic must agree).  SETTINGS="name:K=V;name2:K=V" replaces the default list."""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = ((5, 64, 128, 100, 352), (5, 128, 256, 50, 176), (2, 64, 128, 100, 252))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    sys.path.insert(0, ROOT)
    from coalign_amd import ops

    def timed(fn, n=20, reps=4):
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
    for (N, Ci, Co, H, W) in SHAPES:
        g = torch.Generator(device="cuda").manual_seed(H + Co)
        x = torch.relu(torch.randn((N, Ci, H, W), generator=g, device="cuda")).contiguous(memory_format=torch.channels_last)
        w = ops.pack_conv3x3_emu_weight(torch.randn((Co, Ci, 3, 3), generator=g, device="cuda") / (9 * Ci) ** 0.5, 16, False)
        b = torch.randn(Co, generator=g, device="cuda")
        y = ops.conv3x3_emu_bias_act(x, w, b, Co, None, True, 16, stride=2, out_split=True)
        out[f"{N}x{Ci}x{Co}x{H}x{W}"] = {"us": timed(lambda: ops.conv3x3_emu_bias_act(x, w, b, Co, None, True, 16, stride=2, out_split=True)),
                                         "sha": hashlib.sha1(y.data.cpu().numpy().tobytes()).hexdigest()[:12]}
    print(json.dumps(out))
else:
    settings = [("product", {}), ("kch2_one_patch_buffer", {"COALIGN_EMU_S2_KCH2": "1"})]
    if os.environ.get("SETTINGS"):
        settings = [(t.split(":")[0], dict(kv.split("=") for kv in t.split(":")[1].split(",") if kv)) for t in os.environ["SETTINGS"].split(";")]
    for name, env in settings:
        r = subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, COALIGN_LAB="1", **env), capture_output=True, text=True, timeout=300)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        print(name, line[0] if line else r.stderr[-400:])
