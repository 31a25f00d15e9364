#!/usr/bin/env python
"""Tile-geometry sweep of the split-bf16 convolution: one process per geometry (COALIGN_EMU_GEO), all backbone shapes."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    sys.path.insert(0, ROOT)
    from coalign_amd import ops
    def timed(fn, n=20, warm=5):
        for _ in range(warm): fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n): fn()
        e.record(); torch.cuda.synchronize()
        return s.elapsed_time(e) / n * 1e3
    out = {}
    for (N, Ci, Co, H, W) in ((5, 64, 64, 100, 352), (5, 128, 128, 50, 176), (5, 256, 256, 25, 88), (1, 384, 256, 100, 352), (1, 256, 256, 100, 352)):
        x = torch.randn(N, Ci, H, W, device="cuda"); w = torch.randn(Co, Ci, 3, 3, device="cuda") / (Ci * 9) ** 0.5
        b = torch.randn(Co, device="cuda"); r = torch.randn(N, Co, H, W, device="cuda")
        for terms in [int(v) for v in os.environ.get("TERMS", "2,3").split(",")]:
            ws = ops.pack_conv3x3_emu_weight(w, terms)
            try:
                out[f"{N}x{Ci}x{Co}x{H}x{W} x{terms}"] = round(timed(lambda: ops.conv3x3_emu_bias_act(x, ws, b, Co, r, True, terms)), 1)
            except Exception as e:
                out[f"{N}x{Ci}x{Co}x{H}x{W} x{terms}"] = "fail"
    print(json.dumps(out))
else:
    rows = {}
    for geo in [int(v) for v in os.environ.get('GEOS', '81,82,121,122').split(',')]:
        env = dict(os.environ, COALIGN_EMU_GEO=str(geo))
        r = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True, timeout=200)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        rows[geo] = json.loads(line[0]) if line else r.stderr[-300:]
    names = {g: f"{g // 10}w/k{8 * (g % 10)}" for g in rows}
    for k in next(r for r in rows.values() if isinstance(r, dict)):
        print(k, {names[g]: (rows[g][k] if isinstance(rows[g], dict) else "fail") for g in rows})
