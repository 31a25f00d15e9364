#!/usr/bin/env python
"""fp16 2-way split convolution (the product default), tap-major image: per-layer time of laboratory settings against the product rule, one process per
setting (COALIGN_LAB=1), every backbone shape; each child checks its result against the fp64 convolution and prints a checksum (settings that only
reschedule the same arithmetic must agree bit for bit).  SETTINGS="name:K=V,K=V;name2:..." replaces the default list."""
import json, os, subprocess, sys, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = ((5, 64, 64, 100, 352), (5, 128, 128, 50, 176), (5, 256, 256, 25, 88), (1, 384, 256, 100, 352), (1, 256, 256, 100, 352))
WEIGHT = (5, 7, 9, 1, 1)            # stride-1 3x3 layers of each shape in one OPV2V frame
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    import torch.nn.functional as F
    sys.path.insert(0, ROOT)
    from coalign_amd import ops
    def timed(fn, n=30, warm=5):
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
        ws = ops.pack_conv3x3_emu_weight(w, 16, True)
        key = f"{N}x{Ci}x{Co}x{H}x{W}"
        try:
            got = ops.conv3x3_emu_bias_act(x, ws, b, Co, r, True, 16)
            want = torch.relu(F.conv2d(x.double(), w.double(), b.double(), padding=1) + r.double())
            err = float((got.double() - want).abs().max() / want.abs().max())
            gotcl = ops.conv3x3_emu_bias_act(x, ws, b, Co, r, True, 16, out_channels_last=True)
            same = bool(torch.equal(gotcl, got))
            out[key] = {"us": round(timed(lambda: ops.conv3x3_emu_bias_act(x, ws, b, Co, r, True, 16)), 1),
                        "us_nhwc_out": round(timed(lambda: ops.conv3x3_emu_bias_act(x, ws, b, Co, r, True, 16, out_channels_last=True)), 1),
                        "err": float(f"{err:.2e}"), "nhwc_equal": same, "sha": hashlib.sha1(got.cpu().numpy().tobytes()).hexdigest()[:12]}
        except Exception as e:
            out[key] = f"fail: {e}"
    print(json.dumps(out))
else:
    settings = [("product", {}), ("per_image_tiles", {"COALIGN_EMU_STACK16": "0"}), ("stack_50_only", {"COALIGN_EMU_STACK16": "1"}), ("prio1", {"COALIGN_EMU_PRIO": "1"})]
    if os.environ.get("SETTINGS"):
        settings = [(t.split(":")[0], dict(kv.split("=") for kv in t.split(":")[1].split(",") if kv)) for t in os.environ["SETTINGS"].split(";")]
    rows = {}
    for name, env in settings:
        r = subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, COALIGN_LAB="1", **env), capture_output=True, text=True, timeout=300)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        rows[name] = json.loads(line[0]) if line else r.stderr[-400:]
    keys = ["x".join(map(str, s)) for s in SHAPES]
    for k in keys:
        print(k, {n: (rows[n].get(k) if isinstance(rows[n], dict) else rows[n]) for n in rows})
    score = {n: round(sum(wt * r[k]["us"] for k, wt in zip(keys, WEIGHT))) for n, r in rows.items() if isinstance(r, dict) and all(isinstance(v, dict) for v in r.values())}
    print("weighted us per frame:", score)
    os.makedirs(os.path.join(ROOT, "gpurun_out", "fp16"), exist_ok=True)
    json.dump({"rows": rows, "weighted_us_per_frame": score}, open(os.path.join(ROOT, "gpurun_out", "fp16", "conv_fp16.json"), "w"), indent=1)
