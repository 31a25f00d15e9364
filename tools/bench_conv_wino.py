#!/usr/bin/env python
"""Winograd F(2x2, 3x3) split-bf16 convolution (csrc/conv3x3_wino.hip) against the direct tap-major split-bf16 kernel on the five stride-1 3x3
shapes of the OPV2V frame: microseconds per layer (HIP events, 20 launches), error against the float64 convolution, and the per-frame sum
weighted by how often each shape occurs (5 / 9 / 15 / 1 / 1).  `--abl`: the laboratory build's ablations of the Winograd kernel
(COALIGN_WINO_ABL: 1 no matrix steps, 2 no transform / split, 3 both, 4 operands of half stage 0 only, 7 all) -- what each part costs.
Output: gpurun_out/wino/conv_wino.json."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SHAPES = ((5, 64, 64, 100, 352), (5, 128, 128, 50, 176), (5, 256, 256, 25, 88), (1, 384, 256, 100, 352), (1, 256, 256, 100, 352))
WEIGHT = (5, 9, 15, 1, 1)


def child(check):
    import torch
    import torch.nn.functional as F
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
    tbws = [int(v) for v in os.environ.get("TBWS", "0,8,16").split(",")]
    for (N, Ci, Co, H, W) in SHAPES:
        g = torch.Generator().manual_seed(N + Ci + H)
        x = torch.randn(N, Ci, H, W, generator=g).cuda(); w = (torch.randn(Co, Ci, 3, 3, generator=g) / (Ci * 9) ** 0.5).cuda()
        b = torch.randn(Co, generator=g).cuda(); r = torch.randn(N, Co, H, W, generator=g).cuda()
        xl, rl = x.contiguous(memory_format=torch.channels_last), r.contiguous(memory_format=torch.channels_last)
        wu = ops.pack_conv3x3_wino_weight(w)
        row = {}
        if check:
            wd = ops.pack_conv3x3_emu_weight(w, 3, True)
            want = torch.relu(F.conv2d(x.double(), w.double(), b.double(), padding=1) + r.double())
            sc = float(want.abs().max())
            d = ops.conv3x3_emu_bias_act(x, wd, b, Co, r, True, 3)
            row["direct_err"] = float(f"{float((d.double() - want).abs().max()) / sc:.2e}")
            row["direct_us"] = round(timed(lambda: ops.conv3x3_emu_bias_act(x, wd, b, Co, r, True, 3)), 1)
        for tbw in tbws:
            try:
                y = ops.conv3x3_wino(xl, wu, b, Co, rl, True, tile_block_w=tbw)
                if check:
                    row[f"wino{tbw}_err"] = float(f"{float((y.double() - want).abs().max()) / sc:.2e}")
                row[f"wino{tbw}_us"] = round(timed(lambda: ops.conv3x3_wino(xl, wu, b, Co, rl, True, tile_block_w=tbw)), 1)
            except Exception as e:      # noqa: BLE001
                row[f"wino{tbw}"] = f"fail: {e}"
        out["x".join(map(str, (N, Ci, Co, H, W)))] = row
    print(json.dumps(out))


if len(sys.argv) > 1 and sys.argv[1] == "child":
    child(os.environ.get("CHECK", "1") == "1")
    sys.exit(0)

def run(env):
    r = subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, **env), capture_output=True, text=True, timeout=280)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return json.loads(line[0]) if line else {"fail": r.stderr[-400:]}

out = run({})
for k, v in out.items():
    print(k, v, flush=True)
keys = sorted({k for v in out.values() if isinstance(v, dict) for k in v if k.endswith("_us")})
tot = {k: round(sum(wt * out["x".join(map(str, s))].get(k, float("nan")) for s, wt in zip(SHAPES, WEIGHT))) for k in keys}
print("weighted us per frame:", tot)
out["weighted_us_per_frame"] = tot
if "--abl" in sys.argv:
    out["ablations_wino16_us"] = {}
    for abl in (1, 2, 3, 4, 7):
        r = run({"COALIGN_LAB": "1", "COALIGN_WINO_ABL": str(abl), "CHECK": "0", "TBWS": "16"})
        out["ablations_wino16_us"][abl] = {k: (v.get("wino16_us", v) if isinstance(v, dict) else v) for k, v in r.items()}
        print("ablation", abl, out["ablations_wino16_us"][abl], flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out", "wino"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "wino", "conv_wino.json"), "w"), indent=1)
