#!/usr/bin/env python
"""Engine clock and socket power while one convolution kernel runs back to back (rocm-smi sampled from a side thread):
the fp32-MFMA kernel, MIOpen's Winograd, and the split-bf16 kernels, at the stage-1 shape.  Evidence for DESIGN.md section 8."""
import json, os, re, subprocess, sys, threading, time
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coalign_amd import ops

def sample(stop, out):
    while not stop.is_set():
        try:
            t = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=5).stdout
            d = json.loads(t)
            card = next(iter(d.values()))
            sclk = next((v for k, v in card.items() if "sclk" in k.lower()), None)
            pw = next((v for k, v in card.items() if "power" in k.lower() and "W" in k), None)
            m = re.search(r"(\d+)\s*Mhz", str(sclk), re.I)
            out.append((int(m.group(1)) if m else None, float(pw) if pw not in (None, "N/A") else None))
        except Exception as e:
            out.append(("err", str(e)[:60]))
        time.sleep(0.2)

torch.manual_seed(0)
N, Ci, Co, H, W = 5, 64, 64, 100, 352
x = torch.randn(N, Ci, H, W, device="cuda"); w = torch.randn(Co, Ci, 3, 3, device="cuda") / 24
b = torch.randn(Co, device="cuda"); r = torch.randn(N, Co, H, W, device="cuda")
wp = ops.pack_conv3x3_weight(w); w3 = ops.pack_conv3x3_emu_weight(w, 3); w2 = ops.pack_conv3x3_emu_weight(w, 2)
cases = {"idle": None,
         "fp32_mfma": lambda: ops.conv3x3_bias_act(x, wp, b, r, True),
         "miopen_winograd": lambda: F.conv2d(x, w, b, padding=1),
         "emu_bf16x3": lambda: ops.conv3x3_emu_bias_act(x, w3, b, Co, r, True, 3),
         "emu_bf16x2": lambda: ops.conv3x3_emu_bias_act(x, w2, b, Co, r, True, 2)}
res = {}
torch.backends.cudnn.benchmark = True
for fn in cases.values():          # warm up (MIOpen find, first-launch costs) outside the sampled windows
    if fn is not None:
        for _ in range(5): fn()
torch.cuda.synchronize()
for name, fn in cases.items():
    stop, out = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, out)); th.start()
    t0 = time.time(); n = 0
    while time.time() - t0 < 2.5:
        if fn is None:
            time.sleep(0.05)
        else:
            for _ in range(50): fn()
            torch.cuda.synchronize(); n += 50
    dt = time.time() - t0
    stop.set(); th.join()
    clk = [c for c, _ in out if isinstance(c, int)]; pw = [p for _, p in out if isinstance(p, float)]
    res[name] = {"us_per_call": round(dt / n * 1e6, 1) if n else None, "sclk_MHz": clk[1:], "power_W": [round(p) for p in pw[1:]], "raw": out[:1] if not clk else None}
    res[name].pop("raw")
    print(json.dumps({name: res[name]}), flush=True)
