#!/usr/bin/env python
"""Engine clock and socket power while ONE layer shape runs back to back on the split-bf16 convolution (rocm-smi sampled from a side thread); the kernel
variant is whatever COALIGN_EMU_TAPK_ROWS / COALIGN_EMU_STACK select.  SHAPE=N,C,H,W."""
import json, os, re, subprocess, sys, threading, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coalign_amd import ops

def sample(stop, out):
    while not stop.is_set():
        try:
            d = json.loads(subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=5).stdout)
            card = next(iter(d.values()))
            sclk = next((v for k, v in card.items() if "sclk" in k.lower()), None)
            pw = next((v for k, v in card.items() if "power" in k.lower() and "W" in k), None)
            m = re.search(r"(\d+)\s*Mhz", str(sclk), re.I)
            out.append((int(m.group(1)) if m else None, float(pw) if pw not in (None, "N/A") else None))
        except Exception as e:
            out.append(("err", str(e)[:60]))
        time.sleep(0.2)

N, C, H, W = [int(v) for v in os.environ.get("SHAPE", "5,256,25,88").split(",")]
g = torch.Generator().manual_seed(0)
STRIDE = int(os.environ.get("STRIDE", "1"))          # 2: the strided layers (tap-pair weight image, no residual), H, W = INPUT size
x = torch.randn(N, C, H, W, generator=g).cuda(); w = ops.pack_conv3x3_emu_weight((torch.randn(C, C, 3, 3, generator=g) / (C * 9) ** 0.5).cuda(), 3, STRIDE == 1)
b = torch.randn(C, generator=g).cuda(); r = torch.randn(N, C, H, W, generator=g).cuda()
fn = (lambda: ops.conv3x3_emu_bias_act(x, w, b, C, r, True, 3)) if STRIDE == 1 else (lambda: ops.conv3x3_emu_bias_act(x, w, b, C, None, True, 3, stride=2))
for _ in range(5): fn()
torch.cuda.synchronize()
stop, out = threading.Event(), []
th = threading.Thread(target=sample, args=(stop, out)); th.start()
t0 = time.time(); n = 0
while time.time() - t0 < 3.0:
    for _ in range(50): fn()
    torch.cuda.synchronize(); n += 50
dt = time.time() - t0
stop.set(); th.join()
clk = [c for c, _ in out if isinstance(c, int)]; pw = [p for _, p in out if isinstance(p, float)]
print(json.dumps({"shape": [N, C, H, W], "rows": os.environ.get("COALIGN_EMU_TAPK_ROWS", "default"), "stack": os.environ.get("COALIGN_EMU_STACK", "default"),
                  "stride": STRIDE, "us_per_call": round(dt / n * 1e6, 1), "sclk_MHz": clk[2:-1], "power_W": [round(p) for p in pw[2:-1]]}))
