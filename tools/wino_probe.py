#!/usr/bin/env python
"""One stride-1 3x3 layer shape on the Winograd (KERNEL=wino) or the direct tap-major split-bf16 kernel (KERNEL=direct), back to back.
  python tools/wino_probe.py power    engine clock + board power sampled with rocm-smi for 3 s, microseconds per call
  python tools/wino_probe.py few [n]  n calls (for rocprofv3 --pmc / --kernel-trace passes)
SHAPE=N,C,H,W (Cin = Cout = C).  TBW=0|8|16."""
import json, os, re, subprocess, sys, threading, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coalign_amd import ops

N, C, H, W = [int(v) for v in os.environ.get("SHAPE", "5,256,25,88").split(",")]
kern, tbw = os.environ.get("KERNEL", "wino"), int(os.environ.get("TBW", "0"))
g = torch.Generator().manual_seed(0)
x = torch.randn(N, C, H, W, generator=g).cuda(); w = (torch.randn(C, C, 3, 3, generator=g) / (C * 9) ** 0.5).cuda()
b = torch.randn(C, generator=g).cuda(); r = torch.randn(N, C, H, W, generator=g).cuda()
if kern == "wino":
    xl, rl, u = x.contiguous(memory_format=torch.channels_last), r.contiguous(memory_format=torch.channels_last), ops.pack_conv3x3_wino_weight(w)
    fn = lambda: ops.conv3x3_wino(xl, u, b, C, rl, True, tile_block_w=tbw)
else:
    wd = ops.pack_conv3x3_emu_weight(w, 3, True)
    fn = lambda: ops.conv3x3_emu_bias_act(x, wd, b, C, r, True, 3)
mode = sys.argv[1] if len(sys.argv) > 1 else "power"
if mode == "few":
    for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 5): fn()
    torch.cuda.synchronize()
    sys.exit(0)


def sample(stop, out):
    while not stop.is_set():
        try:
            d = json.loads(subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=5).stdout)
            card = next(iter(d.values()))
            sclk = next((v for k, v in card.items() if "sclk" in k.lower()), None)
            pw = next((v for k, v in card.items() if "power" in k.lower() and "W" in k), None)
            m = re.search(r"(\d+)\s*Mhz", str(sclk), re.I)
            out.append((int(m.group(1)) if m else None, float(pw) if pw not in (None, "N/A") else None))
        except Exception as e:      # noqa: BLE001
            out.append(("err", str(e)[:60]))
        time.sleep(0.2)


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
us = dt / n * 1e6
mean = lambda v: round(sum(v) / max(len(v), 1))
print(json.dumps({"kernel": kern, "shape": [N, C, H, W], "us_per_call": round(us, 1), "sclk_MHz_mean": mean(clk[2:-1]), "power_W_mean": mean(pw[2:-1]),
                  "mJ_per_call": round(mean(pw[2:-1]) * us * 1e-3, 1), "sclk_MHz": clk[2:-1], "power_W": [round(p) for p in pw[2:-1]]}))
