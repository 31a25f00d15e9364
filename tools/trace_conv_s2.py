#!/usr/bin/env python
"""Interval timeline of coalign_conv3x3_sp_s2 (laboratory library, COALIGN_LAB=1): per interval of workgroups 0 and 100, wavefronts 0 and 7 -- clocks spent in the
closing s_waitcnt, in the barrier, in the interval's body (matrix steps + DMA issue + a deferred epilogue) and the total."""
import os
import sys
import ctypes

os.environ["COALIGN_LAB"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from coalign_amd import ops  # noqa: E402

L = ops.hip.lib()
L.coalign_conv3x3_sp_s2_set_trace.argtypes = [ctypes.c_void_p]
shape = tuple(int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "5,64,128,100,352").split(","))
N, Ci, Co, H, W = shape
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.relu(torch.randn((N, Ci, H, W), generator=g, device="cuda"))
w = ops.pack_conv3x3_emu_weight(torch.randn((Co, Ci, 3, 3), generator=g, device="cuda") / (9 * Ci) ** 0.5, 16, True)
b = torch.randn(Co, generator=g, device="cuda")
xs = ops.SplitMap.pack(x)
for _ in range(3):
    ops.conv3x3_sp_s2(xs, w, b, Co, True)
trace = torch.zeros(2 * 8 * 64 * 4, dtype=torch.int64, device="cuda")
L.coalign_conv3x3_sp_s2_set_trace(ctypes.c_void_p(trace.data_ptr()))
ops.conv3x3_sp_s2(xs, w, b, Co, True)
torch.cuda.synchronize()
L.coalign_conv3x3_sp_s2_set_trace(ctypes.c_void_p(0))
t = trace.cpu().reshape(2, 8, 64, 4)
print(f"shape {shape} ablate {os.environ.get('COALIGN_S2_ABLATE', '0')}")
for wg in (0, 1):
    for wave in (0, 7):
        print(f"workgroup {wg * 100} wave {wave}: interval   wait   barrier   body | total  (clocks of s_memtime: 100 MHz x ... see below)")
        r = t[wg, wave]
        n = int((r[:, 0] > 0).sum())
        for i in range(n):
            nxt = r[i + 1, 0] if i + 1 < n else r[i, 3]
            print(f"   {i:3d} {int(r[i, 1] - r[i, 0]):7d} {int(r[i, 2] - r[i, 1]):7d} {int(r[i, 3] - r[i, 2]):7d} | {int(nxt - r[i, 0]):7d}")
        if n:
            print(f"   whole range: {int(r[n - 1, 3] - r[0, 0])} clocks for {n} intervals")
