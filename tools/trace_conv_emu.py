#!/usr/bin/env python
"""Per-chunk timeline of conv3x3_emu_kernel (profiling aid): builds csrc/conv3x3_emu.hip with -DEMU_TRACE into a private
library and prints, per chunk of workgroups 0 and 100, the cycles spent waiting (DMA + barrier), issuing the next DMA, in the
step prologue and in the MFMA steps.  Usage: python tools/trace_conv_emu.py [terms] [N Cin Cout H W]"""
import ctypes, os, subprocess, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
src = os.path.join(ROOT, "coalign_amd", "csrc")
lib = os.path.join(ROOT, "coalign_amd", "lib", "libemu_trace.so")
if not os.path.exists(lib) or os.environ.get("REBUILD"):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-x", "hip", "-O3", "--offload-arch=gfx950", "-shared", "-fPIC", "-DEMU_TRACE", "-I", src,
                           "-I", os.path.join(ROOT, "include"), os.path.join(src, "conv3x3_emu.hip"), os.path.join(src, "status.cpp"), "-o", lib])
if not torch.cuda.is_available():
    sys.exit(0)
from coalign_amd import ops
terms = int(sys.argv[1]) if len(sys.argv) > 1 else 2
N, Ci, Co, H, W = [int(v) for v in sys.argv[2:7]] if len(sys.argv) > 6 else (5, 64, 64, 100, 352)
L = ctypes.CDLL(lib)
L.coalign_conv3x3_emu_bias_act.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 7 + [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
L.coalign_conv3x3_emu_workspace_bytes.restype = ctypes.c_size_t
L.coalign_conv3x3_emu_workspace_bytes.argtypes = [ctypes.c_int] * 6
x = torch.randn(N, Ci, H, W, device="cuda"); w = torch.randn(Co, Ci, 3, 3, device="cuda") / (Ci * 9) ** 0.5
b = torch.randn(Co, device="cuda"); y = torch.empty(N, Co, H, W, device="cuda")
ws = ops.pack_conv3x3_emu_weight(w, terms)
waves = 16
tr = torch.zeros(2 * waves * 64 * 5 + 2 * 4096, dtype=torch.int64, device="cuda")
L.coalign_conv3x3_emu_set_trace(ctypes.c_void_p(tr.data_ptr()))
scratch = torch.empty(max(1, L.coalign_conv3x3_emu_workspace_bytes(N, Ci, Co, H, W, terms)), dtype=torch.uint8, device="cuda")
for _ in range(3):
    tr.zero_()
    rc = L.coalign_conv3x3_emu_bias_act(x.data_ptr(), ws.data_ptr(), b.data_ptr(), None, y.data_ptr(), N, Ci, Co, H, W, 1, terms, scratch.data_ptr(), scratch.numel(), None)
    torch.cuda.synchronize()
assert rc == 0
nw = int(os.environ.get("WAVES", 8))
span = tr.cpu()[2 * waves * 64 * 5:].view(-1, 2)
span = span[span[:, 1] > 0]
t0 = int(span[:, 0].min())
st, en = (span[:, 0] - t0).float() / 100.0, (span[:, 1] - t0).float() / 100.0          # us
print(f"{len(span)} workgroups: start  min {st.min():.1f} median {st.median():.1f} max {st.max():.1f} us;  end  min {en.min():.1f} median {en.median():.1f} max {en.max():.1f} us")
print("  per-workgroup duration: min %.1f median %.1f max %.1f us" % ((en - st).min(), (en - st).median(), (en - st).max()))
t = tr.cpu()[: 2 * waves * 64 * 5].view(2, -1)
t = torch.stack([t[0, : nw * 64 * 5], tr.cpu()[nw * 64 * 5: 2 * nw * 64 * 5]]).reshape(2, nw, 64, 5)
for g in (0,):
    for wv in (0, nw - 1):
        print(f"workgroup {'0' if g == 0 else '100'} wave {wv}: chunk  own-DMA-wait  barrier  issue  steps  | total")
        for c in range(24):
            s = t[g, wv, c]
            if s[4] == 0: break
            nxt = t[g, wv, c + 1][0] if c + 1 < 64 and t[g, wv, c + 1][0] else s[4]
            print(f"   {c:3d} {int(s[1]-s[0]):8d} {int(s[2]-s[1]):8d} {int(s[3]-s[2]):8d} {int(s[4]-s[3]):8d}  | {int(nxt - s[0]):8d}")
