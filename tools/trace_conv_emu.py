#!/usr/bin/env python
"""Per-interval timeline of conv3x3_emu_kernel (profiling aid): builds csrc/conv3x3_emu.hip with -DEMU_TRACE into a private
library and prints, per barrier interval of workgroup 0 (first and last wavefront), the time (s_memtime ticks = shader clocks) spent
  wait   : s_waitcnt(0) at the top (own weight DMA, own LDS writes, own stores of a finished tile)
  barA   : the barrier that opens the interval
  issue  : weight DMA + halo-pixel loads of the next interval
  steps  : the matrix-instruction steps
  barB   : the barrier that frees the single split-patch buffer
  store  : fp32 -> bf16 split + LDS writes of the next interval's pixels
  gap    : from the end of this interval to the top of the next (tile epilogue + next tile's accumulator start, when a tile ends)
Usage: python tools/trace_conv_emu.py [terms] [N Cin Cout H W] ; TAPK=1 selects the tap-major image (COALIGN_EMU_TAPK_ROWS / _VAR apply),
RESIDUAL=1 adds a residual input, STRIDE=2 traces the strided layers (channels-last in, SplitMap out)."""
import ctypes, os, subprocess, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
src = os.path.join(ROOT, "coalign_amd", "csrc")
lib = os.path.join(ROOT, "coalign_amd", "lib", "libemu_trace.so")
if not os.path.exists(lib) or os.environ.get("REBUILD"):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-x", "hip", "-O3", "-std=c++17", "-ffp-contract=off", "--offload-arch=gfx950", "-shared", "-fPIC", "-DEMU_TRACE", "-I", src,
                           "-I", os.path.join(ROOT, "include"), os.path.join(src, "conv3x3_emu.hip"), os.path.join(src, "status.cpp"), "-o", lib])
if not torch.cuda.is_available():
    sys.exit(0)
from coalign_amd import ops
terms = int(sys.argv[1]) if len(sys.argv) > 1 else 3
N, Ci, Co, H, W = [int(v) for v in sys.argv[2:7]] if len(sys.argv) > 6 else (5, 64, 64, 100, 352)
tapk = os.environ.get("TAPK", "0") == "1"
layout = 4 if tapk else 0
stride = int(os.environ.get("STRIDE", "1"))      # STRIDE=2: the first convolution of a stage as the product runs it (channels-last input, SplitMap output, terms 16)
if stride == 2:
    layout = 2 | 8
L = ctypes.CDLL(lib)
L.coalign_conv3x3_emu_ex.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 9 + [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
L.coalign_conv3x3_emu_workspace_bytes_ex.restype = ctypes.c_size_t
L.coalign_conv3x3_emu_workspace_bytes_ex.argtypes = [ctypes.c_int] * 7
x = torch.randn(N, Ci, H, W, device="cuda")
if stride == 2:
    x = x.contiguous(memory_format=torch.channels_last)
w = torch.randn(Co, Ci, 3, 3, device="cuda") / (Ci * 9) ** 0.5
b = torch.randn(Co, device="cuda"); y = torch.empty(N, Co, (H + stride - 1) // stride, (W + stride - 1) // stride, device="cuda")
res = torch.randn(N, Co, H, W, device="cuda") if os.environ.get("RESIDUAL", "0") == "1" else None
ws = ops.pack_conv3x3_emu_weight(w, terms, tapk)
waves, S = 16, 8
tr = torch.zeros(2 * waves * 64 * S + 2 * 4096, dtype=torch.int64, device="cuda")
L.coalign_conv3x3_emu_set_trace(ctypes.c_void_p(tr.data_ptr()))
scratch = torch.empty(max(1, L.coalign_conv3x3_emu_workspace_bytes_ex(N, Ci, Co, H, W, terms, layout)), dtype=torch.uint8, device="cuda")
for _ in range(3):
    tr.zero_()
    rc = L.coalign_conv3x3_emu_ex(x.data_ptr(), ws.data_ptr(), b.data_ptr(), None if res is None else res.data_ptr(), y.data_ptr(), N, Ci, Co, H, W, stride, 1, terms, layout, None,
                                  scratch.data_ptr(), scratch.numel(), None)
    torch.cuda.synchronize()
assert rc == 0, rc
if not os.environ.get("COALIGN_EMU_PC"):        # (the producer / consumer kernel records no stamps)
    nw = int(os.environ.get("WAVES", 8))
    span = tr.cpu()[2 * waves * 64 * S:].view(-1, 2)
    span = span[span[:, 1] > 0]
    t0 = int(span[:, 0].min())
    st, en = (span[:, 0] - t0).float() / 100.0, (span[:, 1] - t0).float() / 100.0          # us
    print(f"{'tap-major' if tapk else 'tap pairs'} {N}x{Ci}->{Co} {H}x{W} terms {terms} residual {res is not None}: {len(span)} workgroups: start  min {st.min():.1f} median {st.median():.1f} max {st.max():.1f} us;"
          f"  end  min {en.min():.1f} median {en.median():.1f} max {en.max():.1f} us")
    print("  per-workgroup duration: min %.1f median %.1f max %.1f us" % ((en - st).min(), (en - st).median(), (en - st).max()))
    t = tr.cpu()[: 2 * waves * 64 * S].view(-1)[: nw * 64 * S].reshape(nw, 64, S)        # workgroup 0
    for wv in (0, nw - 1):
        print(f"workgroup 0 wave {wv}: interval    wait    barA   issue   steps    barB   store     gap | total (clocks)")
        for c in range(20):
            s = t[wv, c]
            if s[6] == 0: break
            nxt = t[wv, c + 1][0] if c + 1 < 64 and t[wv, c + 1][0] else s[6]
            d = [int(s[k + 1] - s[k]) for k in range(6)] + [int(nxt - s[6])]
            print("   %3d %s | %7d" % (c, " ".join("%7d" % v for v in d), int(nxt - s[0])))
else:
    print(f"producer / consumer kernel {N}x{Ci}->{Co} {H}x{W} terms {terms} residual {res is not None}")
# what each part of an interval costs: kernel time with the weight DMA / the halo-pixel loads / the matrix steps switched off
L.coalign_conv3x3_emu_set_ablate.argtypes = [ctypes.c_int]
L.coalign_conv3x3_emu_set_trace(None)
def run():
    return L.coalign_conv3x3_emu_ex(x.data_ptr(), ws.data_ptr(), b.data_ptr(), None if res is None else res.data_ptr(), y.data_ptr(), N, Ci, Co, H, W, stride, 1, terms, layout, None,
                                    scratch.data_ptr(), scratch.numel(), None)
tr.zero_()
L.coalign_conv3x3_emu_set_trace(ctypes.c_void_p(tr.data_ptr()))
out = {}
for ab, name in ((0, "all"), (1, "no weight DMA"), (2, "no pixel loads"), (3, "no DMA, no loads"), (4, "no matrix steps"), (7, "barriers + split only"),
                 (8, "no tile start / epilogue (pc kernel)"), (11, "steps + barriers only (pc kernel)"), (15, "barriers only (pc kernel)")):
    L.coalign_conv3x3_emu_set_ablate(ab)
    for _ in range(3): run()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): run()
    e.record(); torch.cuda.synchronize()
    out[name] = round(s.elapsed_time(e) * 100, 1)
print("kernel us by ablation (trace stamps on):", out)
