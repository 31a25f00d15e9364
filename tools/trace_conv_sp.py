#!/usr/bin/env python
"""Per-interval timeline + ablations of conv3x3_sp_kernel (profiling aid): builds csrc/conv3x3_sp.hip with -DSP_TRACE into a private library
(coalign_amd/lib/libsp_trace.so, built on the CPU side) and prints, per barrier interval of workgroup 0 (first and last wavefront), the shader clocks spent in
  wait  : s_waitcnt(0) at the top (own DMA of this interval, own stores of a finished tile)      barrier : the interval's one barrier
  issue : LDS-DMA of the next interval (+ next tile's plan at a tile's last interval)           steps   : LDS reads + matrix instructions
  gap   : end of the steps to the top of the next interval (tile epilogue + next tile's start value, when a tile ends)
and the kernel time with parts switched off.  Usage: python tools/trace_conv_sp.py N Cin Cout H W [geometry]"""
import ctypes
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
src = os.path.join(ROOT, "coalign_amd", "csrc")
lib = os.path.join(ROOT, "coalign_amd", "lib", "libsp_trace.so")
if not os.path.exists(lib) or os.environ.get("REBUILD"):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-x", "hip", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fno-vectorize", "--offload-arch=gfx950", "-shared", "-fPIC",
                           "-DSP_TRACE", "-I", src, "-I", os.path.join(ROOT, "include"), os.path.join(src, "conv3x3_sp.hip"), os.path.join(src, "conv3x3_emu.hip"),
                           os.path.join(src, "status.cpp"), "-o", lib])
if not torch.cuda.is_available():
    sys.exit(0)
from coalign_amd import ops  # noqa: E402

N, Ci, Co, H, W = [int(v) for v in sys.argv[1:6]] if len(sys.argv) > 5 else (5, 256, 256, 25, 88)
geo = int(sys.argv[6]) if len(sys.argv) > 6 else 0
L = ctypes.CDLL(lib)
P, I = ctypes.c_void_p, ctypes.c_int
L.coalign_conv3x3_sp.argtypes = [P, P, P, P, I, P, I, I, I, I, I, I, I, I, P, P, ctypes.c_size_t, P]
L.coalign_conv3x3_sp_workspace_bytes.restype = ctypes.c_size_t
L.coalign_conv3x3_sp_workspace_bytes.argtypes = [I] * 6
x = torch.relu(torch.randn(N, Ci, H, W, device="cuda"))
w = torch.randn(Co, Ci, 3, 3, device="cuda") / (Ci * 9) ** 0.5
b = torch.randn(Co, device="cuda")
r = torch.randn(N, Co, H, W, device="cuda")
w16 = ops.pack_conv3x3_emu_weight(w, 16, True)
xs, rs = ops.SplitMap.pack(x), ops.SplitMap.pack(r)
out = ops.SplitMap.empty(N, Co, H, W, "cuda")
waves, S = 16, 8
tr = torch.zeros(2 * waves * 64 * S + 3 * 4096, dtype=torch.int64, device="cuda")
L.coalign_conv3x3_sp_set_trace(P(tr.data_ptr()))


wsb = L.coalign_conv3x3_sp_workspace_bytes(N, Ci, Co, H, W, geo)
ws = torch.zeros(max(wsb, 16), dtype=torch.uint8, device="cuda")


PRIO = 32 if os.environ.get("PRIO") else 0      # round 6: progress-based issue priority on (bit 5 of the trace build's ablation word)


def run(ablate=0, n=1):
    L.coalign_conv3x3_sp_set_ablate(ablate | PRIO)
    for _ in range(n):
        rc = L.coalign_conv3x3_sp(xs.data.data_ptr(), w16.data_ptr(), b.data_ptr(), rs.data.data_ptr(), 1, out.data.data_ptr(), 1, N, Ci, Co, H, W, 1, geo, None,
                                  ws.data_ptr() if wsb else None, wsb, None)
        assert rc == 0, rc


for _ in range(3):
    tr.zero_()
    run()
    torch.cuda.synchronize()
hwid = tr.cpu()[2 * waves * 64 * S + 2 * 4096:]
span_all = tr.cpu()[2 * waves * 64 * S: 2 * waves * 64 * S + 2 * 4096].view(-1, 2)
live = span_all[:, 1] > 0
# round 6: where the workgroups ran (HW_ID: CU_ID bits 8-11, SH_ID 12, SE_ID 13-15, TG_ID 16-19; the XCD is not in HW_ID: workgroup g runs on XCD g % 8) and how
# many of them shared a CU at the same time -- the paired mode (geometry 7000 + g) wants two per CU
ids = [(int(g) % 8, (int(h) >> 13) & 7, (int(h) >> 12) & 1, (int(h) >> 8) & 15, (int(h) >> 16) & 15, int(a), int(b)) for g, (h, (a, b)) in enumerate(zip(hwid[: len(span_all)].tolist(), span_all.tolist())) if b > 0]
cus = {}
for xcd, se, sh, cu, tg, a0, b0 in ids:
    cus.setdefault((xcd, se, sh, cu), []).append((a0, b0, tg))
pairs = sum(1 for v in cus.values() for i in range(len(v)) for j in range(i) if min(v[i][1], v[j][1]) - max(v[i][0], v[j][0]) > 0.5 * min(v[i][1] - v[i][0], v[j][1] - v[j][0]))
print(f"{len(ids)} workgroups on {len(cus)} distinct (XCD, SE, SH, CU); workgroups per CU: max {max(len(v) for v in cus.values())}; pairs overlapping in time for > half of the shorter one's life: {pairs}; TG slots seen: {sorted({t for v in cus.values() for _, _, t in v})}")
span = span_all[live]
t0 = int(span[:, 0].min())
st, en = (span[:, 0] - t0).float() / 100.0, (span[:, 1] - t0).float() / 100.0
print(f"conv3x3_sp {N}x{Ci}->{Co} {H}x{W} geometry {geo}: {len(span)} workgroups: start median {st.median():.1f} max {st.max():.1f} us; end min {en.min():.1f} median {en.median():.1f} max {en.max():.1f} us")
t = tr.cpu()[: 2 * waves * 64 * S].view(2, waves, 64, S)
nw = int((t[0, :, 0, 0] > 0).sum())
for wg in (0, 1):
    for wv in (0, nw - 1):
        if int(t[wg, wv, 0, 0]) == 0:
            continue
        print(f"workgroup {'0' if wg == 0 else '100'} wave {wv}: interval   wait barrier  issue  steps    gap | total (clocks)")
        for c in range(40):
            s = t[wg, wv, c]
            if int(s[0]) == 0:
                break
            nxt = int(t[wg, wv, c + 1, 0]) if c + 1 < 64 and int(t[wg, wv, c + 1, 0]) else int(s[4])
            d = [int(s[1] - s[0]), int(s[2] - s[1]), int(s[3] - s[2]), int(s[4] - s[3]), nxt - int(s[4])]
            if geo % 100000 // 1000 == 7 and int(s[5]) > int(s[3]) and int(s[6]) >= int(s[5]) and int(s[4]) >= int(s[6]):      # paired mode: steps | the barrier that ends the interval | DMA issue
                d[3] = int(s[5] - s[3])
                d.insert(4, int(s[6] - s[5]))
                d.insert(5, int(s[4] - s[6]))
            epi = ""
            if int(s[5]) and int(s[7]):      # a tile's last interval: the gap in parts (accumulator join + hand-over | residual wait + conversion | the 8 channel groups: scale, split, stores | to the next top)
                epi = f"   epilogue: join {int(s[5] - s[4])} residual {int(s[6] - s[5])} groups {int(s[7] - s[6])} rest {nxt - int(s[7])}"
            print(f"   {c:3d} " + " ".join(f"{v:7d}" for v in d) + f" | {nxt - int(s[0]):7d}" + epi)


if os.environ.get("ALLWAVES"):      # round 6: who arrives last at an interval's barrier -- every wavefront's stamps of intervals 2..4 of workgroup 0, relative to the barrier's release
    for c in (2, 3, 4):
        rel = int(t[0, :nw, c, 2].max())
        print(f"interval {c} of workgroup 0, clocks relative to the barrier's release: wavefront: top-of-interval  own-DMA-landed  released | issue-done steps-done (of this interval)")
        for wv in range(nw):
            s = t[0, wv, c]
            print(f"   wave {wv:2d} (SIMD slot {wv % 4}): {int(s[0]) - rel:7d} {int(s[1]) - rel:7d} {int(s[2]) - rel:7d} | {int(s[3]) - rel:7d} {int(s[4]) - rel:7d}")


def timed(ablate):
    run(ablate, 3)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    run(ablate, 20)
    e.record()
    torch.cuda.synchronize()
    return round(s.elapsed_time(e) / 20 * 1e3, 1)


names = {0: "all", 1: "no weight DMA", 2: "no patch DMA", 3: "no DMA", 4: "no matrix steps / LDS reads", 7: "barriers + tile start / end only", 8: "no residual / bias start", 16: "no stores",
         24: "no tile start / stores", 28: "DMA + barriers only"}
print("kernel us by ablation (trace build with the stamps on, eager launches):", {v: timed(k) for k, v in names.items()})
