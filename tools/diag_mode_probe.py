"""Runs tools/microbench/mode_probe.hip beside the convolution variant named by COALIGN_EMU_TAPK_ROWS: do all wavefronts still compute the
same bits from the same inputs, and what do MODE / HW_ID say about the ones that do not."""
import collections, ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from coalign_amd import ops
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "microbench", "libmode_probe.so"))
lib.mode_probe_launch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
g = torch.Generator().manual_seed(3)
N, C, H, W = 5, 256, 25, 88
x = torch.randn(N, C, H, W, generator=g).cuda(); w = ops.pack_conv3x3_emu_weight((torch.randn(C, C, 3, 3, generator=g) / (C * 9) ** 0.5).cuda(), 3, True)
b = torch.randn(C, generator=g).cuda(); r = torch.randn(N, C, H, W, generator=g).cuda()
src = torch.randn(1024, generator=g).cuda()
WG, ITERS = int(os.environ.get("WG", 2048)), int(os.environ.get("ITERS", 40))
side = torch.cuda.Stream()
def probe():
    out = torch.zeros(WG * 4, 16, dtype=torch.int32, device="cuda")
    rc = lib.mode_probe_launch(out.data_ptr(), src.data_ptr(), WG, ITERS, torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc
    return out
ref = probe(); torch.cuda.synchronize()
ref = ref.cpu()
print("alone: distinct (mode0, mode1, xa, xb, xd):", collections.Counter(map(tuple, ref[:, [0, 1, 3, 4, 5, 8, 9]].tolist())).most_common(3))
good = tuple(ref[0, [3, 4, 5, 8, 9]].tolist())
events = 0
for it in range(int(os.environ.get("LAUNCHES", 300))):
    with torch.cuda.stream(side):
        ys = [ops.conv3x3_emu_bias_act(x, w, b, C, r, True, 3) for _ in range(2)]
    out = probe()
    torch.cuda.synchronize()
    o = out.cpu()
    bad = [i for i in range(o.shape[0]) if tuple(o[i, [3, 4, 5, 8, 9]].tolist()) != good]
    if bad:
        events += 1
        if events <= 5:
            print(f"launch {it}: {len(bad)} of {o.shape[0]} wavefronts differ")
            for i in bad[:10]:
                m0, m1, hw, xa, xb, xd, st, tr = [v & 0xFFFFFFFF for v in o[i, :8].tolist()]
                print(f"   wave {i}: mode {m0:#x} -> {m1:#x}  hw_id {hw:#010x} (simd {(hw >> 4) & 3}, cu {(hw >> 8) & 15}, se {(hw >> 13) & 7})  status {st:#x} trapsts {tr:#x}  "
                      f"lane mask {int(o[i, 9]) & 0xFFFFFFFF:#010x}{int(o[i, 8]) & 0xFFFFFFFF:08x}  a^ {xa:#010x} (want {good[0] & 0xFFFFFFFF:#010x})  b^ {xb:#010x} ({good[1] & 0xFFFFFFFF:#010x})  d^ {xd:#010x} ({good[2] & 0xFFFFFFFF:#010x})")
            modes = collections.Counter((int(o[i, 0]) & 0xFFFFFFFF, int(o[i, 1]) & 0xFFFFFFFF) for i in range(o.shape[0]))
            print("   (mode0, mode1) over all wavefronts of this launch:", {f"{a:#x}->{b:#x}": c for (a, b), c in modes.items()})
print(f"launches with differing wavefronts: {events}")
