"""The warped maps themselves (fusion mode NONE: bilinear gather only) while the convolution variant runs beside the kernel: which elements
are wrong and where do the wrong values come from."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from coalign_amd import ops
from coalign_amd.config import builtin_config
from coalign_amd.pose import normalize_pairwise_tfm
from coalign_amd.synthetic import make_frame
g = torch.Generator().manual_seed(3)
N = 5
h = builtin_config("opv2v_coalign")
fr = make_frame(h, N, pillars_per_agent=100, seed=303, noise=(0.2, 0.2))
theta = normalize_pairwise_tfm(fr["pairwise_t_matrix"].cuda(), 200, 704, 0.4)[0, 0, :N].contiguous()
xcl = [torch.randn(N, C, H, W, generator=g).cuda().contiguous(memory_format=torch.channels_last) for C, H, W in ((64, 100, 352), (128, 50, 176), (256, 25, 88))]
C, H, W = 256, 25, 88
x = torch.randn(N, C, H, W, generator=g).cuda(); w = ops.pack_conv3x3_emu_weight((torch.randn(C, C, 3, 3, generator=g) / (C * 9) ** 0.5).cuda(), 3, True)
b = torch.randn(C, generator=g).cuda(); r = torch.randn(N, C, H, W, generator=g).cuda()
side = torch.cuda.Stream()
MODE = ops.FUSE_NONE
ref = [t.clone() for t in ops.warp_fuse_nhwc(xcl, theta, MODE)]
yref = ops.conv3x3_emu_bias_act(x, w, b, C, r, True, 3).clone()
torch.cuda.synchronize()
pools = {"xcl0": xcl[0], "xcl1": xcl[1], "xcl2": xcl[2], "conv x": x, "conv residual": r, "conv y": yref, "conv bias": b, "conv weights (as f32 words)": w.view(torch.float32) if w.dtype != torch.float32 else w}
shown = 0
for it in range(300):
    with torch.cuda.stream(side):
        ys = [ops.conv3x3_emu_bias_act(x, w, b, C, r, True, 3) for _ in range(3)]
    out = ops.warp_fuse_nhwc(xcl, theta, MODE)
    torch.cuda.synchronize()
    for k, (a, rf) in enumerate(zip(out, ref)):
        if not torch.equal(a, rf) and shown < 8:
            shown += 1
            d = (a != rf).nonzero()
            print(f"iter {it} scale {k} ({tuple(a.shape)}): {d.shape[0]} elements differ; agents {sorted(set(d[:, 0].tolist()))}, pixels {sorted(set((y_, x_) for _, _, y_, x_ in d.tolist()))[:8]}, "
                  f"channels {sorted(set(d[:, 1].tolist()))[:40]}")
            per = {}
            for n_, c_, y_, x_ in d.tolist():
                per.setdefault((n_, y_, x_), []).append(c_)
            Cs = a.shape[1]
            for key in list(per)[:6]:
                chs = per[key]
                print(f"   agent {key[0]} pixel ({key[1]}, {key[2]}): {len(chs)} channels, first {chs[:8]}")
                c0 = (chs[0] // 4) * 4 if chs[0] < Cs // 2 else ((chs[0] - Cs // 2) // 4) * 4      # low group of the lane that owns the first wrong channel
                lane_ch = [c0, c0 + 1, c0 + 2, c0 + 3, Cs // 2 + c0, Cs // 2 + c0 + 1, Cs // 2 + c0 + 2, Cs // 2 + c0 + 3]
                print("      lane registers X[0..7] got :", [round(a[key[0], c, key[1], key[2]].item(), 6) for c in lane_ch])
                print("      lane registers X[0..7] want:", [round(rf[key[0], c, key[1], key[2]].item(), 6) for c in lane_ch])
            for idx in d[:0].tolist():
                got, want = a[tuple(idx)].item(), rf[tuple(idx)].item()
                src = []
                for name, t in pools.items():
                    hit = (t == got).nonzero()
                    if hit.shape[0]:
                        src.append(f"{name}{hit[0].tolist()}")
                print(f"   {idx}: got {got!r} want {want!r}  got-value found in: {src or 'nowhere'}")
print("differing launches shown:", shown)
