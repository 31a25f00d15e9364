"""Is the stacked-tile convolution deterministic next to a second busy stream?  Per shape: 300 launches, every output compared on the device."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from coalign_amd import ops
g = torch.Generator().manual_seed(7)
side = torch.cuda.Stream()
x2 = torch.randn(5, 64, 100, 352, generator=g).cuda(); w2 = ops.pack_conv3x3_emu_weight((torch.randn(64, 64, 3, 3, generator=g) / 24.0).cuda(), 3, True); b2 = torch.randn(64, generator=g).cuda()
scratch = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
for (N, C, H, W) in ((5, 128, 50, 176), (5, 256, 25, 88)):
    x = torch.randn(N, C, H, W, generator=g).cuda(); w = (torch.randn(C, C, 3, 3, generator=g) / (C * 9) ** 0.5).cuda()
    b = torch.randn(C, generator=g).cuda(); r = torch.randn(N, C, H, W, generator=g).cuda()
    ws = ops.pack_conv3x3_emu_weight(w, 3, True)
    for cl in (False, True):
        run = lambda: ops.conv3x3_emu_bias_act(x, ws, b, C, r, True, 3, out_channels_last=cl)
        ref = run().clone()
        for mode in ("alone", "busy"):
            bad = torch.zeros((), dtype=torch.int64, device="cuda")
            for it in range(300):
                if mode == "busy" and it % 2 == 0:
                    with torch.cuda.stream(side):
                        ops.conv3x3_emu_bias_act(x2, w2, b2, 64, None, True, 3)
                        if it % 6 == 0:
                            scratch.zero_()
                y = run()
                bad += (y != ref).any()
            torch.cuda.synchronize()
            print(f"{N}x{C}x{H}x{W} nhwc_out={cl} {mode}: {int(bad)} of 300 launches differ")
