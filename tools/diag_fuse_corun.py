"""Does warp_fuse_nhwc give the same result while another stream runs convolution kernels?  500 fusion launches per co-runner, every output compared."""
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
def conv_setup(N_, C, H, W):
    x = torch.randn(N_, C, H, W, generator=g).cuda(); w = ops.pack_conv3x3_emu_weight((torch.randn(C, C, 3, 3, generator=g) / (C * 9) ** 0.5).cuda(), 3, True)
    b = torch.randn(C, generator=g).cuda(); r = torch.randn(N_, C, H, W, generator=g).cuda()
    return lambda cl=False: ops.conv3x3_emu_bias_act(x, w, b, C, r, True, 3, out_channels_last=cl)
corun = {"conv 5x256x25x88": conv_setup(5, 256, 25, 88)} if os.environ.get("ONLY25") else {
         "none": None, "conv 5x64x100x352 (12-row tiles)": conv_setup(5, 64, 100, 352), "conv 5x128x50x176": conv_setup(5, 128, 50, 176), "conv 5x256x25x88": conv_setup(5, 256, 25, 88),
         "conv 1x256x100x352 (stream-K)": conv_setup(1, 256, 100, 352)}
side = torch.cuda.Stream()
VICTIM = os.environ.get("VICTIM", "fuse")
big = torch.randn(3, 8 << 20, generator=g).cuda()
vconv = conv_setup(5, 64, 100, 352)
def victim():
    if VICTIM == "fuse":
        return ops.warp_fuse_nhwc(xcl, theta, ops.FUSE_ATT)
    if VICTIM == "clone":
        return [big[0].clone(), big[1].clone(), big[2].clone()]
    if VICTIM == "fma":
        return [big[0] * 1.5 + big[1], big[1] * big[2], torch.softmax(big[2].view(-1, 256), 1)]
    if VICTIM == "f64":
        d = [b[: 2 << 20].double() for b in big]
        return [(d[0] * 1.0000001 + d[1]) / (d[2].abs() + 1.0), (2.0 * d[1] + 1.0) / 352.0 - 1.0, (d[0] * d[1] + d[2]).float()]
    if VICTIM == "conv":
        os.environ.pop("COALIGN_EMU_TAPK_ROWS", None)      # (read once by the library: the forced rows apply to this one too -- 64 channels at 100 x 352 take any)
        return [vconv(False), vconv(True), vconv(False)]
    raise SystemExit("VICTIM?")
ref = [t.clone() for t in victim()]
for name, fn in corun.items():
    for cl in ((False,) if fn is None else (False, True)):
        bad = torch.zeros((), dtype=torch.int64, device="cuda")
        per_scale = torch.zeros(3, dtype=torch.int64, device="cuda")
        for it in range(500):
            if fn is not None:
                with torch.cuda.stream(side):
                    fn(cl)
            out = victim()
            for k, (a, b) in enumerate(zip(out, ref)):
                d = (a != b).any()
                bad += d
                per_scale[k] += d
        torch.cuda.synchronize()
        print(f"victim {VICTIM}, co-runner {name} nhwc_out={cl}: {int(bad)} of 1500 fused maps differ (per scale {per_scale.tolist()})")
