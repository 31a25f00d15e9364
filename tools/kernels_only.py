"""Launch only the hand-written ops on bench-shaped data (for rocprofv3 --pmc passes and quick A/B timing).
usage: python tools/kernels_only.py [iters]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from coalign_amd import ops
from coalign_amd.config import builtin_config
from coalign_amd.detector import build_model, to_device
from coalign_amd.pose import normalize_pairwise_tfm
from coalign_amd.synthetic import fill_parameters_, make_frame

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device("cuda:0")
h = builtin_config("opv2v_coalign")
N = int(os.environ.get("AGENTS", "5"))
model = build_model(h); fill_parameters_(model, seed=0); model = model.to(dev).eval()
fr = make_frame(h, N, pillars_per_agent=8000, seed=303, noise=(0.2, 0.2))
frd = to_device(fr, dev)
aff = normalize_pairwise_tfm(frd["pairwise_t_matrix"], 200, 704, 0.4)
theta = aff[0, 0, :N].contiguous()
g = torch.Generator(device="cpu").manual_seed(0)
xs = [torch.randn(N, C, H, W, generator=g).to(dev) for C, H, W in ((64, 100, 352), (128, 50, 176), (256, 25, 88))]
pl = frd["processed_lidar"]
pfn = model.pillar_vfe.pfn_layers[0]
bn = (pfn.norm.weight, pfn.norm.bias, pfn.norm.running_mean, pfn.norm.running_var)

def timed(fn, n):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3

res = {}
res["pillar_vfe_scatter_us"] = timed(lambda: ops.pillar_vfe_scatter(pl["voxel_features"], pl["voxel_num_points"], pl["voxel_coords"], pfn.linear.weight, None, bn, 1e-3, True, False, h["model"]["args"]["voxel_size"], h["model"]["args"]["lidar_range"][:3], N, 200, 704), iters)
res["pillar_vfe_scatter_nhwc_us"] = timed(lambda: ops.pillar_vfe_scatter(pl["voxel_features"], pl["voxel_num_points"], pl["voxel_coords"], pfn.linear.weight, None, bn, 1e-3, True, False, h["model"]["args"]["voxel_size"], h["model"]["args"]["lidar_range"][:3], N, 200, 704, channels_last=True), iters)
for x in xs:
    res[f"warp_fuse_att_C{x.shape[1]}_us"] = timed(lambda: ops.warp_fuse(x, theta, [N], ops.FUSE_ATT), iters)
xcl = [x.contiguous(memory_format=torch.channels_last) for x in xs]
res["warp_fuse_nhwc_3scales_us"] = timed(lambda: ops.warp_fuse_nhwc(xcl, theta, ops.FUSE_ATT), iters)
for x in xcl:
    res[f"warp_fuse_nhwc_C{x.shape[1]}_us"] = timed(lambda: ops.warp_fuse_nhwc([x], theta, ops.FUSE_ATT), iters)
# the matrix-core convolution at the stage-1 shape (64 -> 64 channels, 100 x 352, N agents), with residual + ReLU
wconv = torch.randn(64, 64, 3, 3, generator=g).to(dev) / 24.0
wp, bconv, rconv = ops.pack_conv3x3_weight(wconv), torch.randn(64, generator=g).to(dev), torch.randn(N, 64, 100, 352, generator=g).to(dev)
res["conv3x3_bias_act_64ch_us"] = timed(lambda: ops.conv3x3_bias_act(xs[0], wp, bconv, rconv, True), iters)
# the opt-in split-bf16 convolution at the same shape (3-way and 2-way split)
for terms in (3, 2):
    wsplit = ops.pack_conv3x3_emu_weight(wconv, terms)
    res[f"conv3x3_emu_bf16x{terms}_64ch_us"] = timed(lambda: ops.conv3x3_emu_bias_act(xs[0], wsplit, bconv, 64, rconv, True, terms), iters)
# the pointwise kernel at the 4x4 up-sampling head shape (256 -> 128 channels, 25 x 88 -> 100 x 352, ego only)
wt = torch.randn(256, 128, 4, 4, generator=g).to(dev) / 16.0
res["pointwise_up4_us"] = timed(lambda: ops.pointwise_conv(xs[2][:1], ops.pack_pointwise_weight(wt, True), bconv.repeat(2), 128, up=4), iters)
print(json.dumps(res))
