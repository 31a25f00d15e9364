"""Launch only the hand-written ops on bench-shaped data (for rocprofv3 --pmc passes and quick A/B timing).
usage: python tools/kernels_only.py [iters] [op ...]      ops: see OPS below (default: all)
Every selected op is called exactly ``iters + 1`` times (one untimed), so per-op counter totals divide cleanly."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from coalign_amd import ops
from coalign_amd.config import builtin_config
from coalign_amd.detector import build_model, to_device
from coalign_amd.pose import normalize_pairwise_tfm
from coalign_amd.synthetic import fill_parameters_, make_frame

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 5
want = sys.argv[2:]
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
xcl = [x.contiguous(memory_format=torch.channels_last) for x in xs]
pl = frd["processed_lidar"]
pfn = model.pillar_vfe.pfn_layers[0]
bn = (pfn.norm.weight, pfn.norm.bias, pfn.norm.running_mean, pfn.norm.running_var)
margs = h["model"]["args"]
wconv = torch.randn(64, 64, 3, 3, generator=g).to(dev) / 24.0
wp, bconv, rconv = ops.pack_conv3x3_weight(wconv), torch.randn(64, generator=g).to(dev), torch.randn(N, 64, 100, 352, generator=g).to(dev)
wsplit = {t: ops.pack_conv3x3_emu_weight(wconv, t) for t in (3, 2)}                   # tap pairs: the strided layers, the 2-way split
wtapm = ops.pack_conv3x3_emu_weight(wconv, 3, True)                                   # tap-major: the detector's stride-1 layers (default)
canvas_cl = torch.randn(N, 64, 200, 704, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
wt = torch.randn(256, 128, 4, 4, generator=g).to(dev) / 16.0
wtp = ops.pack_pointwise_weight(wt, True)
wtp_emu = ops.pack_pointwise_emu_weight(wtp)
_pw = {}
for _tag, (_ci, _co, _up) in {"up1": (64, 128, 1), "up2": (128, 128, 2), "skip1": (64, 64, 1), "skip2": (64, 128, 1), "skip3": (128, 256, 1)}.items():
    _w = torch.randn(_ci, _co, _up, _up, generator=g).to(dev) / _ci ** 0.5
    _f = ops.pack_pointwise_weight(_w, True)
    _pw[_tag] = (_f, ops.pack_pointwise_emu_weight(_f), torch.randn(_co, generator=g).to(dev), _co, _up)


def pillar(cl):
    return lambda: ops.pillar_vfe_scatter(pl["voxel_features"], pl["voxel_num_points"], pl["voxel_coords"], pfn.linear.weight, None, bn, 1e-3, True, False,
                                          margs["voxel_size"], margs["lidar_range"][:3], N, 200, 704, channels_last=cl)


# post-processing on ~600 candidates shaped like a calibrated frame's (clusters of overlapping boxes around 120 objects)
import numpy as np
from oracle import coalign_oracle as _oracle
_rs = np.random.RandomState(11)
_K = 600
_ctr = _rs.uniform([-120, -36], [120, 36], size=(120, 2))
_b7 = np.zeros((_K, 7), np.float32)
_pick = _rs.randint(0, 120, _K)
_b7[:, 0:2] = _ctr[_pick] + _rs.normal(0, 0.4, (_K, 2)); _b7[:, 2] = -1.0; _b7[:, 3] = 1.5; _b7[:, 4] = _rs.uniform(1.5, 2.0, _K)
_b7[:, 5] = _rs.uniform(3.5, 5.0, _K); _b7[:, 6] = _rs.uniform(-0.3, 0.3, _K) + (_pick % 2) * 1.57
_corners = _oracle.boxes_to_corners_3d(torch.from_numpy(_b7), "hwl").to(dev)
_scores = torch.from_numpy(_rs.uniform(0.2, 1, _K).astype(np.float32)).to(dev)
_valid = torch.ones(_K, dtype=torch.uint8, device=dev)
_L = ops.hip.lib()
_nws = torch.empty(_L.coalign_nms_rotated_workspace_bytes(_K, 1000), dtype=torch.uint8, device=dev)
_keep, _kc = torch.empty(1000, dtype=torch.int32, device=dev), torch.zeros(1, dtype=torch.int32, device=dev)
_oc, _os, _on = torch.empty(1000, 8, 3, device=dev), torch.empty(1000, device=dev), torch.zeros(1, dtype=torch.int32, device=dev)
_rng = [-140.8, -40, -3, 140.8, 40, 1]


def _nms_fused():
    ops.nms_rotated_gather(_corners, _scores, 0.15, 1000, _valid, None, _keep, _kc, _rng, _oc, _os, _on, _nws)


def _nms_two_calls():
    ops.nms_rotated_device(_corners, _scores, 0.15, 1000, valid=_valid, keep=_keep, keep_count=_kc, ws=_nws)
    ops.gather_in_range(_corners, _scores, _keep, _kc, _rng, _oc, _os, _on)


_cache = {}
# round 4: the fp16 2-way split (default arithmetic), the Winograd kernel (opt-in), the sparse canvas and its consumers, the merged 1x1 heads
_w16 = ops.pack_conv3x3_emu_weight(wconv, 16, True)
_w16_s2 = ops.pack_conv3x3_emu_weight(wconv, 16, False)
_w128 = torch.randn(128, 128, 3, 3, generator=g).to(dev) / 34.0
_w256 = torch.randn(256, 256, 3, 3, generator=g).to(dev) / 48.0
_w16_128, _w16_256 = ops.pack_conv3x3_emu_weight(_w128, 16, True), ops.pack_conv3x3_emu_weight(_w256, 16, True)
_b128, _b256 = torch.randn(128, generator=g).to(dev), torch.randn(256, generator=g).to(dev)
_r128, _r256 = torch.randn(N, 128, 50, 176, generator=g).to(dev), torch.randn(N, 256, 25, 88, generator=g).to(dev)
_wino = {}
def _wu(name, w):                                  # (the laboratory library is loaded only when a Winograd op is asked for)
    if name not in _wino:
        _wino[name] = ops.pack_conv3x3_wino_weight(w)
    return _wino[name]
_rcl64, _rcl256 = rconv.contiguous(memory_format=torch.channels_last), _r256.contiguous(memory_format=torch.channels_last)
_x256 = torch.randn(1, 256, 100, 352, generator=g).to(dev)
_whead = ops.pack_pointwise_emu_weight(ops.pack_pointwise_weight(torch.randn(20, 256, 1, 1, generator=g).to(dev) / 16.0, False))
_bhead = torch.randn(20, generator=g).to(dev)
_sc_cache = {}


def _sc():
    if "sc" not in _sc_cache:
        _sc_cache["sc"] = ops.pillar_encode_sparse(pl["voxel_features"], pl["voxel_num_points"], pl["voxel_coords"], pfn.linear.weight, None, bn, 1e-3, True,
                                                   margs["voxel_size"], margs["lidar_range"][:3], N, 200, 704, canvas_cache={})
    return _sc_cache["sc"]


_fold_cache = {}
def _folded():
    if "f" not in _fold_cache:
        _fold_cache["f"] = ops.pillar_fold_params(pfn.linear.weight, None, bn, 1e-3, True)
    return _fold_cache["f"]


# round 5: the producer-split convolutions on SplitMaps (csrc/conv3x3_sp.hip) and the layers that write the first SplitMap of a chain
_sp = {}
def _spx(i):
    if i not in _sp:
        _sp[i] = (ops.SplitMap.pack(torch.relu(xs[i])), ops.SplitMap.pack((rconv, _r128, _r256)[i]))
    return _sp[i]
_x256s = {}
def _x256sp():
    if "x" not in _x256s:
        _x256s["x"] = ops.SplitMap.pack(torch.relu(_x256))
    return _x256s["x"]
_x384 = torch.randn(1, 384, 100, 352, generator=g).to(dev)
def _x384sp():
    if "x384" not in _x256s:
        _x256s["x384"] = ops.SplitMap.pack(torch.relu(_x384))
    return _x256s["x384"]
def _headmap():
    if "heads" not in _x256s:
        _x256s["heads"] = ops.SplitMap.empty(1, 384, 100, 352, dev)
    return _x256s["heads"]
_w384 = ops.pack_conv3x3_emu_weight(torch.randn(256, 384, 3, 3, generator=g).to(dev) / 59.0, 16, True)
_w16_s2_128 = ops.pack_conv3x3_emu_weight(torch.randn(128, 64, 3, 3, generator=g).to(dev) / 24.0, 16, False)
# round 6: the strided layers on split operands (csrc/conv3x3_sp_s2.hip) and the three heads in one launch
_w16_tap64 = ops.pack_conv3x3_emu_weight(torch.randn(64, 64, 3, 3, generator=g).to(dev) / 24.0, 16, True)
_xs_s2a = ops.SplitMap.pack(torch.relu(torch.randn(5, 64, 100, 352, generator=g).to(dev)))
_w16_s2a = ops.pack_conv3x3_emu_weight(torch.randn(128, 64, 3, 3, generator=g).to(dev) / 24.0, 16, True)
_b_s2a = torch.randn(128, generator=g).to(dev)
_xs_s2b = ops.SplitMap.pack(torch.relu(torch.randn(5, 128, 50, 176, generator=g).to(dev)))
_w16_s2b = ops.pack_conv3x3_emu_weight(torch.randn(256, 128, 3, 3, generator=g).to(dev) / 34.0, 16, True)
_b_s2b = torch.randn(256, generator=g).to(dev)
_ws_64 = ops.pack_conv1x1_sp_weight(torch.randn(64, 64, 1, 1, generator=g).to(dev) / 8.0)          # the blocks' 1 x 1 / stride-2 skip convolutions ride as a tenth tap (9g)
_ws_s2a = ops.pack_conv1x1_sp_weight(torch.randn(128, 64, 1, 1, generator=g).to(dev) / 8.0)
_ws_s2b = ops.pack_conv1x1_sp_weight(torch.randn(256, 128, 1, 1, generator=g).to(dev) / 11.0)

OPS = {
    "conv_sp_64ch": lambda: ops.conv3x3_sp(_spx(0)[0], _w16, bconv, 64, _spx(0)[1], True, out_split=True),
    "conv_sp_128ch": lambda: ops.conv3x3_sp(_spx(1)[0], _w16_128, _b128, 128, _spx(1)[1], True, out_split=True),
    "conv_sp_256ch": lambda: ops.conv3x3_sp(_spx(2)[0], _w16_256, _b256, 256, _spx(2)[1], True, out_split=True),
    "conv_sp_256ch_nhwc_out": lambda: ops.conv3x3_sp(_spx(2)[0], _w16_256, _b256, 256, _spx(2)[1], True, out_split=False),
    "conv_sp_shrink2_256ch_100x352": lambda: ops.conv3x3_sp(_x256sp(), _w16_256, _b256, 256, None, True, out_split=False),
    "conv_sp_shrink1_384ch_100x352": lambda: ops.conv3x3_sp(_x384sp(), _w384, _b256, 256, None, True, out_split=True),
    "pointwise_up1_split_out": lambda: ops.pointwise_conv(xs[0][:1], _pw["up1"][1], _pw["up1"][2], 128, up=1, out=_headmap(), c_off=0),
    "pointwise_up2_split_out": lambda: ops.pointwise_conv(xs[1][:1], _pw["up2"][1], _pw["up2"][2], 128, up=2, out=_headmap(), c_off=128),
    "pointwise_heads_one_launch": lambda: ops.pointwise_heads_split([(xs[0][:1], _pw["up1"][1], _pw["up1"][2], 128, 1, 0), (xs[1][:1], _pw["up2"][1], _pw["up2"][2], 128, 2, 128),
                                                                     (xs[2][:1], wtp_emu, bconv.repeat(2), 128, 4, 256)], _headmap()),
    "pointwise_up4_split_out": lambda: ops.pointwise_conv(xs[2][:1], wtp_emu, bconv.repeat(2), 128, up=4, out=_headmap(), c_off=256),
    "conv_fp16x2_shrink1_384ch_split_out": lambda: ops.conv3x3_emu_bias_act(_x384, _w384, _b256, 256, None, True, 16, out_split=True),
    "conv_fp16x2_s2_64to128_nhwc_in_split_out": lambda: ops.conv3x3_emu_bias_act(xcl[0], _w16_s2_128, _b128, 128, None, True, 16, stride=2, out_split=True),
    "conv_sp_s2_sparse_canvas_with_row_pack": lambda: ops.conv3x3_sp_s2(_sc(), _w16_tap64, bconv, 64, True, w_skip=_ws_64),
    "conv_sp_s2_64to128": lambda: ops.conv3x3_sp_s2(_xs_s2a, _w16_s2a, _b_s2a, 128, True, w_skip=_ws_s2a),
    "conv_sp_s2_128to256": lambda: ops.conv3x3_sp_s2(_xs_s2b, _w16_s2b, _b_s2b, 256, True, w_skip=_ws_s2b),
    "conv_sp_s2_64to128_without_skip_tap": lambda: ops.conv3x3_sp_s2(_xs_s2a, _w16_s2a, _b_s2a, 128, True),
    "conv_fp16x2_s2_sparse_canvas_split_out": lambda: ops.conv3x3_emu_sparse(_sc(), _w16_s2, bconv, 64, True, 16, out_channels_last=False, out_split=True),
    "nms_gather_K600": _nms_fused,
    "nms_then_gather_K600": _nms_two_calls,
    "pillar_nchw": pillar(False),
    "pillar_nhwc": pillar(True),
    "pillar_nhwc_persistent": lambda: ops.pillar_vfe_scatter(pl["voxel_features"], pl["voxel_num_points"], pl["voxel_coords"], pfn.linear.weight, None, bn, 1e-3, True, False,
                                                             margs["voxel_size"], margs["lidar_range"][:3], N, 200, 704, channels_last=True, canvas_cache=_cache),
    "pillar_sparse": lambda: ops.pillar_encode_sparse(pl["voxel_features"], pl["voxel_num_points"], pl["voxel_coords"], pfn.linear.weight, None, bn, 1e-3, True,
                                                      margs["voxel_size"], margs["lidar_range"][:3], N, 200, 704, canvas_cache=_cache, folded=_folded()),
    "conv_fp16x2_64ch": lambda: ops.conv3x3_emu_bias_act(xs[0], _w16, bconv, 64, rconv, True, 16),
    "conv_fp16x2_128ch": lambda: ops.conv3x3_emu_bias_act(xs[1], _w16_128, _b128, 128, _r128, True, 16),
    "conv_fp16x2_256ch": lambda: ops.conv3x3_emu_bias_act(xs[2], _w16_256, _b256, 256, _r256, True, 16),
    "conv_wino_bf16x3_64ch": lambda: ops.conv3x3_wino(xcl[0], _wu("64", wconv), bconv, 64, _rcl64, True),
    "conv_wino_bf16x3_256ch": lambda: ops.conv3x3_wino(xcl[2], _wu("256", _w256), _b256, 256, _rcl256, True),
    "conv_fp16x2_s2_sparse_canvas": lambda: ops.conv3x3_emu_sparse(_sc(), _w16_s2, bconv, 64, True, 16, out_channels_last=False),
    "pointwise_skip1_sparse_canvas": lambda: ops.pointwise_conv_sparse(_sc(), _pw["skip1"][1], _pw["skip1"][2], 64, False, out_channels_last=False),
    "heads_1x1_merged_bf16x3": lambda: ops.pointwise_conv(_x256, _whead, _bhead, 20, relu=False),
    "fuse_nchw_C64": lambda: ops.warp_fuse(xs[0], theta, [N], ops.FUSE_ATT),
    "fuse_nchw_C128": lambda: ops.warp_fuse(xs[1], theta, [N], ops.FUSE_ATT),
    "fuse_nchw_C256": lambda: ops.warp_fuse(xs[2], theta, [N], ops.FUSE_ATT),
    "fuse_nhwc_3scales": lambda: ops.warp_fuse_nhwc(xcl, theta, ops.FUSE_ATT),
    "conv_f32_64ch": lambda: ops.conv3x3_bias_act(xs[0], wp, bconv, rconv, True),
    "conv_bf16x3_64ch": lambda: ops.conv3x3_emu_bias_act(xs[0], wtapm, bconv, 64, rconv, True, 3),
    "conv_bf16x3_64ch_tap_pairs": lambda: ops.conv3x3_emu_bias_act(xs[0], wsplit[3], bconv, 64, rconv, True, 3),
    "conv_bf16x2_64ch": lambda: ops.conv3x3_emu_bias_act(xs[0], wsplit[2], bconv, 64, rconv, True, 2),
    "conv_bf16x3_64ch_nhwc_out": lambda: ops.conv3x3_emu_bias_act(xs[0], wtapm, bconv, 64, rconv, True, 3, out_channels_last=True),
    "conv_bf16x3_s2_canvas_nhwc_in": lambda: ops.conv3x3_emu_bias_act(canvas_cl, wsplit[3], bconv, 64, None, True, 3, stride=2),
    "pointwise_up4": lambda: ops.pointwise_conv(xs[2][:1], wtp, bconv.repeat(2), 128, up=4),
    "pointwise_up4_bf16x3": lambda: ops.pointwise_conv(xs[2][:1], wtp_emu, bconv.repeat(2), 128, up=4),
    "pointwise_up2": lambda: ops.pointwise_conv(xs[1][:1], _pw["up2"][0], _pw["up2"][2], 128, up=2),
    "pointwise_up2_bf16x3": lambda: ops.pointwise_conv(xs[1][:1], _pw["up2"][1], _pw["up2"][2], 128, up=2),
    "pointwise_up1": lambda: ops.pointwise_conv(xs[0][:1], _pw["up1"][0], _pw["up1"][2], 128, up=1),
    "pointwise_up1_bf16x3": lambda: ops.pointwise_conv(xs[0][:1], _pw["up1"][1], _pw["up1"][2], 128, up=1),
    "pointwise_skip1_canvas": lambda: ops.pointwise_conv(canvas_cl, _pw["skip1"][0], _pw["skip1"][2], 64, in_stride=2, relu=False),
    "pointwise_skip1_canvas_bf16x3": lambda: ops.pointwise_conv(canvas_cl, _pw["skip1"][1], _pw["skip1"][2], 64, in_stride=2, relu=False),
    "pointwise_skip2": lambda: ops.pointwise_conv(xcl[0], _pw["skip2"][0], _pw["skip2"][2], 128, in_stride=2, relu=False),
    "pointwise_skip2_bf16x3": lambda: ops.pointwise_conv(xcl[0], _pw["skip2"][1], _pw["skip2"][2], 128, in_stride=2, relu=False),
    "pointwise_skip3": lambda: ops.pointwise_conv(xcl[1], _pw["skip3"][0], _pw["skip3"][2], 256, in_stride=2, relu=False),
    "pointwise_skip3_bf16x3": lambda: ops.pointwise_conv(xcl[1], _pw["skip3"][1], _pw["skip3"][2], 256, in_stride=2, relu=False),
}


def timed(fn, n):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


res = {"iters": iters}
for name, fn in OPS.items():
    if want and name not in want:
        continue
    res[name + "_us"] = round(timed(fn, iters), 2)
print(json.dumps(res))
