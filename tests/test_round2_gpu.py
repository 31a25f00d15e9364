"""GPU parity tests of the round-2 additions: NaiveCompressor against the reference golden (row D), the device bitmask NMS with
OpenPCDet semantics (row N), the fusion kernel's agent row table (agent-sharded exchange), the strided-plane epilogue."""
import copy

import numpy as np
import pytest
import torch

from oracle import coalign_oracle as oracle
from coalign_amd import ops
from coalign_amd.config import builtin_config
from coalign_amd.detector import build_model, to_device
from coalign_amd.synthetic import fill_parameters_

pytestmark = pytest.mark.gpu
T = torch.from_numpy
DEV = "cuda:0"


def rel_err(got, ref):
    got = got.detach().float().cpu()
    ref = (ref if torch.is_tensor(ref) else T(np.asarray(ref))).float()
    return float((got - ref).abs().max()) / max(float(ref.abs().max()), 1e-30)


# ------------------------------------------------------------------------------------------------ row D
def test_naive_compressor_vs_reference_golden(golden):
    """NaiveCompressor (naive_compress.py:5-31) on the device (BN folded, fused epilogue) against the reference module's own
    outputs (randomised BN statistics), and the CoAlign model built from a config with ``compression: 4`` against the reference
    model built from the same yaml key (point_pillar_baseline_multiscale.py:50-53,113-114)."""
    from coalign_amd.backbone import NaiveCompressor
    g = golden("naive_compress.npz")
    x = T(g["x"]).to(DEV)
    for ratio in (2, 8):
        m = NaiveCompressor(64, ratio).eval()
        fill_parameters_(m, seed=40 + ratio)
        with torch.no_grad():
            y = m.to(DEV)(x)
        e = rel_err(y, g[f"y_r{ratio}"])
        print(f"NaiveCompressor ratio {ratio}: {e:.2e}")
        assert e < 1e-4
    h = copy.deepcopy(builtin_config("mini_coalign"))
    h["model"]["args"]["compression"] = int(g["model_ratio"])
    model = build_model(h)
    fill_parameters_(model, seed=0, cls_bias=-1.0)
    model = model.to(DEV).eval()
    gm = golden("model_mini.npz")
    batch = {"processed_lidar": {"voxel_features": T(gm["voxel_features"]), "voxel_coords": T(gm["voxel_coords"]),
                                 "voxel_num_points": T(gm["voxel_num_points"])},
             "record_len": T(gm["record_len"]), "pairwise_t_matrix": T(gm["pairwise_t_matrix"])}
    with torch.no_grad():
        out = model(to_device(batch, DEV))
    for k in ("cls_preds", "reg_preds", "dir_preds"):
        e = rel_err(out[k], g[k])
        print(f"compression model {k}: {e:.2e}")
        assert e < 1e-4, k


# ------------------------------------------------------------------------------------------------ row N
def _random_boxes(n, seed, spread=60.0, size=1.0):
    rs = np.random.RandomState(seed)
    b = np.zeros((n, 7), dtype=np.float32)
    b[:, 0] = rs.uniform(-spread, spread, n); b[:, 1] = rs.uniform(-spread / 2, spread / 2, n); b[:, 2] = rs.uniform(-1.2, -0.8, n)
    b[:, 3] = rs.uniform(3.5, 5, n) * size; b[:, 4] = rs.uniform(1.6, 2.1, n) * size; b[:, 5] = rs.uniform(1.4, 1.8, n); b[:, 6] = rs.uniform(-3.1, 3.1, n)
    s = rs.uniform(0, 1, n).astype(np.float32)
    s[::97] = s[1::97][: len(s[::97])]                      # some exactly equal scores: ties keep input order (stable sort)
    return b, s


def _safe_threshold(boxes, start, normal):
    """A threshold no pair's IoU comes within 1e-5 of: the device's sinf / cosf / atan2f differ from glibc's in the last bit
    (a corner moves by at most one ulp of a <= 60 m coordinate, an IoU by ~1e-6), so a pair sitting on the threshold could
    legitimately flip (the oracle for this row is a restatement of a CUDA extension that cannot be built here -- parity unpinned,
    see DESIGN.md -- and the comparison is made robust instead of lucky)."""
    for k in range(60):
        thr = start + 0.003 * k
        if oracle.pcdet_min_margin(boxes, thr, normal) > 1e-5:
            return thr
    raise AssertionError("no safe threshold found")


@pytest.mark.parametrize("n,normal", [(4096, False), (4096, True), (1000, False), (65, False), (6000, False)])
def test_pcdet_bitmask_nms_vs_oracle(n, normal):
    """nms_gpu / nms_normal_gpu (iou3d_nms_utils.py:255-289, kernels iou3d_nms_kernel.cu:267-372) entirely on the device: keep
    lists bit-equal to the oracle's greedy walk at K = 4096 (64 x 64 mask words), beyond one word per lane (6000), with ties."""
    from coalign_amd import pcdet
    # >= 4096 boxes: pedestrian-sized, so that the +-60 m scene holds a few thousand overlapping pairs, not a few hundred thousand
    boxes, scores = _random_boxes(n, seed=100 + n + int(normal), spread=60.0 if n >= 4096 else 35.0 * (n / 1000.0) ** 0.5 + 4.0, size=0.3 if n >= 4096 else 1.0)
    order = np.argsort(-scores, kind="stable")
    thr = _safe_threshold(boxes[order], 0.1 if n != 65 else 0.01, normal)
    bd, sd = T(boxes).to(DEV), T(scores).to(DEV)
    fn = pcdet.nms_normal_gpu if normal else pcdet.nms_gpu
    keep, none = fn(bd, sd, thr)
    want = (oracle.pcdet_nms_normal if normal else oracle.pcdet_nms)(boxes, scores, thr)
    assert none is None and keep.dtype == torch.int64 and keep.is_cuda
    assert keep.cpu().numpy().tolist() == want.tolist()
    assert 0.02 * n < len(want) < 0.99 * n                 # real suppression happened, and real survivors remain
    if not normal:
        pre = n // 2
        keep2, _ = pcdet.nms_gpu(bd, sd, thr, pre_maxsize=pre)
        assert keep2.cpu().numpy().tolist() == oracle.pcdet_nms(boxes, scores, thr, pre).tolist()


def test_pcdet_nms_edge_cases():
    from coalign_amd import pcdet
    empty = torch.zeros((0, 7), device=DEV)
    keep, _ = pcdet.nms_gpu(empty, torch.zeros(0, device=DEV), 0.1)
    assert keep.numel() == 0
    one = torch.tensor([[0., 0, 0, 4, 2, 1.5, 0.3]], device=DEV)
    assert pcdet.nms_gpu(one, torch.tensor([0.5], device=DEV), 0.1)[0].tolist() == [0]
    dup = one.repeat(70, 1)                                  # identical boxes: only the best-scored survives
    sc = torch.linspace(0.1, 0.9, 70, device=DEV)
    assert pcdet.nms_gpu(dup, sc, 0.5)[0].tolist() == [69]
    assert pcdet.nms_normal_gpu(dup, sc, 0.5)[0].tolist() == [69]
    far = one.repeat(130, 1)
    far[:, 0] = torch.arange(130, device=DEV) * 10.0         # disjoint boxes: everything survives, in score order
    sc = torch.rand(130, device=DEV)
    assert pcdet.nms_gpu(far, sc, 0.01)[0].tolist() == sc.sort(dim=0, descending=True, stable=True)[1].tolist()


# ------------------------------------------------------------------------------------------------ agent row table
@pytest.mark.parametrize("mode", [ops.FUSE_ATT, ops.FUSE_MAX, ops.FUSE_NONE])
@pytest.mark.parametrize("n,C,H,W", [(5, 64, 100, 352), (3, 128, 50, 176), (5, 256, 25, 88), (2, 64, 50, 126), (8, 64, 60, 60)])
def test_warp_fuse_row_table_equals_reordered_input(mode, n, C, H, W):
    """coalign_warp_fuse_rows: agents stored in another order + the row table == the same call on agent-ordered input, bit for bit
    (what makes the sharded run reproduce the 1-GPU run), on the staged route, the generic route (W % 4 != 0) and 8 agents."""
    gen = torch.Generator().manual_seed(n * 1000 + C)
    x = torch.randn(n, C, H, W, generator=gen).to(DEV)
    ang = torch.rand(n, generator=gen) * 0.8 - 0.4
    theta = torch.zeros(n, 2, 3, dtype=torch.float64)
    theta[:, 0, 0] = torch.cos(ang); theta[:, 0, 1] = -torch.sin(ang) * H / W
    theta[:, 1, 0] = torch.sin(ang) * W / H; theta[:, 1, 1] = torch.cos(ang)
    theta[:, :, 2] = (torch.rand(n, 2, generator=gen) - 0.5) * 0.3
    theta[0] = torch.tensor([[1, 0, 0], [0, 1, 0]], dtype=torch.float64)
    theta = theta.to(DEV)
    want = ops.warp_fuse(x, theta, [n], mode)
    perm = torch.randperm(n, generator=gen).tolist()          # physical row r holds logical agent perm[r]
    rows = [0] * n
    for r, a in enumerate(perm):
        rows[a] = r
    got = ops.warp_fuse(x[perm].contiguous(), theta, [n], mode, rows=rows)
    assert torch.equal(got, want)
    with pytest.raises(Exception):
        ops.warp_fuse(x, theta, [n], mode, rows=[0] * n if n > 1 else [1])       # not a permutation


def test_bias_act_many_planes():
    """N * C beyond HIP's grid.y limit of 65535 (batched stage-1 precompute): the kernel strides over the planes."""
    N, C, HW = 180, 384, 8
    y = torch.randn(N, C, 2, 4, device=DEV)
    b = torch.randn(C, device=DEV)
    r = torch.randn(N, C, 2, 4, device=DEV)
    want = torch.relu(y + b.view(1, C, 1, 1) + r)
    got = ops.bias_act_(y.clone(), b, r, True)
    assert torch.equal(got, want)


# ------------------------------------------------------------------------------------------------ strided / channels-last convolution
@pytest.mark.parametrize("terms,tol", [(3, 5e-6), (2, 2e-5)])
@pytest.mark.parametrize("shape", [(5, 64, 64, 200, 704), (3, 64, 128, 100, 352), (2, 128, 256, 50, 176), (2, 64, 64, 37, 53), (1, 8, 64, 5, 4), (1, 16, 64, 1, 1), (2, 64, 64, 200, 504)])
def test_conv3x3_emu_stride2_and_layouts_vs_fp64(shape, terms, tol):
    """coalign_conv3x3_emu_ex: the stride-2 first convolution of every ResNet stage (resblock.py:150-174; odd and even map sizes,
    the 704 x 200 canvas) against the fp64 convolution, from NCHW and from channels-last input (bit-identical to each other), and the
    channels-last OUTPUT variant of the stride-1 kernel (bit-identical values to the NCHW one)."""
    import torch.nn.functional as F
    N, Ci, Co, H, W = shape
    gen = torch.Generator(device="cpu").manual_seed(sum(shape) + terms)
    x = torch.randn(N, Ci, H, W, generator=gen).to(DEV)
    w = (torch.randn(Co, Ci, 3, 3, generator=gen) / (Ci * 9) ** 0.5).to(DEV)
    b = torch.randn(Co, generator=gen).to(DEV)
    ws = ops.pack_conv3x3_emu_weight(w, terms)
    want = torch.relu(F.conv2d(x.double(), w.double(), b.double(), stride=2, padding=1))
    got = ops.conv3x3_emu_bias_act(x, ws, b, Co, None, True, terms, stride=2)
    assert got.shape == want.shape and got.is_contiguous()
    assert float((got.double() - want).abs().max()) <= tol * float(want.abs().max())
    xcl = x.contiguous(memory_format=torch.channels_last)
    if ops.is_channels_last(xcl):
        got_cl = ops.conv3x3_emu_bias_act(xcl, ws, b, Co, None, True, terms, stride=2)
        assert got_cl.is_contiguous() and torch.equal(got_cl, got)
    nor = ops.conv3x3_emu_bias_act(x, ws, b, Co, None, False, terms, stride=2)
    assert float((nor.double() - F.conv2d(x.double(), w.double(), b.double(), stride=2, padding=1)).abs().max()) <= tol * float(want.abs().max())
    if Ci == Co:                                             # stride 1, channels-last output, with residual
        r = torch.randn(N, Co, H, W, generator=gen).to(DEV)
        y_nchw = ops.conv3x3_emu_bias_act(x, ws, b, Co, r, True, terms)
        y_cl = ops.conv3x3_emu_bias_act(x, ws, b, Co, r, True, terms, out_channels_last=True)
        assert y_cl.shape == y_nchw.shape and (ops.is_channels_last(y_cl) or H * W == 1)
        assert torch.equal(y_cl.contiguous(), y_nchw)
