"""Layer-by-layer run-to-run determinism of ResNet stage 3 (256 channels, 25 x 88)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from coalign_amd import ops, backbone as bb
from coalign_amd.config import builtin_config
from coalign_amd.detector import build_model, to_device
from coalign_amd.synthetic import fill_parameters_, make_frame
dev = torch.device("cuda:0")
h = builtin_config("opv2v_coalign")
f = to_device(make_frame(h, 5, pillars_per_agent=8000, seed=303, noise=(0.2, 0.2)), dev); f["record_len"] = [5]
model = build_model(h); fill_parameters_(model, seed=0); model = model.to(dev).eval()
def same(a, b): return "same" if torch.equal(a, b) else f"DIFF n={int((a != b).sum())} max={float((a - b).abs().max()):.3e} scale={float(a.abs().max()):.3e}"
with torch.no_grad():
    for terms in (0, 3):
        bb.CONV_EMU_TERMS = terms
        feats, _ = model.encode(f)
        x1 = feats[1].clone()
        torch.cuda.synchronize()
        print("terms", terms, "feat1 again:", same(model.encode(f)[0][1], x1))
        layer = model.backbone.resnet.layer2
        blk = layer[0]
        w1, b1, w2, b2, wd, p1, p2, pd = blk._folded()
        for rep in range(2):
            a = F.conv2d(x1, w1, None, 2, 1); b = F.conv2d(x1, w1, None, 2, 1); torch.cuda.synchronize()
            print(" miopen s2 conv 128->256:", same(a, b))
            a = ops.pointwise_conv(x1, pd[0], pd[1], wd.shape[0], in_stride=2, relu=False); b = ops.pointwise_conv(x1, pd[0], pd[1], wd.shape[0], in_stride=2, relu=False); torch.cuda.synchronize()
            print(" pointwise s2 128->256:", same(a, b))
            y = ops.bias_act_(F.conv2d(x1, w1, None, 2, 1), b1, None, True)
            a = bb.conv3x3_fused(y, p2, w2, b2, None); b = bb.conv3x3_fused(y, p2, w2, b2, None); torch.cuda.synchronize()
            print(" conv2 256->256 (policy route):", same(a, b))
            a = F.conv2d(y, w2, None, 1, 1); b = F.conv2d(y, w2, None, 1, 1); torch.cuda.synchronize()
            print(" miopen conv 256->256:", same(a, b))
            a = blk(x1); b = blk(x1); torch.cuda.synchronize()
            print(" block0:", same(a, b))
            a = layer(x1); b = layer(x1); torch.cuda.synchronize()
            print(" layer2:", same(a, b))
