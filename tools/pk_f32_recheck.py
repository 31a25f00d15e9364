#!/usr/bin/env python
"""Round-3 finding re-examined: "packed fp32 instructions of a wavefront that shares a SIMD with the three matrix
wavefronts of the 6 x 32 stacked convolution return wrong lanes 48-63".  The same commit that removed packed fp32 from the library also added the missing
"m0" clobber to the inline-assembly LDS-DMA -- the likelier cause of cross-kernel corruption.  This script separates the two: it runs the fusion kernel
(warp_fuse_nhwc, the victim) beside each convolution geometry (the aggressor, COALIGN_EMU_STACK = 3 selects the 6 x 32 / 32-channel variant the finding was
made with) in three library builds and counts fused maps that differ from the map computed alone:
   lab      -DCOALIGN_LAB, no packed fp32 (the product's flags)                                       expected 0
   labvec   -DCOALIGN_LAB, packed fp32 allowed again (v_pk_mul_f32 / v_pk_add_f32 back in the fusion kernel), m0 clobber in place
If `labvec` shows 0 differing maps as well, the packed-fp32 hypothesis is not supported and the flags are a precaution, not a fix.
Output: one JSON line per (build, stack) + the packed instruction counts of the fusion kernel in both builds."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import sys, torch
from coalign_amd import ops
from coalign_amd.config import builtin_config
from coalign_amd.pose import normalize_pairwise_tfm
from coalign_amd.synthetic import make_frame
g = torch.Generator().manual_seed(3)
N = 5
fr = make_frame(builtin_config("opv2v_coalign"), N, pillars_per_agent=100, seed=303, noise=(0.2, 0.2))
theta = normalize_pairwise_tfm(fr["pairwise_t_matrix"].cuda(), 200, 704, 0.4)[0, 0, :N].contiguous()
xcl = [torch.randn(N, C, H, W, generator=g).cuda().contiguous(memory_format=torch.channels_last) for C, H, W in ((64, 100, 352), (128, 50, 176), (256, 25, 88))]
def conv(N_, C, H, W):
    x = torch.randn(N_, C, H, W, generator=g).cuda(); w = ops.pack_conv3x3_emu_weight((torch.randn(C, C, 3, 3, generator=g) / (C * 9) ** 0.5).cuda(), 3, True)
    b = torch.randn(C, generator=g).cuda(); r = torch.randn(N_, C, H, W, generator=g).cuda()
    return lambda: ops.conv3x3_emu_bias_act(x, w, b, C, r, True, 3)
side = torch.cuda.Stream()
ref = [t.clone() for t in ops.warp_fuse_nhwc(xcl, theta, ops.FUSE_ATT)]
bad = torch.zeros((), dtype=torch.int64, device="cuda")
runs = 0
for fn in (conv(5, 256, 25, 88), conv(5, 128, 50, 176), conv(5, 64, 100, 352)):
    for it in range(300):
        with torch.cuda.stream(side):
            fn()
        for a, b in zip(ops.warp_fuse_nhwc(xcl, theta, ops.FUSE_ATT), ref):
            bad += (a != b).any()
            runs += 1
torch.cuda.synchronize()
print("RESULT", int(bad), runs)
"""
out = []
for lab in ("1", "vec"):
    for stack in ("3", "5"):
        r = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, PYTHONPATH=ROOT, COALIGN_LAB=lab, COALIGN_EMU_STACK=stack), capture_output=True, text=True, timeout=600, cwd=ROOT)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
        row = {"build": "lab (no packed fp32)" if lab == "1" else "labvec (packed fp32 allowed)", "COALIGN_EMU_STACK": stack,
               "fused_maps_differing": int(line[0].split()[1]) if line else None, "fused_maps_checked": int(line[0].split()[2]) if line else None,
               "error": None if line else r.stderr[-300:]}
        out.append(row)
        print(json.dumps(row), flush=True)
sys.path.insert(0, ROOT)
from coalign_amd import build
import re, tempfile
def pk_count(src, vec):
    flags = [f for f in build.FLAGS if not (vec and f in ("-fno-slp-vectorize", "-fno-vectorize"))]
    with tempfile.TemporaryDirectory() as tmp:
        o = os.path.join(tmp, "k.s")
        subprocess.run([build._hipcc(), "-x", "hip"] + flags + ["--cuda-device-only", "-S", os.path.join(build.CSRC, src), "-o", o], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        return len(re.findall(r"\bv_pk_(?:add|mul|fma)_f32\b", open(o).read()))
print(json.dumps({"packed_fp32_instructions_in_warp_fuse_nhwc": {"product flags": pk_count("warp_fuse_nhwc.hip", False), "vectoriser on": pk_count("warp_fuse_nhwc.hip", True)}}))
