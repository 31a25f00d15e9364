"""Which stage loses bit-equality when several frames are in flight?  encode / tail / post-process of 4 frames on 4 streams against the serial results."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from coalign_amd.config import builtin_config
from coalign_amd.detector import build_model, to_device
from coalign_amd.postprocess import build_postprocessor
from coalign_amd.synthetic import calibrate_heads_, fill_parameters_, make_frame
dev = torch.device("cuda:0")
h = builtin_config("opv2v_coalign")
frames = []
for i in range(4):
    d = to_device(make_frame(h, 5, pillars_per_agent=8000, seed=303 + i, noise=(0.2, 0.2)), dev); d["record_len"] = [5]; frames.append(d)
model = build_model(h); fill_parameters_(model, seed=0); model = model.to(dev).eval()
pp = build_postprocessor(h["postprocess"], False); pp.buffer_sets = 8
calibrate_heads_(model, frames[0], 0.2, 600)
anchors = torch.from_numpy(pp.generate_anchor_box())
meta = {"ego": {"transformation_matrix": torch.eye(4, device=dev), "anchor_box": anchors}}
persist = os.environ.get("PERSIST", "1") == "1"
model.pillar_vfe.persistent_canvas = persist
with torch.no_grad():
    ser = []
    for f in frames:
        feats, aff = model.encode(f)
        feats = [x.clone() for x in feats]
        out = model.fuse_and_head(feats, [5], aff)
        out = {k: v.clone() for k, v in out.items()}
        ser.append((feats, aff, out, pp.post_process(meta, {"ego": out})))
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(4)]
    bad = {"encode": 0, "tail": 0, "post": 0}
    for rep in range(6):
        res = [None] * 4
        for k, s in enumerate(streams):
            with torch.cuda.stream(s):
                feats, aff = model.encode(frames[k])
                res[k] = ([x.clone() for x in feats], aff)
        torch.cuda.synchronize()
        for k in range(4):
            if not all(torch.equal(a, b) for a, b in zip(res[k][0], ser[k][0])):
                bad["encode"] += 1
                for i, (a, b) in enumerate(zip(res[k][0], ser[k][0])):
                    if not torch.equal(a, b):
                        print(f"rep {rep} frame {k}: scale {i} differs at {int((a != b).sum())} elements, max {float((a - b).abs().max()):.3e}")
        outs = [None] * 4
        for k, s in enumerate(streams):
            with torch.cuda.stream(s):
                outs[k] = model.fuse_and_head([x.clone() for x in ser[k][0]], [5], ser[k][1])
        torch.cuda.synchronize()
        for k in range(4):
            if not all(torch.equal(outs[k][n], ser[k][2][n]) for n in outs[k]):
                bad["tail"] += 1
        posts = [None] * 4
        for k, s in enumerate(streams):
            with torch.cuda.stream(s):
                posts[k] = pp.post_process_async(meta, {"ego": ser[k][2]}, side_stream=False)
        for k in range(4):
            b, sc = posts[k].result()
            if not (torch.equal(b, ser[k][3][0]) and torch.equal(sc, ser[k][3][1])):
                bad["post"] += 1
    print("mismatches", bad, "persistent canvas", persist)
