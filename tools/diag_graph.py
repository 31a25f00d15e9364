import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from coalign_amd.config import builtin_config
from coalign_amd.detector import build_model, to_device
from coalign_amd.pipeline import FramePipeline
from coalign_amd.postprocess import build_postprocessor
from coalign_amd.synthetic import calibrate_heads_, fill_parameters_, make_frame
dev = torch.device("cuda:0")
h = builtin_config("opv2v_coalign")
frames = []
for i in range(8):
    f = to_device(make_frame(h, 5, pillars_per_agent=8000, seed=303 + i, noise=(0.2, 0.2)), dev); f["record_len"] = [5]; frames.append(f)
model = build_model(h); fill_parameters_(model, seed=0); model = model.to(dev).eval()
pp = build_postprocessor(h["postprocess"], False)
anchors = torch.from_numpy(pp.generate_anchor_box())
calibrate_heads_(model, frames[0], 0.2, 600)
meta = {"ego": {"transformation_matrix": torch.eye(4, device=dev), "anchor_box": anchors}}
with torch.no_grad():
    sync = [pp.post_process(meta, {"ego": model(f)}) for f in frames]
for graph in (True, False):
    pipe = FramePipeline(model, pp, anchors, lanes=4, result_lag=1, graph=graph)
    order = [0, 1, 2, 3] + [i % 8 for i in range(8)] + [i % 8 for i in range(40)]
    res = []
    for phase in (order[:4], order[4:12], order[12:]):          # bench.py's phases: capture, warm-up, timed loop, drained in between
        for i in phase:
            res += [(b, s) for _, b, s in pipe.submit(frames[i])]
        res += [(b, s) for _, b, s in pipe.drain()]
        torch.cuda.synchronize()
    bad = []
    for k, (i, (b, s)) in enumerate(zip(order, res)):
        sb, ss = sync[i]
        if b is None or b.shape != sb.shape or not torch.equal(b, sb) or not torch.equal(s, ss):
            n = 0 if b is None else min(b.shape[0], sb.shape[0])
            # which sync frame does the result equal (if any)?
            same_as = [j for j in range(8) if b is not None and b.shape == sync[j][0].shape and torch.equal(b, sync[j][0])]
            prefix = None if b is None else int((b[:n] == sb[:n]).all(dim=2).all(dim=1).sum())
            bad.append((k, i, None if b is None else b.shape[0], sb.shape[0], "equals frame %s" % same_as, "rows equal in prefix: %s" % prefix))
    print("graph" if graph else "eager", "mismatches:", bad[:12], len(bad))
