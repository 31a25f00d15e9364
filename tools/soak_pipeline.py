"""Soak: FRAMES frames through the 3-lane graph pipeline (the bench configuration) from pillars, every result compared bit for bit with the
synchronous single-stream path.  Run with the convolution variant under test (COALIGN_EMU_STACK=...)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from coalign_amd.config import builtin_config
from coalign_amd.detector import build_model, to_device
from coalign_amd.pipeline import FramePipeline
from coalign_amd.postprocess import build_postprocessor
from coalign_amd.synthetic import calibrate_heads_, fill_parameters_, make_frame
dev = torch.device("cuda:0")
FRAMES, POOL, LANES = int(os.environ.get("FRAMES", 3000)), 8, int(os.environ.get("LANES", 3))
h = builtin_config("opv2v_coalign")
frames = []
for i in range(POOL):
    c = make_frame(h, 5, pillars_per_agent=8000, seed=303 + i, noise=(0.2, 0.2))
    d = to_device(c, dev); d["record_len"] = [5]
    if os.environ.get("HOST_POSES", "1") == "1":      # the bench's form: the dataset's host copy of the pose matrices travels with the batch (frame record + host-normalised poses)
        d["pairwise_t_matrix_host"] = c["pairwise_t_matrix"]
    frames.append(d)
model = build_model(h); fill_parameters_(model, seed=0); model = model.to(dev).eval()
pp = build_postprocessor(h["postprocess"], False)
calibrate_heads_(model, frames[0], 0.2, 600)
anchors = torch.from_numpy(pp.generate_anchor_box())
meta = {"ego": {"transformation_matrix": torch.eye(4, device=dev), "anchor_box": anchors}}
with torch.no_grad():
    sync = [pp.post_process(meta, {"ego": model(f)}) for f in frames]
    sync = [(b.clone(), s.clone()) for b, s in sync]
DEPTH = int(os.environ.get("DEPTH", 1))             # LANES=2 DEPTH=3: the bench's throughput configuration (frames queued per stream)
pipe = FramePipeline(model, pp, anchors, lanes=LANES, queue_depth=DEPTH, result_lag=LANES * DEPTH - 1 if DEPTH > 1 else 1, graph=os.environ.get("GRAPH", "1") == "1")
bad, n = [], 0
t0 = time.time()
def check(results):
    global n
    for idx, b, s in results:
        want = sync[idx % POOL]
        if not (torch.equal(b, want[0]) and torch.equal(s, want[1])):
            bad.append(idx)
        n += 1
for i in range(FRAMES):
    check(pipe.submit(frames[i % POOL]))
check(pipe.drain())
torch.cuda.synchronize()
print(f"soak: {n} frames in {time.time() - t0:.1f} s, streams {LANES} x {DEPTH} queued, frames read in place {pipe.frames_in_place}, copied {pipe.frames_copied}, graphs captured {pipe.graphs_captured}, "
      f"split-map range exceeded {pipe.range_exceeded()}: {len(bad)} frames differ from the synchronous path {bad[:10]}")
pipe.close()
