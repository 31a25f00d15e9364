"""Round-2 diagnostics on the GPU: (A) run-to-run determinism of every stage, (B) pipeline throughput eager vs graph."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from coalign_amd import ops, backbone as bb
from coalign_amd.config import builtin_config
from coalign_amd.detector import build_model, to_device
from coalign_amd.pipeline import FramePipeline
from coalign_amd.postprocess import build_postprocessor
from coalign_amd.synthetic import fill_parameters_, make_frame
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(__file__), "..", "bench.py"))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)

dev = torch.device("cuda:0")
h = builtin_config("opv2v_coalign")
frames = []
for i in range(4):
    f = to_device(make_frame(h, 5, pillars_per_agent=8000, seed=303 + i, noise=(0.2, 0.2)), dev); f["record_len"] = [5]; frames.append(f)
model = build_model(h); fill_parameters_(model, seed=0); model = model.to(dev).eval()
pp = build_postprocessor(h["postprocess"], False)
anchors = torch.from_numpy(pp.generate_anchor_box())
from coalign_amd.synthetic import calibrate_heads_
calibrate_heads_(model, frames[0], 0.2, 600)
meta = {"ego": {"transformation_matrix": torch.eye(4, device=dev), "anchor_box": anchors}}
which = sys.argv[1] if len(sys.argv) > 1 else "AB"

def stages(f):
    with torch.no_grad():
        bd = {"voxel_features": f["processed_lidar"]["voxel_features"], "voxel_coords": f["processed_lidar"]["voxel_coords"],
              "voxel_num_points": f["processed_lidar"]["voxel_num_points"], "record_len": [5]}
        bd = model.scatter(model.pillar_vfe(bd))
        canvas = bd["spatial_features"]
        feats, aff = model.encode(f)
        fused = model._fuse_scales(list(feats), [5], aff)
        out = model.fuse_and_head(list(feats), [5], aff)
        b, s = pp.post_process(meta, {"ego": out})
    torch.cuda.synchronize()
    return {"pillar_features": bd["pillar_features"], "canvas": canvas, "feat0": feats[0], "feat1": feats[1], "feat2": feats[2],
            "fused0": fused[0], "fused1": fused[1], "fused2": fused[2], "cls": out["cls_preds"], "reg": out["reg_preds"], "dir": out["dir_preds"],
            "boxes": b, "scores": s}

if "A" in which:
    for terms in (0, 3):
        bb.CONV_EMU_TERMS = terms
        for fi in (0, 1):
            ref = stages(frames[fi])
            for rep in range(3):
                cur = stages(frames[fi])
                bad = []
                for k in ref:
                    if ref[k].shape != cur[k].shape: bad.append((k, "shape", tuple(ref[k].shape), tuple(cur[k].shape)))
                    elif not torch.equal(ref[k], cur[k]): bad.append((k, int((ref[k] != cur[k]).sum()), float((ref[k] - cur[k]).abs().max())))
                print(f"terms {terms} frame {fi} rep {rep}: {'deterministic' if not bad else bad}", flush=True)
    bb.CONV_EMU_TERMS = 0

if "B" in which:
    for terms in (0, 3):
        bb.CONV_EMU_TERMS = terms
        for graph in (False, True):
            for lanes in (1, 4):
                pipe = FramePipeline(model, pp, anchors, lanes=lanes, result_lag=1, graph=graph)
                for s in range(3 * lanes): pipe.submit(frames[s % 4])
                pipe.drain(); torch.cuda.synchronize()
                pipe.host_enqueue_s = 0.0
                t0 = time.perf_counter(); n = 40
                for s in range(n): pipe.submit(frames[s % 4])
                pipe.drain(); torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                print(json.dumps({"terms": terms, "graph": graph, "lanes": lanes, "fps": round(n / dt, 1), "host_enqueue_ms": round(pipe.host_enqueue_s / n * 1e3, 3)}), flush=True)
                del pipe
