#!/usr/bin/env python
"""Benchmark of the CoAlign per-frame detection hot path on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): frames/s on synthetic 5-agent OPV2V-shaped scenes.  A step is one pass of the whole hot
path (pillar encode + scatter -> BEV backbone -> pose-aware warp + attention fusion at 3 scales -> heads ->
decode + rotated NMS) over one frame per rank; inputs are resident in HBM before the timed region and the timed loop
ROTATES over a pool of distinct frames.  The loop is ``coalign_amd.pipeline.FramePipeline`` -- the product's frame runner
(single GPU: two HIP streams with three frames queued on each, one HIP graph replay per frame; with R > 1 ranks three streams, one frame each, two replays --
encoder, ego tail -- around the lane's collective), the same object the parity tests drive.  ``python bench.py --gpus N`` without a launcher starts its N ranks itself (torch.distributed.run,
127.0.0.1) and refuses to run when fewer than N devices are visible.  With R ranks a step
processes R frames in the agent-sharded "frame ring" of coalign_amd/sharded.py (weak scaling): every rank encodes the
agents the ring assigns to it out of the SAME frame pool, so the per-frame detection checksums printed here are equal
for every --gpus value.  Rank 0 prints ONE JSON line; it carries the roofline of the dominant hand-written kernel, the
north-star HBM figure of the pillar-scatter + warp path, and, at N=1, the CPU oracle timed on the host cores.
"""
import argparse
import json
import os
import sys
import time

# Multi-rank runs: lanes + the collective backend's streams (+ the post-processing side streams of an eager fall-back) are more HIP streams than the runtime's
# default four hardware queues, and streams that share a queue serialise (498 -> 514 frames/s for four eager lanes on one GPU with eight
# queues; three graph lanes -- the single-GPU default -- are not affected and keep the runtime's default).  Must be set before the HIP runtime starts.
if int(os.environ.get("WORLD_SIZE", "1")) > 1:
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def self_launch(n: int) -> None:
    """``python bench.py --gpus N`` (N > 1) without a launcher: become ``torch.distributed.run --nproc-per-node N`` of this very command line (the reference's
    own bring-up is a launcher too: opencood/tools/multi_gpu_utils.py:31-37).  Never returns.  Fewer than N visible devices: exit 2 with the reason -- an
    N = 1 line for an N = 8 request would be worse than no line."""
    import socket
    import torch
    one_gpu = os.environ.get("COALIGN_BENCH_ONE_GPU") == "1" or "--launch-check" in sys.argv      # functional tests: ranks share device 0 / need no device
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if not one_gpu and have < n:
        print(f"bench: --gpus {n} requested but {have} GPU(s) are visible to this process: not launching (a smaller run would print a misleading line)", file=sys.stderr, flush=True)
        raise SystemExit(2)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, COALIGN_BENCH_SELF_LAUNCHED="1", GPU_MAX_HW_QUEUES=os.environ.get("GPU_MAX_HW_QUEUES", "8"), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
               OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "8"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    print("bench: self-launch: " + " ".join(cmd), file=sys.stderr, flush=True)
    os.execvpe(sys.executable, cmd, env)

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from coalign_amd import ops  # noqa: E402
from coalign_amd import sharded  # noqa: E402
from coalign_amd import backbone as backbone_mod  # noqa: E402
from coalign_amd.config import builtin_config  # noqa: E402
from coalign_amd.detector import build_model, to_device  # noqa: E402
from coalign_amd.pipeline import FramePipeline  # noqa: E402
from coalign_amd.postprocess import build_postprocessor  # noqa: E402
from coalign_amd.sharded import AgentGather, FrameRing, ring_batch, split_agents, stack_agents  # noqa: E402
from coalign_amd.synthetic import calibrate_heads_, fill_parameters_, make_frame, make_points_frame  # noqa: E402

F32_MFMA_PEAK_TFLOPS = 157.3      # dense fp32 matrix peak, /opt/skills/guides/MI355X_MICROARCH.md
BF16_MFMA_PEAK_TFLOPS = 2500.0    # dense bf16 matrix peak, same guide
HBM_PEAK_GBPS = 8000.0            # 8 TB/s spec (about 6.3 TB/s achievable)
POOL = 8                          # distinct frames the timed loop rotates over


def checksum(boxes, scores):
    """Order-sensitive digest of one frame's detections: (count, sum of scores, sum of |corner| in float64)."""
    if boxes is None:
        return [0, 0.0, 0.0]
    return [int(boxes.shape[0]), round(float(scores.double().sum()), 9), round(float(boxes.double().abs().sum()), 6)]


def hip_time(fn, iters=10, warm=3):
    """Average duration (ms) of ``fn`` alone on the GPU: HIP events on the stream the kernels are launched on."""
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def pctl(xs, q):
    xs = sorted(xs)
    return None if not xs else round(xs[min(len(xs) - 1, int(round(q * (len(xs) - 1))))], 4)


def latency_stats(lat_ms):
    return {"p50": pctl(lat_ms, 0.5), "p99": pctl(lat_ms, 0.99), "max": pctl(lat_ms, 1.0), "frames": len(lat_ms)}


def kernel_trace_us():
    """Average kernel durations (us) of the pillar op and the fusion from the committed rocprofv3 --kernel-trace --stats summary of
    tools/kernels_only.py (same workload as the roofline figures): an independent clock beside the HIP events."""
    import csv
    path = next((q for q in (os.path.join(ROOT, "profiles", r, "kernels_isolated_stats.csv") for r in ("round6", "round5", "round4", "round3")) if os.path.exists(q)), "")
    if not path:
        return None
    out = {}
    for r in csv.DictReader(open(path)):
        n, us = r["Name"], float(r["AverageNs"]) / 1e3
        for key, tag in (("pillar_prep_kernel", "pillar_prep_us"), ("pillar_rows_mx_kernel", "pillar_rows_us"), ("warp_fuse_nhwc_kernel", "fuse_us"), ("pillar_sparse_kernel", "pillar_sparse_us")):
            if key in n:
                out[tag] = round(us, 2)
    if "pillar_sparse_us" in out:                                  # the timed configuration's pillar op since round 4
        out["pillar_op_us"] = out["pillar_sparse_us"]
    elif "pillar_prep_us" in out and "pillar_rows_us" in out:
        out["pillar_op_us"] = round(out["pillar_prep_us"] + out["pillar_rows_us"], 2)
    out["source"] = os.path.relpath(path, ROOT)
    return out


def graph_time(fn, dev, iters=10, reps=8):
    """GPU-side duration (ms) of the launches ONE call of ``fn`` makes: ``reps`` calls are captured into one HIP graph and the graph is replayed, so
    that neither the host-side gaps between small dependent launches (ctypes + allocator, ~10 us each) nor the fixed cost of a graph replay
    (measured round 4: 17 us per replay of a graph holding ONE empty kernel -- MI355X_MICROARCH.md "graph-replay-floor") count as kernel time.
    Rounds 1-3 captured one call per graph: ops shorter than the floor read as the floor.  Warm-up and capture run on the same stream (per-stream
    state such as the persistent canvas or the sparse canvas's frame tag must exist before the capture)."""
    gs = torch.cuda.Stream(device=dev)
    gs.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(gs):
        fn()
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    keep = []
    with torch.cuda.graph(g, stream=gs):
        for _ in range(reps):
            keep.append(fn())
    torch.cuda.synchronize()
    return hip_time(g.replay, iters=iters) / reps


def pillar_bytes_moved(M):
    """Bytes the persistent-canvas pillar op moves per call (the timed configuration; no dense zero-fill): per pillar 532 B in (32 x 16 B points +
    16 B coordinates + 4 B count), 256 B feature row + 256 B canvas row out, 256 B to clear the row the previous frame wrote, 12 B of slot list / cell map."""
    return M * (532 + 256 + 256 + 256 + 12)


def pillar_bytes_sparse(M):
    """Bytes of the one-launch sparse-canvas pillar op per call: per pillar 532 B in, 256 B feature row out, one 8-byte stamp read-modify-write (16 B)."""
    return M * (532 + 256 + 16)


def conv_layer_rooflines(dev, N, ny, nx, pmc):
    """the six 3x3 layer shapes of the frame, each alone on the GPU (HIP events around graph replays of 8 launches): executed 16-bit
    TFLOP/s against the 2.5 PFLOP/s dense fp16 matrix peak, algorithmic HBM bytes (maps in + residual + out + weights once), launches per frame, and -- from the
    committed rocprofv3 --pmc passes where present -- the matrix-pipe busy share and the corrected HBM bytes.  Kernels: coalign_conv3x3_sp on SplitMaps (the
    stride-1 layers inside a stage and both convolutions of the shrink header) and coalign_conv3x3_sp_s2 (round 6: the strided layers that open stages 2 and 3;
    rounds 4-5 ran them on the consumer-split kernel of csrc/conv3x3_emu.hip)."""
    H1, W1 = ny // 2, nx // 2
    cases = [("stage 1: 64 -> 64 @ %dx%d x %d agents" % (H1, W1, N), "conv_sp_64ch", N, 64, 64, H1, W1, True, 5),
             ("stage 2: 128 -> 128 @ %dx%d x %d" % (H1 // 2, W1 // 2, N), "conv_sp_128ch", N, 128, 128, H1 // 2, W1 // 2, True, 9),
             ("stage 3: 256 -> 256 @ %dx%d x %d" % (H1 // 4, W1 // 4, N), "conv_sp_256ch", N, 256, 256, H1 // 4, W1 // 4, True, 15),
             ("shrink header 2nd: 256 -> 256 @ %dx%d x 1" % (H1, W1), "conv_sp_shrink2_256ch_100x352", 1, 256, 256, H1, W1, True, 1),
             ("shrink header 1st: 384 -> 256 @ %dx%d x 1 (SplitMap from the up-sampling heads in, SplitMap out)" % (H1, W1), "conv_sp_shrink1_384ch_100x352", 1, 384, 256, H1, W1, True, 1),
             ("first of stage 2 (stride 2): 64 -> 128 @ %dx%d -> %dx%d x %d (round 6: SplitMap in, SplitMap out)" % (H1, W1, H1 // 2, W1 // 2, N),
              "conv_sp_s2_64to128", N, 64, 128, H1, W1, False, 1),
             ("first of stage 3 (stride 2): 128 -> 256 @ %dx%d -> %dx%d x %d" % (H1 // 2, W1 // 2, H1 // 4, W1 // 4, N), "conv_sp_s2_128to256", N, 128, 256, H1 // 2, W1 // 2, False, 1)]
    rows = []
    for name, key, n, ci, co, H, W, sp, per_frame in cases:
        try:
            g = torch.Generator(device=dev).manual_seed(ci + H)
            x = torch.relu(torch.randn((n, ci, H, W), generator=g, device=dev))
            w16 = ops.pack_conv3x3_emu_weight(torch.randn((co, ci, 3, 3), generator=g, device=dev) / (9 * ci) ** 0.5, 16, True)
            b = torch.randn(co, generator=g, device=dev)
            if sp:
                res = None if n == 1 else ops.SplitMap.pack(torch.randn((n, co, H, W), generator=g, device=dev))
                xs_ = ops.SplitMap.pack(x)
                fn = lambda: ops.conv3x3_sp(xs_, w16, b, co, res, True, out_split=n > 1 or ci != co)
                kern = "conv3x3_sp_kernel (csrc/conv3x3_sp.hip)" + (", stream-K" if ops.conv3x3_sp_is_split(n, ci, co, H, W) else "")
            else:                                         # the strided first convolution of a stage (round 6: on split operands too; the first stage's reads the sparse canvas)
                res = None
                xs_ = ops.SplitMap.pack(x)
                wsk = ops.pack_conv1x1_sp_weight(torch.randn((co, ci, 1, 1), generator=g, device=dev) / ci ** 0.5)
                fn = lambda: ops.conv3x3_sp_s2(xs_, w16, b, co, True, w_skip=wsk)
                kern = "conv3x3_sp_s2_kernel with the block's 1x1 skip convolution as a tenth tap (csrc/conv3x3_sp_s2.hip)"
            ms = graph_time(fn, dev)
            so = 1 if sp else 4                           # output pixels per input pixel: 1, or 1/4 for the strided layer
            executed = 3 * 2 * (9 if sp else 10) * ci * co * H * W * n // so      # (the strided layers carry the skip convolution as a tenth tap)
            hbm = 4 * n * H * W * ci + 4 * n * H * W * (co + (co if res is not None else 0)) // so + w16.numel()
            row = {"layer": name, "kernel": kern, "launches_per_frame": per_frame, "us": round(ms * 1e3, 2), "executed_TFLOPs": round(executed / ms / 1e9, 1),
                   "frac_of_fp16_peak": round(executed / ms / 1e9 / BF16_MFMA_PEAK_TFLOPS, 4), "fp32_equivalent_TFLOPs": round(executed / 3 / ms / 1e9, 1),
                   "algorithmic_hbm_MB": round(hbm / 1e6, 2), "hbm_GBps_if_streamed_once": round(hbm / ms / 1e6, 1)}
            e = pmc.get(key) if pmc else None
            if e:
                c = e.get("per_op_call", {})
                if c.get("SQ_VALU_MFMA_BUSY_CYCLES") and c.get("GRBM_GUI_ACTIVE"):      # busy cycles are summed over the 1024 SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs
                    row["pmc_mfma_busy_share"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (128.0 * c["GRBM_GUI_ACTIVE"]), 4)
                if "hbm_bytes_read_x2" in e:
                    row["pmc_hbm_MB"] = round(e["hbm_bytes_read_x2"] / 1e6, 2)
            rows.append(row)
            del x, w16, b, res
        except Exception as exc:      # noqa: BLE001  a side report must never cost the headline line
            rows.append({"layer": name, "error": f"{type(exc).__name__}: {str(exc)[:160]}"})
            torch.cuda.synchronize()
    torch.cuda.empty_cache()
    return rows


def size_sweep(dev, steps=20, streams=None, lanes=3, result_lag=1, queue_depth=1):
    """VERDICT r03 item 3 / SURVEY 8d: the HBM-bound ops where the roofline bites.  Pillar op at M in {8000, 32000, 70000} pillars per agent
    (max_voxel_train / max_voxel_test of pointpillar_coalign.yaml:52-54) x 5 agents, at N = 2 (cfg 2) and at DAIR-V2X geometry 504 x 200 (cfg 4);
    the fusion launch per geometry / agent count; whole-path frames/s for cfg 2 and cfg 4.  Times: HIP events around graph replays of the op's
    launches (as `north_star_hbm`); `frac` = model bytes / time / 8 TB/s -- algorithmic, per launch; PMC traffic lives in profiles/."""
    out = []
    cases = [("opv2v_coalign", 5, 8000, False), ("opv2v_coalign", 5, 32000, False), ("opv2v_coalign", 5, 70000, False),
             ("opv2v_coalign", 2, 8000, True), ("dairv2x_coalign", 2, 8000, True)]
    models = {}
    for cfg, n, m, whole in cases:
        try:
            hy = builtin_config(cfg)
            nx, ny, _ = [int(v) for v in hy["model"]["args"]["point_pillar_scatter"]["grid_size"]]
            if cfg not in models:
                mdl = build_model(hy)
                fill_parameters_(mdl, seed=0)
                models[cfg] = mdl.to(dev).eval()
            mdl = models[cfg]
            fr = to_device(make_frame(hy, n, pillars_per_agent=m, seed=303, noise=(0.2, 0.2), infra_agent=cfg.startswith("dair")), dev)
            fr["record_len"] = [n]
            M = int(fr["processed_lidar"]["voxel_features"].shape[0])
            pl_in = dict(fr["processed_lidar"], record_len=[n])
            from coalign_amd import detector as det_mod
            rn = getattr(mdl.backbone, "resnet", None)
            sparse = bool(det_mod.SPARSE_CANVAS and rn is not None and hasattr(rn, "layer0") and rn.layer0[0].takes_sparse_canvas())
            keep = (mdl.pillar_vfe.persistent_canvas, mdl.pillar_vfe.sparse_canvas)
            mdl.pillar_vfe.persistent_canvas, mdl.pillar_vfe.sparse_canvas = True, sparse
            with torch.no_grad():
                t_p = graph_time(lambda: mdl.pillar_vfe(dict(pl_in)), dev)
                mdl.pillar_vfe.sparse_canvas = keep[1]
                feats, aff = mdl.encode(fr)
                t_f = graph_time(lambda: mdl._fuse_scales(list(feats), [n], aff), dev)
            mdl.pillar_vfe.persistent_canvas = keep[0]
            fb = sum((n + 1) * int(f.shape[1]) * int(f.shape[2]) * int(f.shape[3]) * 4 for f in feats)
            pb = pillar_bytes_sparse(M) if sparse else pillar_bytes_moved(M)
            row = {"config": cfg, "canvas": [nx, ny], "agents": n, "pillars_per_agent": m,
                   "pillar_op": {"form": "one launch, sparse canvas" if sparse else "persistent dense canvas, two launches", "us": round(t_p * 1e3, 2), "bytes_moved_model": pb, "GBps": round(pb / t_p / 1e6, 1), "frac": round(pb / t_p / 1e6 / HBM_PEAK_GBPS, 4),
                                 "read_bytes_survey_8d": M * 532, "frac_read_only": round(M * 532 / t_p / 1e6 / HBM_PEAK_GBPS, 4)},
                   "fusion": {"us": round(t_f * 1e3, 2), "algorithmic_bytes": fb, "GBps": round(fb / t_f / 1e6, 1), "frac": round(fb / t_f / 1e6 / HBM_PEAK_GBPS, 4)},
                   "path_frac": round((pb + fb) / (t_p + t_f) / 1e6 / HBM_PEAK_GBPS, 4)}
            del feats, aff
            if whole:
                pp = build_postprocessor(hy["postprocess"], False)
                anchors = torch.from_numpy(pp.generate_anchor_box())
                pool = [fr] + [dict(to_device(make_frame(hy, n, pillars_per_agent=m, seed=304 + i, noise=(0.2, 0.2), infra_agent=cfg.startswith("dair")), dev), record_len=[n]) for i in range(3)]
                import copy
                mc = copy.deepcopy(mdl)
                calibrate_heads_(mc, pool[0], pp.params["target_args"]["score_threshold"], 600)
                pipe = FramePipeline(mc, pp, anchors, lanes=lanes, queue_depth=queue_depth, result_lag=result_lag, graph=True, device=dev, streams=streams if streams is not None and len(streams) >= lanes else None)
                for i in range(8):
                    pipe.submit(pool[i % 4])
                pipe.drain(); torch.cuda.synchronize()
                t0 = time.perf_counter()
                res = []
                for i in range(steps):
                    res += pipe.submit(pool[i % 4])
                res += pipe.drain(); torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                row["whole_path"] = {"frames_per_s": round(steps / dt, 2), "ms_per_frame": round(dt / steps * 1e3, 4), "frames_in_flight": lanes * queue_depth, "hip_graph": True,
                                     "detections_last_frame": 0 if res[-1][1] is None else int(res[-1][1].shape[0])}
                pipe.close()
                del mc, pipe
            out.append(row)
            del fr
            torch.cuda.empty_cache()
        except Exception as e:      # noqa: BLE001  a side report must never cost the headline line
            out.append({"config": cfg, "agents": n, "pillars_per_agent": m, "error": f"{type(e).__name__}: {str(e)[:200]}"})
            torch.cuda.synchronize()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--agents", type=int, default=5)
    ap.add_argument("--pillars", type=int, default=8000)
    ap.add_argument("--config", default="opv2v_coalign")
    ap.add_argument("--lanes", type=int, default=0, help="HIP streams that carry frames (0 = 2 for the single-GPU graph pipeline, 3 for multi-rank graph runs, 4 with eager launches; "
                    "final code, 300 steps: 2 x 3 queued 650-659, 3 x 2 queued 649, 3 x 1 (lag 2) 646-648, 3 x 1 (lag 1) 639-642, 4 x 1 (lag 3) 593 frames/s)")
    ap.add_argument("--result-lag", type=int, default=-1, help="frames between enqueueing a frame and collecting its detections on the host (-1: 1, or lanes x depth - 1 with a queue depth > 1: the host waits for the OLDEST queued frame only)")
    ap.add_argument("--queue-depth", type=int, default=0, help="frames queued per lane stream: lanes x depth pipeline lanes share `lanes` HIP streams -- a stream's next frames are enqueued before its current one has finished, so no stream waits for the host between frames (0: 3 for the single-GPU graph pipeline, else 1)")
    ap.add_argument("--no-miopen-find", action="store_true", help="leave torch.backends.cudnn.benchmark off (MIOpen immediate mode)")
    ap.add_argument("--no-graph", action="store_true", help="~150 eager launches per frame (1.5 ms of host time) instead of one HIP graph replay per frame "
                    "(0.1 ms; single-GPU default: measured +1 ... +4 % frames/s on the channels-last route; multi-rank runs are always eager)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-numerics", action="store_true", help="skip the float64 end-to-end error report (`numerics`: two float64 CPU forwards of the benchmarked frame, ~20-40 s)")
    ap.add_argument("--conv-emu", type=int, default=None, choices=(0, 2, 3, 16),
                    help="override the 3x3 convolution arithmetic: 0 = native fp32 MFMA / MIOpen, 3 / 2 = split-bf16 products (default: the package default)")
    ap.add_argument("--no-side-modes", action="store_true", help="skip the extra timed passes of the other convolution modes")
    ap.add_argument("--from-points", action="store_true", help="ALSO time the loop fed from raw point clouds in pinned host memory (async H2D + device "
                    "voxeliser + encoder in every frame: `from_points` in the JSON line; `value` stays the from-pillars metric of BASELINE.json); on by default at --gpus 1")
    ap.add_argument("--no-from-points", action="store_true")
    ap.add_argument("--no-latency", action="store_true", help="skip the one-frame-in-flight latency pass")
    ap.add_argument("--no-size-sweep", action="store_true", help="skip the `size_sweep` report (pillar op / fusion at 8 000 / 32 000 / 70 000 pillars per agent, cfg 2, cfg 4)")
    ap.add_argument("--mode", choices=("ring", "gather"), default="ring", help="multi-GPU schedule: 'ring' = one frame per rank and step, agents routed by all-to-all "
                    "(weak scaling, the default); 'gather' = ONE frame per step, rank r encodes its block of the agents, one all-gather per scale, every rank then "
                    "holds all maps and runs the ego tail (north_star's one-agent-per-GPU wording; strong scaling of a single frame)")
    ap.add_argument("--comm-per-lane", action="store_true", help="one RCCL communicator per lane instead of ONE shared by all lanes (collectives are issued in frame order "
                    "on every rank, so one communicator is deadlock-free by construction; several communicators used concurrently are not)")
    ap.add_argument("--force-dist", action="store_true", help="run the MULTI-RANK path with whatever world size there is -- at --gpus 1: init_process_group('nccl', world_size=1), "
                    "FrameRing / AgentGather execute their collective as a self exchange on RCCL, two HIP graphs per lane around it (first contact with RCCL on a 1-GPU box; "
                    "the line's `rccl` key reports it; not the headline configuration)")
    ap.add_argument("--dry-run", action="store_true", help="no GPU work: build the exchange plans and shape-only buffers of EVERY rank for --gpus N and validate them against what "
                    "RCCL's all_to_all_single / all_gather_into_tensor require (coalign_amd.sharded.preflight); prints the report as one JSON line")
    ap.add_argument("--launch-check", action="store_true", help="no GPU work: bring the ranks up (self-launch included), one all-reduce over gloo, rank 0 prints {launch_check, world, "
                    "n_gpus} as one JSON line -- the CPU test of `python bench.py --gpus N` launching N ranks by itself (tests/test_sharded_cpu.py)")
    ap.add_argument("--cpu-frames", type=int, default=10)
    ap.add_argument("--cpu-threads", type=int, default=0, help="torch CPU threads for the oracle (0 = best of the committed sweep, else 16)")
    ap.add_argument("--cpu-budget-s", type=float, default=30.0)
    args = ap.parse_args()
    if args.dry_run:
        from coalign_amd.sharded import preflight
        hy = builtin_config(args.config)
        nx_, ny_, _ = [int(v) for v in hy["model"]["args"]["point_pillar_scatter"]["grid_size"]]
        shapes = [(64, ny_ // 2, nx_ // 2), (128, ny_ // 4, nx_ // 4), (256, ny_ // 8, nx_ // 8)]
        lanes = args.lanes if args.lanes > 0 else 4
        rep = {"dry_run": True, "gpus": args.gpus, "ring": preflight(args.gpus, args.agents, shapes, True, None, lanes, "ring"),
               "gather": preflight(args.gpus, args.agents, shapes, True, None, lanes, "gather"),
               "launch": f"python -m torch.distributed.run --nnodes=1 --nproc-per-node {args.gpus} --master-addr 127.0.0.1 --master-port P bench.py --gpus {args.gpus} --mode {args.mode}"}
        print(json.dumps(rep), flush=True)
        return

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args.gpus)                                     # (does not return)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # --force-dist (round 6): the multi-rank code path -- process group, control group, schedule negotiation, two graphs per lane around the collective -- with a
    # world of ONE, the collectives running as self all-to-all / self all-gather on the real backend: how RCCL is exercised on a 1-GPU box
    multi = world > 1 or args.force_dist
    if args.force_dist and world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's rank count and --gpus must agree")
    if args.launch_check:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        t = torch.tensor([rank + 1], dtype=torch.int64)
        dist.all_reduce(t)
        ok = int(t.item()) == world * (world + 1) // 2
        if rank == 0:
            print(json.dumps({"launch_check": bool(ok), "world": world, "n_gpus": world, "launcher": "self" if os.environ.get("COALIGN_BENCH_SELF_LAUNCHED") == "1" else "external"}), flush=True)
        dist.barrier()
        dist.destroy_process_group()
        if not ok:
            raise SystemExit(3)
        return
    # COALIGN_BENCH_BACKEND=gloo + COALIGN_BENCH_ONE_GPU=1: functional test of the multi-rank code path on a single
    # GPU (ranks share device 0, the exchange is staged through host memory).  Never used for reported numbers.
    backend = os.environ.get("COALIGN_BENCH_BACKEND", "nccl")
    if os.environ.get("COALIGN_BENCH_ONE_GPU") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if multi:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # a collective a peer never enters must end as an exception, not as a hang until the driver's clock runs out: 10 minutes (default 30)
        import datetime
        pg_timeout = datetime.timedelta(seconds=int(os.environ.get("COALIGN_BENCH_PG_TIMEOUT_S", "600")))
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=pg_timeout)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=pg_timeout)

    if args.conv_emu is not None:
        backbone_mod.CONV_EMU_TERMS = args.conv_emu
    default_terms = backbone_mod.CONV_EMU_TERMS
    hypes = builtin_config(args.config)
    N = args.agents
    nx, ny, _ = [int(v) for v in hypes["model"]["args"]["point_pillar_scatter"]["grid_size"]]
    model = build_model(hypes)
    fill_parameters_(model, seed=0)
    model = model.to(dev).eval()
    pp = build_postprocessor(hypes["postprocess"], False)
    anchors = torch.from_numpy(pp.generate_anchor_box())

    # the frame pool: POOL distinct 5-agent frames (seed 303 + i, pose noise 0.2 m / 0.2 deg), identical on every rank
    pool_n = POOL if (not multi) else world * max(1, POOL // world)
    frames_cpu = [make_frame(hypes, N, pillars_per_agent=args.pillars, seed=303 + i, noise=(0.2, 0.2)) for i in range(pool_n)]
    frames = []
    for f in frames_cpu:
        d = to_device(f, dev)
        d["record_len"] = [N]                      # host-side agent counts: no device->host sync per frame
        d["pairwise_t_matrix_host"] = f["pairwise_t_matrix"]      # the dataset's host copy of the 5 x 5 pose matrices travels with the batch (normalised on the host, pipeline.py)
        frames.append(d)
    calibrate_heads_(model, frames[0], pp.params["target_args"]["score_threshold"], 600)     # same frame, same weights on every rank
    if multi:   # identical weights everywhere, bit for bit
        for p in model.parameters():
            if backend == "nccl":
                dist.broadcast(p.data, 0)
            else:
                buf = p.data.cpu()
                dist.broadcast(buf, 0)
                p.data.copy_(buf)

    use_graph = not args.no_graph          # (multi-rank: two graphs per lane and frame around the collective, coalign_amd/pipeline.py)
    # Single GPU, HIP graphs (the headline configuration, second half of round 5): TWO streams, THREE frames queued on each.  Two frames running side by side is what the
    # GPU takes (3 / 4 concurrent frames measure -0 / -7 %); what the earlier "3 lanes, result lag 1" left on the table was the host round trip between a stream's frames
    # (collect, stage, launch: 0.15-0.2 ms during which the other frame ran alone).  Same box, 300 steps: 639-642 -> 650-659 frames/s; 20 steps: 597-600 -> 622-628.
    n_lanes = args.lanes if args.lanes > 0 else ((2 if (not multi) else 3) if use_graph else 4)
    queue_depth = args.queue_depth if args.queue_depth > 0 else (3 if use_graph and (not multi) else 1)
    rings = None
    exchanges = None
    mode = args.mode
    rccl = None
    ctl = None

    def agree(ok: bool) -> bool:
        """All ranks take the same fall-back decision, over a gloo control group that does not depend on the data-plane backend."""
        return sharded.control_agree(ok, ctl)

    def setup_mode(m):
        """-> (rings, exchanges, step_batches) of schedule m in {"ring", "gather", "replicas"}."""
        # ONE communicator for all lanes by default: every rank issues its collectives in frame order (frame i -> lane i mod lanes, the
        # exchange enqueued inside submit()), so a single RCCL communicator sees the same sequence everywhere -- deadlock-free by
        # construction; the lanes' exchanges then serialise on the communicator's stream (0.1 ms each against ~2 ms of compute).
        # --comm-per-lane restores one communicator per lane (concurrent use of several communicators is documented as deadlock-prone).
        if m == "replicas":       # no collective at all: every rank runs whole frames out of the common pool (rank r starts at frame r)
            return None, None, [frames[(i + rank) % len(frames)] for i in range(len(frames))]
        groups = [dist.new_group(backend=backend) for _ in range(n_lanes)] if args.comm_per_lane else [None] * n_lanes
        by_agent = [split_agents(f) for f in frames]
        if m == "ring":
            rs = [FrameRing(N, group=g, force_collective=args.force_dist) for g in groups]
            period = pool_n // world
            return rs, [r.exchange for r in rs], [ring_batch(by_agent, [f["pairwise_t_matrix"] for f in frames], rank, world, N, s) for s in range(period)]
        # gather mode: every step is ONE frame; this rank encodes its contiguous block of the agents (empty slots where the block runs past
        # N), the per-scale maps are all-gathered, and the ego tail runs with all N agents (on every rank: same latency, rank 0 reports)
        gathers = [AgentGather(N, group=g, force_collective=args.force_dist) for g in groups]
        per = gathers[0].per
        sb = []
        for g_, f in enumerate(frames):
            sets = [by_agent[g_][a] if a < N else None for a in range(rank * per, rank * per + per)]
            if all(s_ is None for s_ in sets):       # a rank without agents still takes part in the collective: one empty slot set
                sets = [{k: v[:0] for k, v in by_agent[g_][0].items()}] + [None] * (per - 1)
            sb.append({"processed_lidar": stack_agents(sets), "record_len": [per], "tail_record_len": [N], "pairwise_t_matrix": f["pairwise_t_matrix"],
                       "pairwise_t_matrix_host": f["pairwise_t_matrix_host"]})
        return gathers, [(lambda feats, _g=g: (_g.gather(feats), None)) for g in gathers], sb

    if multi:
        # ---- first contact with RCCL, hardened (the reference's own bring-up: opencood/tools/multi_gpu_utils.py:31-37).
        #      (1) who is here: every rank's device over the control group; (2) one small all-gather on the data plane; (3) the chosen schedule is
        #      warmed up inside try / except on every rank and the ranks agree on the outcome: ring -> gather -> independent replicas.
        ctl = dist.new_group(backend="gloo")
        props = torch.cuda.get_device_properties(dev)
        me = {"rank": rank, "local_rank": local_rank, "device": torch.cuda.get_device_name(dev), "uuid": str(getattr(props, "uuid", "")),
              "pci_bus_id": getattr(props, "pci_bus_id", None), "host": os.uname().nodename}
        seen = [None] * world
        dist.all_gather_object(seen, me, group=ctl)
        ok = True
        try:
            if backend == "nccl":
                t = torch.full((4,), rank, dtype=torch.int32, device=dev)
                o = torch.empty(4 * world, dtype=torch.int32, device=dev)
                dist.all_gather_into_tensor(o, t)
                torch.cuda.synchronize()
                ok = o.view(world, 4)[:, 0].cpu().tolist() == list(range(world))
        except Exception as e:      # noqa: BLE001
            print(f"bench[{rank}]: data-plane all-gather failed: {type(e).__name__}: {str(e)[:200]}", file=sys.stderr, flush=True)
            ok = False
        rccl = {"world": world, "backend": "RCCL" if backend == "nccl" else backend, "ranks_seen": seen,
                "distinct_devices": len({(d["host"], d["uuid"] or d["pci_bus_id"] or d["local_rank"]) for d in seen}), "data_plane_all_gather_ok": agree(ok),
                "requested_mode": args.mode, "fallbacks": [], "forced_world_one": bool(args.force_dist and world == 1)}
        if not rccl["data_plane_all_gather_ok"]:
            mode = "replicas"
            rccl["fallbacks"].append("data-plane all-gather failed -> replicas")
        rings, exchanges, step_batches = setup_mode(mode)
    else:
        step_batches = frames
    torch.backends.cudnn.benchmark = not args.no_miopen_find     # MIOpen find during warm-up for whatever still runs on it

    # every pipeline of this process runs on the SAME lane streams (FramePipeline(streams=...)): streams share a handful of hardware queues
    n_streams = n_lanes
    if multi:
        queue_depth = 1                                          # (multi-rank lanes own an exchange each: one frame per stream there)
    n_lanes = n_streams * queue_depth                            # pipeline lanes = frames queued or running; lane i runs on stream i mod n_streams (FramePipeline(queue_depth=...))
    result_lag = args.result_lag if args.result_lag >= 0 else (n_lanes - 1 if queue_depth > 1 else 1)
    lane_streams = [torch.cuda.Stream(device=dev) for _ in range(n_streams)]

    def make_pipe(graph):
        if not graph and use_graph and (not multi):               # the eager side mode / profile pass of a graph run: host-bound launches want one frame per stream (rounds 3-4's form)
            while len(lane_streams) < 3:
                lane_streams.append(torch.cuda.Stream(device=dev))
            return FramePipeline(model, pp, anchors, lanes=3, result_lag=1, graph=False, device=dev, streams=lane_streams)
        return FramePipeline(model, pp, anchors, lanes=n_streams, queue_depth=queue_depth, result_lag=result_lag, graph=graph, device=dev,
                             exchange=exchanges, streams=lane_streams)

    def sync():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_run(pipe, steps, warmup, batches=None, points=False):
        batches = step_batches if batches is None else batches
        submit = pipe.submit_points if points else pipe.submit
        for s in range(warmup):
            submit(batches[s % len(batches)])
        pipe.drain()
        sync()
        pipe.host_enqueue_s = 0.0
        pipe.latencies_ms.clear()            # submit -> detections on the host, per frame of the timed loop
        results = []
        base = pipe._count                   # frame indices of the results below are made relative to the timed loop
        t0 = time.perf_counter()
        for s in range(steps):
            results += submit(batches[s % len(batches)])
        t_issue = pipe.host_enqueue_s
        results += pipe.drain()              # the last frames' detections: all K frames are complete inside the bracket
        sync()
        return time.perf_counter() - t0, t_issue, [(i - base, b, sc) for i, b, sc in results]

    pipe = make_pipe(use_graph)
    # `warmup` is what was asked for.  The lanes' HIP graphs are captured BEFORE the warm-up, in a separate untimed pass reported as
    # `config.lane_capture_frames` (one capture + one replay per pipeline lane: rounds 2-5 folded them into a raised warm-up count)
    warm = args.warmup
    capture_frames = 2 * n_lanes if use_graph else 0
    if multi:
        # the chosen schedule's first exchanges, guarded: any error on any rank moves ALL ranks to the next schedule (ring -> gather -> replicas;
        # coalign_amd.sharded.negotiate_schedule, CPU-tested with gloo ranks in tests/test_sharded_cpu.py)
        state = {"mode": mode}

        def attempt(m):
            nonlocal rings, exchanges, step_batches, pipe, use_graph
            if m != state["mode"]:                   # a fall-back: rebuild the schedule and a pipeline around it
                try:
                    torch.cuda.synchronize()
                except Exception:      # noqa: BLE001
                    pass
                rings, exchanges, step_batches = setup_mode(m)
                pipe = make_pipe(use_graph)
                state["mode"] = m
            # fault injection for the fall-back chain's own test (tools/gpu_multirank_check.sh): COALIGN_BENCH_INJECT_FAIL="ring:1,gather:0" makes the first
            # exchanges of schedule `ring` raise on rank 1 and of `gather` on rank 0
            for item in filter(None, os.environ.get("COALIGN_BENCH_INJECT_FAIL", "").split(",")):
                sched, r_ = item.split(":")
                if sched == m and int(r_) == rank:
                    raise RuntimeError(f"injected failure of schedule '{m}' on rank {rank}")
            ok_g = True
            try:
                for s_ in range(n_lanes):
                    pipe.submit(step_batches[s_ % len(step_batches)])
                pipe.drain()
                torch.cuda.synchronize()
            except Exception as e:      # noqa: BLE001
                if not use_graph:
                    raise
                ok_g = False
                print(f"bench[{rank}]: schedule '{m}' with HIP graphs failed ({type(e).__name__}: {str(e)[:160]}); trying eager launches", file=sys.stderr, flush=True)
            if use_graph and not agree(ok_g):        # any rank's capture failed: every rank runs this schedule with eager launches
                try:
                    torch.cuda.synchronize()
                except Exception:      # noqa: BLE001
                    pass
                use_graph = False
                rccl["fallbacks"].append(f"{m}: HIP-graph capture failed on some rank -> eager launches")
                rings, exchanges, step_batches = setup_mode(m)
                pipe = make_pipe(False)
                for s_ in range(n_lanes):
                    pipe.submit(step_batches[s_ % len(step_batches)])
                pipe.drain()
                torch.cuda.synchronize()

        def report(m, e):
            print(f"bench[{rank}]: schedule '{m}' failed in its first exchanges: {type(e).__name__}: {str(e)[:200]}", file=sys.stderr, flush=True)

        mode, taken = sharded.negotiate_schedule(mode, attempt, agree, report)
        rccl["fallbacks"] += taken
        rccl["mode_run"] = mode
        # what one exchange costs on this node: the feature maps of one step through the schedule's collective, HIP events, max over ranks
        if exchanges is not None:
            try:
                with torch.no_grad():
                    feats_x, _ = model.encode(step_batches[0])
                    ex_ms = hip_time(lambda: exchanges[0](list(feats_x)), iters=5, warm=2)
                t = torch.tensor([ex_ms], dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX, group=ctl)
                rccl["exchange_ms_per_frame"] = round(float(t.item()), 4)
                rccl["bytes_sent_per_rank_per_exchange"] = int(rings[0].bytes_sent_last)
                rccl["bytes_per_agent_fp32"] = int(sum(f.numel() // max(f.shape[0], 1) for f in feats_x) * 4)
                del feats_x
            except Exception as e:      # noqa: BLE001
                rccl["exchange_timing_error"] = f"{type(e).__name__}: {str(e)[:160]}"
    if use_graph and (not multi):       # a capture that fails here (driver / allocator state of this box) must not cost the bench line: fall back to eager
        try:
            for s_ in range(capture_frames):
                pipe.submit(step_batches[s_ % len(step_batches)])
            pipe.drain()
            torch.cuda.synchronize()
        except Exception as e:      # noqa: BLE001
            print(f"bench: HIP-graph capture failed ({type(e).__name__}: {str(e)[:160]}); running eager", file=sys.stderr, flush=True)
            torch.cuda.synchronize()
            use_graph = False
            pipe = make_pipe(False)
    # ---- kernels alone on the GPU (outside the timed region): the roofline figures comparable to a rocprofv3 kernel trace
    iso = {}
    if rank == 0:
        with torch.no_grad():
            f0 = frames[0]
            pl_in = dict(f0["processed_lidar"], record_len=f0["record_len"])
            # the pillar op of the timed configuration: the one-launch sparse-canvas encoder when the model's first ResNet block reads it (round 4), else
            # the persistent dense canvas (two launches); the other forms beside it
            from coalign_amd import detector as det_mod
            rn = getattr(model.backbone, "resnet", None)
            sparse_on = bool(det_mod.SPARSE_CANVAS and rn is not None and hasattr(rn, "layer0") and rn.layer0[0].takes_sparse_canvas() and not getattr(model, "compression", False))
            tp = (lambda fn: graph_time(fn, dev)) if (not multi) else hip_time
            if sparse_on:
                model.pillar_vfe.sparse_canvas = True
                iso["pillar_ms"] = tp(lambda: model.pillar_vfe(dict(pl_in)))
                model.pillar_vfe.sparse_canvas = False
                iso["pillar_dense_persistent_ms"] = tp(lambda: model.pillar_vfe(dict(pl_in)))
            else:
                iso["pillar_ms"] = tp(lambda: model.pillar_vfe(dict(pl_in)))
            if model.pillar_vfe.persistent_canvas and (not multi):      # the same op on a fresh canvas every call (dense memset included)
                model.pillar_vfe.persistent_canvas = False
                iso["pillar_fresh_canvas_ms"] = graph_time(lambda: model.pillar_vfe(dict(pl_in)), dev)
                model.pillar_vfe.persistent_canvas = True
            gx = torch.randn(N, 64, ny // 2, nx // 2, device=dev)
            gwt = torch.randn(64, 64, 3, 3, device=dev) / 24.0
            gb, gr = torch.randn(64, device=dev), torch.randn(N, 64, ny // 2, nx // 2, device=dev)
            gw = ops.pack_conv3x3_weight(gwt)
            iso["conv_f32_ms"] = hip_time(lambda: ops.conv3x3_bias_act(gx, gw, gb, gr, True))
            from coalign_amd import backbone as _bb
            for terms in (3, 2, 16):   # the weight image the detector uses (tap-major by default: backbone.CONV_EMU_TAP_MAJOR)
                gws = ops.pack_conv3x3_emu_weight(gwt, terms, _bb.CONV_EMU_TAP_MAJOR)
                iso["conv_fp16x2_ms" if terms == 16 else f"conv_bf16x{terms}_ms"] = hip_time(lambda: ops.conv3x3_emu_bias_act(gx, gws, gb, 64, gr, True, terms))
            del gx, gwt, gb, gr, gw
            try:
                # warp + attention fusion of all three scales, launched as the model launches them, replayed from a HIP graph so
                # that host launch overhead does not enter the GPU-side duration
                feats_iso, affine_iso = model.encode(f0)
                for _ in range(3):
                    model._fuse_scales(list(feats_iso), [N], affine_iso)
                torch.cuda.synchronize()
                if (not multi):
                    iso["fuse_ms"] = graph_time(lambda: model._fuse_scales(list(feats_iso), [N], affine_iso), dev)      # (several calls per graph: no replay floor)
                else:       # no stream capture next to a live RCCL communicator (its watchdog thread polls events): plain launches
                    iso["fuse_ms"] = hip_time(lambda: model._fuse_scales(list(feats_iso), [N], affine_iso))
                del feats_iso, affine_iso
            except Exception as e:       # measurement aid only: the bench line must survive it
                iso["fuse_error"] = f"{type(e).__name__}: {str(e)[:120]}"
                torch.cuda.synchronize()
    sync()

    dt, t_issue, results = timed_run(pipe, args.steps, warm)
    lat_default = list(pipe.latencies_ms)
    range_hit = pipe.range_exceeded()          # (outside the bracket) did any SplitMap value leave the fp16 split's operating range?
    if multi:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- per-frame detection digests: pool frame -> digest; every recurrence of a pool frame must reproduce it exactly
    digests, consistent, mismatches = {}, True, []
    for idx, boxes, scores in results[-min(len(results), 2 * pool_n):]:
        g = (idx * world + rank) % pool_n if (multi and mode == "ring") else (idx + rank) % pool_n if (multi and mode == "replicas") else idx % pool_n
        d = checksum(boxes, scores)
        if digests.setdefault(g, d) != d:
            consistent = False
            mismatches.append([idx, g, digests[g], d])
    if multi:
        allg = [None] * world
        dist.all_gather_object(allg, (digests, consistent))
        digests = {}
        for dg, ok in allg:
            consistent = consistent and ok
            for g, d in dg.items():
                consistent = consistent and digests.setdefault(g, d) == d
    last_boxes = results[-1][1] if results else None

    # ---- per-kernel HIP-event profile inside ordinary (eager) steps right after the region (graph replays bypass the op hooks)
    ops.PROFILE = {}
    prof_pipe = pipe if not use_graph else make_pipe(False)
    for s in range(2 * n_lanes):
        prof_pipe.submit(step_batches[s % len(step_batches)])
    prof_pipe.drain()
    sync()
    prof, ops.PROFILE = ops.PROFILE, None

    # ---- per-frame latency: ONE frame in flight, detections collected before the next frame is submitted (the reference's strictly
    #      serial loop, opencood/tools/inference.py:125-179) -- what a 10 Hz LiDAR consumer sees
    latency = None
    if (not multi) and not args.no_latency:
        latency = {"default_pipeline": dict(latency_stats(lat_default), frames_in_flight=n_lanes, streams=n_streams, frames_queued_per_stream=n_lanes // n_streams, result_lag_frames=pipe.result_lag)}
        try:
            p1 = FramePipeline(model, pp, anchors, lanes=1, result_lag=0, graph=use_graph, device=dev, streams=lane_streams)
            d1, _, _ = timed_run(p1, args.steps, max(4, args.warmup))
            latency["one_frame_in_flight"] = dict(latency_stats(p1.latencies_ms), frames_per_s=round(args.steps / d1, 3))
            p1.close()               # (gives the model's encoder flags back and drops the lane's captured graphs: the sections below start from the same state)
            del p1
        except Exception as e:      # noqa: BLE001
            latency["error"] = f"{type(e).__name__}: {str(e)[:200]}"
            torch.cuda.synchronize()

    # ---- the loop fed from RAW POINTS in host memory: pinned staging -> async H2D -> coalign_voxelize -> encoder (count on the device)
    #      -> the same frame.  A second reported mode; `value` stays the from-pillars metric BASELINE.json names.
    from_points = None
    if (not multi) and (args.from_points or not args.no_from_points):
        try:
            from coalign_amd.preprocess import build_preprocessor
            pre = build_preprocessor(hypes["preprocess"], False, dev)
            pframes = [make_points_frame(hypes, N, seed=303 + i, noise=(0.2, 0.2)) for i in range(POOL)]
            n_pts = [int(sum(len(c) for c in f["clouds"])) for f in pframes]
            pv = pre.preprocess_clouds(pframes[0]["clouds"], ego_filter=True)
            m_pillars = int(pv["voxel_features"].shape[0])
            # the heads were calibrated on the pillar pool: calibrate a copy of the model on the first point frame (same ~600 candidates)
            import copy
            model_p = copy.deepcopy(model)
            calibrate_heads_(model_p, {"processed_lidar": {k: pv[k] for k in ("voxel_features", "voxel_coords", "voxel_num_points")}, "record_len": [N],
                                       "pairwise_t_matrix": pframes[0]["pairwise_t_matrix"].to(dev)}, pp.params["target_args"]["score_threshold"], 600)
            del pv
            slot = 1 << (max(max(len(c) for c in f["clouds"]) for f in pframes) - 1).bit_length()
            pp_p = build_postprocessor(hypes["postprocess"], False)
            # (the feeder path moves 10.5 MB host -> device per frame on the frame's own stream: a third stream lets that copy run beside two computing frames;
            #  COALIGN_BENCH_FP="streams,depth,lag" overrides for measurements)
            fps_, fpd_, fpl_ = [int(v) for v in os.environ.get("COALIGN_BENCH_FP", "3,2,3" if use_graph and (not multi) and args.lanes <= 0 and args.queue_depth <= 0 else f"{n_streams},{queue_depth},{result_lag}").split(",")]
            while len(lane_streams) < fps_:
                lane_streams.append(torch.cuda.Stream(device=dev))
            fp = FramePipeline(model_p, pp_p, anchors, lanes=fps_, queue_depth=fpd_, result_lag=fpl_, graph=use_graph, device=dev, preprocessor=pre,
                               points_per_cloud=slot, streams=lane_streams)
            dp, tip, rp = timed_run(fp, args.steps, warm, batches=pframes, points=True)
            from_points = {"value": round(args.steps / dp, 3), "unit": "frames/s", "ms_per_step": round(dp / args.steps * 1e3, 4),
                           "host_enqueue_ms_per_step": round(tip / args.steps * 1e3, 4), "latency_ms": latency_stats(fp.latencies_ms),
                           "streams": fps_, "frames_queued_per_stream": fpd_, "result_lag_frames": fpl_,
                           "points_per_frame": n_pts[0], "pillars_frame0": m_pillars, "h2d_bytes_per_frame": N * slot * 16,
                           "detections_last_frame": 0 if not rp or rp[-1][1] is None else int(rp[-1][1].shape[0]),
                           "input": f"{N} raw 64-beam sweeps per frame (coalign_amd.synthetic.make_point_cloud) in pinned host memory, "
                                    f"{POOL} distinct frames in rotation; per frame: async copy of {N} x {slot} point slots, coalign_voxelize (4 launches), "
                                    "coalign_pillar_encode_sparse with the pillar count on the device (capacity-sized arrays), then the same path as `value`"}
            fp.close()
            if not args.no_latency:
                fp1 = FramePipeline(model_p, pp_p, anchors, lanes=1, result_lag=0, graph=use_graph, device=dev, preprocessor=pre, points_per_cloud=slot,
                                    streams=lane_streams)
                d1, _, _ = timed_run(fp1, args.steps, max(4, args.warmup), batches=pframes, points=True)
                from_points["one_frame_in_flight"] = dict(latency_stats(fp1.latencies_ms), frames_per_s=round(args.steps / d1, 3))
                fp1.close()
            del model_p
        except Exception as e:      # noqa: BLE001  a second mode must never cost the headline line
            from_points = {"error": f"{type(e).__name__}: {str(e)[:300]}"}
            torch.cuda.synchronize()
    if hasattr(model, "pillar_vfe"):
        model.pillar_vfe.persistent_canvas = True             # (a closed side pipeline hands the flag back)

    # ---- the same bracket with the other convolution arithmetics (reported beside `value`, never as `value`)
    side = None
    if (not multi) and not args.no_side_modes:
        side = {}
        try:
            for terms in (0, 3, 16, 2):
                if terms == default_terms:
                    continue
                backbone_mod.CONV_EMU_TERMS = terms
                p2 = make_pipe(use_graph)
                d2, ti2, _ = timed_run(p2, args.steps, warm)
                side["native_fp32" if terms == 0 else "fp16x2" if terms == 16 else f"bf16x{terms}"] = {
                    "value": round(args.steps / d2, 3), "unit": "frames/s", "ms_per_step": round(d2 / args.steps * 1e3, 4),
                    "host_enqueue_ms_per_step": round(ti2 / args.steps * 1e3, 4)}
                del p2
            backbone_mod.CONV_EMU_TERMS = default_terms
            p2 = make_pipe(not use_graph)                   # the other launch mode, same arithmetic
            w2 = max(args.warmup, 2 * n_lanes)
            d2, ti2, _ = timed_run(p2, args.steps, w2)
            side["eager_launches" if use_graph else "hip_graph_replay"] = {
                "value": round(args.steps / d2, 3), "unit": "frames/s", "ms_per_step": round(d2 / args.steps * 1e3, 4),
                "host_enqueue_ms_per_step": round(ti2 / args.steps * 1e3, 4)}
            del p2
        except Exception as e:      # a side report must never cost the headline line
            side["error"] = f"{type(e).__name__}: {str(e)[:200]}"
            torch.cuda.synchronize()
        backbone_mod.CONV_EMU_TERMS = default_terms

    if rank == 0:
        ms_step = dt / args.steps * 1e3
        frames_per_step = world if mode in ("ring", "replicas") else 1        # gather mode: the ranks share ONE frame per step
        fps = frames_per_step * args.steps / dt
        M = int(frames[0]["processed_lidar"]["voxel_features"].shape[0])
        scales = [(64, ny // 2, nx // 2), (128, ny // 4, nx // 4), (256, ny // 8, nx // 8)]
        alg_bytes = {"pillar_vfe_scatter": M * (32 * 4 * 4 + 4 * 4 + 4 + 64 * 4) + N * 64 * ny * nx * 4}
        for C, H, W in scales:
            alg_bytes[f"warp_fuse_C{C}"] = (N + 1) * C * H * W * 4
        fuse_bytes = sum(alg_bytes[f"warp_fuse_C{C}"] for C, _, _ in scales)
        conv_flops = 2 * N * 64 * (ny // 2) * (nx // 2) * 64 * 9          # one 64 -> 64 channel 3x3 layer of the first ResNet stage
        alg_flops = {"conv3x3_bias_act": conv_flops, "conv3x3_emu_bias_act": conv_flops}
        kernels = []
        alg_live = dict(alg_bytes)
        if default_terms in (2, 3, 16):     # the pipeline's persistent canvas: bytes really moved per call (see pillar_moved_model below), not the dense-canvas formula
            alg_live["pillar_vfe_scatter"] = M * (532 + 256 + 256 + 256 + 12)
            alg_live["pillar_encode_sparse"] = pillar_bytes_sparse(M)
        for name, pairs in sorted(prof.items()):
            ms = sum(s.elapsed_time(e) for s, e in pairs) / len(pairs)
            b = alg_live.get(name)
            kernels.append({"name": name, "launches_timed": len(pairs), "avg_ms": round(ms, 5), "algorithmic_bytes": b,
                            "GBps": None if b is None else round(b / ms / 1e6, 1),
                            "frac_of_8TBps": None if b is None else round(b / ms / 1e6 / HBM_PEAK_GBPS, 4)})
        live = {k["name"]: k for k in kernels}

        # HBM traffic per launch from the committed rocprofv3 --pmc passes (separate passes, tools/gpu_round_artifacts.sh):
        # reads = 2 x FETCH_SIZE (gfx950 tallies the 128-B requests of 16 B/lane streaming loads at 64 B, MI355X_MICROARCH.md
        # "HBM"), writes = WRITE_SIZE; null when no summary is committed for this workload
        pmc, pmc_src = {}, None
        path = next((q for q in (os.path.join(ROOT, "profiles", r, "pmc_summary.json") for r in ("round6", "round5", "round4", "round3", "round2")) if os.path.exists(q)), "")
        if path and N == 5 and args.pillars == 8000 and args.config == "opv2v_coalign":
            pmc, pmc_src = json.load(open(path)), os.path.relpath(path, ROOT)

        def traffic_of(op):
            """corrected HBM bytes of one call of the op (all its kernels and memsets), profiles/roundN/pmc_summary.json"""
            e = pmc.get(op)
            return int(e["hbm_bytes_read_x2"]) if e and "hbm_bytes_read_x2" in e else None

        def hbm_entry(name, ms, nbytes, traffic, live_name=None):
            e = {"kernel": name, "bound": "hbm", "achieved": round(nbytes / ms / 1e6, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                 "frac": round(nbytes / ms / 1e6 / HBM_PEAK_GBPS, 4), "algorithmic_bytes_per_launch": nbytes, "avg_launch_ms": round(ms, 5),
                 "traffic": traffic, "traffic_over_algorithmic": None if not traffic else round(traffic / nbytes, 3)}
            if live_name in live:
                e["in_timed_steps"] = {"avg_launch_ms": live[live_name]["avg_ms"], "frac": live[live_name]["frac_of_8TBps"]}
            return e

        persistent = default_terms in (2, 3, 16)
        # ---- north star: the pillar-scatter + warp path against the HBM roofline.  ADVICE r03: `frac` = ALGORITHMIC bytes / time measured in this
        #      run, nothing else; HBM traffic from the committed rocprofv3 --pmc passes is reported beside it (`traffic`), never mixed into `frac`.
        #      Algorithmic bytes of the pillar op = what the timed configuration has to move per call (pillar_bytes_moved: the persistent canvas
        #      writes and clears rows, never a dense zero-fill); SURVEY 8d's formula, which counts a dense canvas write, is `frac_survey_8d_formula`
        #      beside the time of the SAME path with a freshly zero-filled canvas per call (rounds 1-2 reported that one).
        sparse_on = "pillar_dense_persistent_ms" in iso
        pillar_alg = pillar_bytes_sparse(M) if sparse_on else pillar_bytes_moved(M) if persistent else alg_bytes["pillar_vfe_scatter"]
        pillar = hbm_entry("pillar_vfe_scatter = pillar_sparse_kernel: ONE launch, feature rows + 8-byte cell stamps, no dense canvas, nothing cleared (csrc/pillar_sparse.hip)" if sparse_on else
                           "pillar_vfe_scatter = pillar_prep_kernel + pillar_rows_mx_kernel (persistent channels-last canvas: only the previous frame's rows are cleared)" if persistent
                           else "pillar_vfe_scatter = memset + cellmap_kernel + pillar_canvas_kernel", iso["pillar_ms"], pillar_alg,
                           traffic_of("pillar_sparse" if sparse_on else "pillar_nhwc_persistent" if persistent else "pillar_nchw"), "pillar_encode_sparse" if sparse_on else "pillar_vfe_scatter")
        if sparse_on:
            pillar["dense_persistent_canvas_form"] = {"avg_launch_ms": round(iso["pillar_dense_persistent_ms"], 5), "bytes_moved_model": pillar_bytes_moved(M),
                                                      "frac": round(pillar_bytes_moved(M) / iso["pillar_dense_persistent_ms"] / 1e6 / HBM_PEAK_GBPS, 4)}
        pillar["survey_8d_formula_bytes"] = alg_bytes["pillar_vfe_scatter"]
        pillar["read_bytes"] = M * 532
        pillar["frac_read_only"] = round(M * 532 / iso["pillar_ms"] / 1e6 / HBM_PEAK_GBPS, 4)
        north = {"target": 0.40, "pillar_vfe_scatter": pillar}
        if "pillar_fresh_canvas_ms" in iso:
            north["pillar_vfe_scatter_fresh_canvas"] = hbm_entry("pillar_vfe_scatter on a fresh canvas = canvas memset + cell-map memset + cellmap_kernel + pillar_rows_mx_kernel",
                                                                 iso["pillar_fresh_canvas_ms"], alg_bytes["pillar_vfe_scatter"], traffic_of("pillar_nhwc"))
        ktrace = kernel_trace_us()
        if ktrace:
            north["kernel_trace_us"] = ktrace
        if "fuse_ms" in iso:
            fuse = hbm_entry("warp + attention fusion, 3 scales (coalign_warp_fuse_nhwc: one launch)" if persistent else "coalign_warp_fuse x 3 scales",
                             iso["fuse_ms"], fuse_bytes, traffic_of("fuse_nhwc_3scales") if persistent else None)
            north["warp_fuse_all_scales"] = fuse
            tot_ms = iso["pillar_ms"] + iso["fuse_ms"]
            alg = pillar_alg + fuse_bytes
            north.update({"frac": round(alg / tot_ms / 1e6 / HBM_PEAK_GBPS, 4), "achieved": round(alg / tot_ms / 1e6, 1), "unit": "GB/s",
                          "algorithmic_bytes": alg, "ms": round(tot_ms, 5)})
            # north_star's literal wording is the HBM-READ roofline: bytes READ by the path (532 B per pillar; every agent's map at the three scales) / the same time
            read_bytes = M * 532 + sum(N * C * H * W * 4 for C, H, W in scales)
            north["read_bytes"] = read_bytes
            north["frac_read_only"] = round(read_bytes / tot_ms / 1e6 / HBM_PEAK_GBPS, 4)
            if pillar["traffic"] and fuse["traffic"]:      # the committed PMC passes of this workload (may predate a kernel change: see its README)
                north["traffic"] = pillar["traffic"] + fuse["traffic"]
                north["traffic_over_algorithmic"] = round(north["traffic"] / alg, 3)
                north["traffic_source"] = pmc_src
            if ktrace and "pillar_op_us" in ktrace and "fuse_us" in ktrace:      # the same bytes over the kernel durations of the committed rocprofv3 trace
                north["frac_kernel_trace"] = round(alg / ((ktrace["pillar_op_us"] + ktrace["fuse_us"]) * 1e-3) / 1e6 / HBM_PEAK_GBPS, 4)
            if "pillar_fresh_canvas_ms" in iso:      # SURVEY 8d's formula (dense canvas write counted) where a dense canvas is really written
                tot_b = alg_bytes["pillar_vfe_scatter"] + fuse_bytes
                north["frac_survey_8d_formula"] = round(tot_b / (iso["pillar_fresh_canvas_ms"] + iso["fuse_ms"]) / 1e6 / HBM_PEAK_GBPS, 4)
        else:
            north.update({"frac": pillar["frac"], "note": "fusion timing failed: " + iso.get("fuse_error", "?")})
        north["note"] = ("north_star: >= 40 % of the HBM roofline on the pillar-scatter + warp path.  Each part alone on the GPU (HIP events around "
                         "10 replays of a HIP graph of the op's launches, right before the timed region).  `frac` = algorithmic bytes of the timed "
                         "configuration / time of this run / 8 TB/s; `traffic` = corrected PMC bytes of the committed profile, reported separately; "
                         "`frac_survey_8d_formula` = SURVEY 8d's byte formula over the same path with a freshly zero-filled canvas per call; "
                         "`size_sweep` repeats the figures at 32 000 / 70 000 pillars per agent, N = 2 and DAIR geometry")

        # `roofline` = the hand-written kernel the frame spends most of its time in.  Round 5: with the SplitMap route (fp16 split) that is coalign_conv3x3_sp at
        # the stage-3 shape (256 -> 256 channels at ny/8 x nx/8, N agents: 15 launches per frame); `roofline_by_layer` lists all five 3x3 layer shapes.
        layer_rows = conv_layer_rooflines(dev, N, ny, nx, pmc) if (default_terms == 16 and backbone_mod.split_maps_active()) else None
        dom = next((r for r in (layer_rows or []) if r.get("layer", "").startswith("stage 3") and "us" in r), None)
        if dom is not None:
            executed = 3 * 2 * 9 * 256 * 256 * (ny // 8) * (nx // 8) * N
            roofline = {"kernel": f"coalign_conv3x3_sp (csrc/conv3x3_sp.hip: v_mfma_f32_32x32x16_f16 on sp16 pairs, operands by LDS-DMA from the producer's SplitMap; "
                                  f"256->256 channels at {ny // 8}x{nx // 8}, N={N}; {dom['launches_per_frame']} launches per frame)",
                        "bound": "mfma", "achieved": dom["executed_TFLOPs"], "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": dom["frac_of_fp16_peak"],
                        "avg_launch_ms": round(dom["us"] / 1e3, 5),
                        # ALGORITHMIC = the layer's fp32 multiply-adds; EXECUTED = the 16-bit products the kernel issues for them (3 per
                        # fp32 product).  `achieved` / `frac` price the executed products against the fp16 peak (what the matrix pipe does); `frac_algorithmic` prices
                        # the algorithmic flops against the same peak, `frac_algorithmic_of_fp32_mfma_peak` against the 157.3 TFLOP/s an fp32-MFMA kernel is bounded by
                        "algorithmic_flops_per_launch": executed // 3, "executed_flops_per_launch": executed,
                        "frac_algorithmic": round(executed / 3 / (dom["us"] * 1e-6) / 1e12 / BF16_MFMA_PEAK_TFLOPS, 4),
                        "frac_algorithmic_of_fp32_mfma_peak": round(executed / 3 / (dom["us"] * 1e-6) / 1e12 / F32_MFMA_PEAK_TFLOPS, 3),
                        "fp32_equivalent_TFLOPs": dom["fp32_equivalent_TFLOPs"], "traffic": int(dom["pmc_hbm_MB"] * 1e6) if "pmc_hbm_MB" in dom else None,
                        "note": "executed 16-bit products = 3 per fp32 product (w_h x_h, w_h x_l, w_l x_h of 22-bit operands); peak = dense fp16 MFMA (MI355X_MICROARCH.md); "
                                "192 tiles of 16 intervals on 256 CUs: alone on the GPU a quarter of the CUs idle (in the 3-lane pipeline other frames' kernels take them)"}
            if "pmc_mfma_busy_share" in dom:
                roofline["pmc_mfma_busy_share"] = dom["pmc_mfma_busy_share"]
        elif default_terms in (2, 3, 16):
            t = default_terms
            ms = iso["conv_fp16x2_ms" if t == 16 else f"conv_bf16x{t}_ms"]
            executed = conv_flops * (6 if t == 3 else 3)
            roofline = {"kernel": (f"conv3x3_emu_bias_act (v_mfma_f32_32x32x16_f16, fp32 operands as sp16 pairs split by the consumer, 64->64 channels at {ny // 2}x{nx // 2}, N={N})" if t == 16 else
                                   f"conv3x3_emu_bias_act (v_mfma_f32_32x32x16_bf16, fp32 operands split {t}-way, 64->64 channels at {ny // 2}x{nx // 2}, N={N})"),
                        "bound": "mfma", "achieved": round(executed / ms / 1e9, 1), "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(executed / ms / 1e9 / BF16_MFMA_PEAK_TFLOPS, 4), "avg_launch_ms": round(ms, 5),
                        "algorithmic_flops_per_launch": conv_flops, "executed_flops_per_launch": executed,
                        "frac_algorithmic": round(conv_flops / ms / 1e9 / BF16_MFMA_PEAK_TFLOPS, 4),
                        "fp32_equivalent_TFLOPs": round(conv_flops / ms / 1e9, 1), "traffic": traffic_of(f"conv_bf16x{t}_64ch"),
                        "note": f"executed 16-bit products = {6 if t == 3 else 3} per fp32 product; peak = dense bf16 / fp16 MFMA (MI355X_MICROARCH.md)"}
        else:
            ms = iso["conv_f32_ms"]
            roofline = {"kernel": f"conv3x3_bias_act (v_mfma_f32_32x32x2_f32 implicit GEMM, 64->64 channels at {ny // 2}x{nx // 2}, N={N})", "bound": "mfma",
                        "achieved": round(conv_flops / ms / 1e9, 2), "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(conv_flops / ms / 1e9 / F32_MFMA_PEAK_TFLOPS, 4), "avg_launch_ms": round(ms, 5),
                        "algorithmic_flops_per_launch": conv_flops, "traffic": traffic_of("conv_f32_64ch")}
        roofline["isolated_ms"] = {k: round(v, 5) for k, v in iso.items() if k.endswith("_ms")}
        roofline["hbm_bound_kernel"] = pillar

        if default_terms == 16:
            dtype = ("f32 (3x3 convolutions: every operand rounded to 22 significant bits and split into two fp16 terms, the low term scaled by 2^10 and the "
                     "weights by a per-output-channel power of two so that both terms are normal fp16 numbers at any scale; three fp16 matrix products per fp32 "
                     "product, f32 accumulation; maps between the 3x3 layers of a stage are stored as such 22-bit pairs; error against float64 <= the native fp32 "
                     "matrix kernel's over weight scales 2e-4..2e-1 x activation scales 1e-2..1e2, tests/test_round5_gpu.py)")
        else:
            dtype = "f32" if default_terms == 0 else (f"f32 (3x3 convolution products evaluated as {default_terms}-way split bf16 products on the bf16 matrix cores, "
                                                      "f32 accumulation" + ("; dropped terms <= 2^-24 |w x|, i.e. fp32-width arithmetic)" if default_terms == 3 else ")"))
        result = {
            "metric": "frames_per_s_5agent_opv2v_synthetic", "value": round(fps, 3), "unit": "frames/s", "n_gpus": world,
            "steps": args.steps, "warmup": warm, "ms_per_step": round(ms_step, 4), "higher_is_better": True,
            "scaling": "weak" if mode in ("ring", "replicas") else "strong", "vs_baseline": None, "dtype": dtype, "data": "synthetic",
            "config": {"workload": f"OPV2V PointPillar + CoAlign multiscale attention fusion ({args.config}.yaml, BASELINE configs[2] "
                                   f"geometry): {N} agents/frame, {args.pillars} pillars/agent, canvas {nx}x{ny}, 70400 anchors, "
                                   f"full path incl. decode + rotated NMS, {pool_n} distinct frames in rotation",
                       "agents_per_frame": N, "pillars_per_agent": args.pillars, "frames_per_step": frames_per_step, "frames_in_flight": n_lanes, "streams": n_streams, "frames_queued_per_stream": n_lanes // n_streams,
                       "result_lag_frames": pipe.result_lag, "hip_graph": use_graph, "lane_capture_frames": capture_frames,
                       "inputs": ("resident in HBM, read in place through a 32-byte device record; pose matrices normalised on the host; one small host-to-device transfer per frame"
                                  if pipe.frames_in_place and not pipe.frames_copied else "resident in HBM, copied into the captured graph's input buffers every frame" if use_graph else "resident in HBM"), "conv_arithmetic": "native fp32" if default_terms == 0 else ("fp16 split (sp16 pairs)" + (", SplitMaps between the 3x3 layers" if backbone_mod.split_maps_active() else "")) if default_terms == 16 else f"bf16x{default_terms}",
                       "parallelism": "single GPU" if (not multi) else (f"agent-sharded frame ring x{world}, {'RCCL' if backend == 'nccl' else backend + ' (functional test)'} all-to-all" if mode == "ring" else
                                                                                      f"{world} independent replicas, no collective (fall-back: see `rccl.fallbacks`)" if mode == "replicas" else
                                                                                      f"one frame over {world} ranks (agent blocks), {'RCCL' if backend == 'nccl' else backend + ' (functional test)'} all-gather, ego tail on every rank") +
                                       (", one communicator per lane" if args.comm_per_lane else ", one communicator"),
                       "detections_last_frame": 0 if last_boxes is None else int(last_boxes.shape[0]),
                       "candidates_last_frame": pp.last_counts["candidates"]},
            "roofline": roofline, "roofline_by_layer": layer_rows, "north_star_hbm": north, "kernels": kernels,
            "host_enqueue_ms_per_step": round(t_issue / args.steps * 1e3, 4),
            "frame_digests": {str(k): digests[k] for k in sorted(digests)}, "frame_digests_reproducible": bool(consistent),
            "frame_digest_mismatches": mismatches[:8],
            "split_map_range_exceeded": bool(range_hit),
        }
        if rings is not None:
            result["exchange_bytes_sent_per_rank_per_step"] = rings[0].bytes_sent_last
        if rccl is not None:
            result["rccl"] = rccl
        if latency is not None:
            result["latency_ms"] = latency
        if from_points is not None:
            result["from_points"] = from_points
        if side is not None:
            result["other_modes"] = side
        if (not multi) and not args.no_size_sweep:
            pipe.close()
            del pipe
            torch.cuda.empty_cache()
            result["size_sweep"] = size_sweep(dev, streams=lane_streams, lanes=n_streams if use_graph else 3, result_lag=result_lag if use_graph else 1, queue_depth=queue_depth if use_graph else 1)
        if (not multi) and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(hypes, model, frames_cpu, anchors, args.cpu_frames, args.cpu_threads, args.cpu_budget_s)
        if (not multi) and not args.no_numerics:
            # checker leg (like cpu_baseline: rank 0, N = 1, outside every timed region): the benchmarked geometry end to end against a FLOAT64 evaluation of the
            # reference's forward, in the default arithmetic, bf16 x 3 and native fp32, beside the reference's own fp32 (tests/numerics_table.py)
            try:
                sys.path.insert(0, os.path.join(ROOT, "tests"))
                import numerics_table
                torch.cuda.empty_cache()
                rows = [numerics_table.measure("cfg3", prm, dev, threads=min(32, os.cpu_count() or 1)) for prm in ("random_init", "trained_like")]
                result["numerics"] = {"against": "oracle.coalign_forward(dtype=float64) on the same float32 inputs and parameters; errors as a fraction of max |float64 tensor|, worst head",
                                      "rows": numerics_table.summarize(rows)["rows"], "per_head": rows,
                                      "full_table": "profiles/round6/numerics.json (cfg 2 / 3 / 4 x random-init / trained-like; tests/test_round6_gpu.py asserts it)"}
            except Exception as e:      # noqa: BLE001  a side report must never cost the headline line
                result["numerics"] = {"error": f"{type(e).__name__}: {str(e)[:200]}"}
        print(json.dumps(result), flush=True)
    if multi:
        dist.barrier()
        dist.destroy_process_group()


def cpu_model_string():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(hypes, model, frames_cpu, anchors, n_frames, threads, budget_s):
    """The CPU oracle (numpy / torch-CPU restatement of the reference path, oracle/) on the SAME frames:
    1 warm-up frame, then up to ``n_frames`` timed frames rotating over the pool, stopping once ``budget_s`` seconds of timed
    work are spent (the default bench run has to finish within minutes).  Threads: the best setting of the committed sweep
    (profiles/round2/cpu_baseline_sweep.json, tools/cpu_baseline_sweep.py) -- one torch thread per core is pathologically slow
    for the tiny batched matmuls of the attention fusion on a many-core host."""
    from oracle import coalign_oracle as oracle
    sweep_src, sweep_tab = None, None
    import glob as _glob
    found = sorted(_glob.glob(os.path.join(ROOT, "profiles", "round*", "cpu_baseline_sweep.json")), key=lambda q: int("".join(c for c in os.path.basename(os.path.dirname(q)) if c.isdigit()) or 0))
    if found:                                    # the NEWEST committed sweep (tools/cpu_baseline_sweep.py on the current oracle, re-run every round the oracle changes)
        sweep_src = os.path.relpath(found[-1], ROOT)
        sw = json.load(open(found[-1]))
        sweep_tab = {k: v.get("frames_per_s") for k, v in sw.get("threads", {}).items()}
    if threads <= 0:
        threads = int(sw.get("best_threads", 16)) if found else 16
    cores = max(1, min(threads, os.cpu_count() or 1))
    torch.set_num_threads(cores)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    margs = hypes["model"]["args"]

    def one(frame):
        with torch.no_grad():
            out = oracle.coalign_forward(sd, margs, frame)
            oracle.post_process([out], anchors, hypes["postprocess"])

    one(frames_cpu[0])                          # warm-up (allocator, thread pool)
    done, t0 = 0, time.perf_counter()
    while done < n_frames and (done == 0 or time.perf_counter() - t0 < budget_s):
        one(frames_cpu[done % len(frames_cpu)])
        done += 1
    dt = time.perf_counter() - t0
    return {"value": round(done / dt, 4), "unit": "frames/s", "cores": cores, "kind": "port", "cpu_model": cpu_model_string(),
            "host_cores": os.cpu_count(), "thread_sweep_frames_per_s": sweep_tab, "thread_sweep_source": sweep_src,
            "sample": f"1 warm-up + {done} timed frame(s) of the same 5-agent synthetic pool ({dt:.1f} s, cap {budget_s:.0f} s), torch CPU threads = {cores} "
                      f"of {os.cpu_count()} host cores = the best count of the committed thread sweep ({sweep_src}: one torch thread per core is pathologically slow "
                      f"for the oracle's per-pixel batched matmuls; the sweep includes os.cpu_count() threads)"}


if __name__ == "__main__":
    main()
