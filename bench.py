#!/usr/bin/env python
"""Benchmark of the CoAlign per-frame detection hot path on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): frames/s on synthetic 5-agent OPV2V-shaped scenes.  A step is one pass of the whole hot
path (pillar encode + scatter -> BEV backbone -> pose-aware warp + attention fusion at 3 scales -> heads ->
decode + rotated NMS) over one frame per rank; inputs are resident in HBM before the timed region.  With R ranks
a step processes R frames in the agent-sharded "frame ring" schedule of coalign_amd/sharded.py (weak scaling).
Rank 0 prints ONE JSON line; it also carries the roofline of the dominant hand-written kernel (the fp32 matrix-core
convolution; the HBM-bound pillar encoder rides along as `hbm_bound_kernel`; both timed with HIP events inside the timed steps) and, at N=1, the CPU oracle timed on the host cores.
"""
import argparse
import json
import math
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from coalign_amd import ops  # noqa: E402
from coalign_amd.config import builtin_config  # noqa: E402
from coalign_amd.detector import build_model, to_device  # noqa: E402
from coalign_amd.postprocess import build_postprocessor  # noqa: E402
from coalign_amd.sharded import FrameRing, encode_assignments  # noqa: E402
from coalign_amd.synthetic import fill_parameters_, make_frame  # noqa: E402

F32_MFMA_PEAK_TFLOPS = 157.3      # dense fp32 matrix peak, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)


def calibrate_cls_bias(model, pp, batch, target=600):
    """Random-init heads give arbitrary logits; shift cls_head.bias so that ~`target` anchors pass the score
    threshold ("head biases shifted so K ~ 300-1000 candidates"), as a trained detector would."""
    with torch.no_grad():
        out = model(batch)
        # box deltas of a trained detector are small: rescale the random regression head to std 0.1 so that the decoded
        # boxes stay car sized / inside the z range (they pass the sanity filters) and neighbouring anchors overlap
        model.reg_head.weight *= 0.1 / float(out["reg_preds"].std())
        model.reg_head.bias.zero_()
        logits = out["cls_preds"].flatten()
        k = min(target, logits.numel() - 1)
        v = torch.topk(logits, k + 1).values[-1]
        thr = pp.params["target_args"]["score_threshold"]
        model.cls_head.bias += (math.log(thr / (1 - thr)) - float(v))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--agents", type=int, default=5)
    ap.add_argument("--pillars", type=int, default=8000)
    ap.add_argument("--config", default="opv2v_coalign")
    ap.add_argument("--lanes", type=int, default=4, help="frames in flight on separate HIP streams")
    ap.add_argument("--result-lag", type=int, default=1, help="frames between enqueueing a frame and collecting its detections on the host "
                    "(1 = collect the previous frame's; deeper lags were measured no faster)")
    ap.add_argument("--no-miopen-find", action="store_true", help="leave torch.backends.cudnn.benchmark off (MIOpen immediate mode)")
    ap.add_argument("--graph", action="store_true", help="capture encode + fuse + heads of a frame into one HIP graph per lane and replay it "
                    "(single GPU; the synthetic frame has a fixed shape) -- removes the ~150 host-side launches per frame")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--conv-emu", type=int, default=0, choices=(0, 2, 3),
                    help="measure with the opt-in split-bf16 3x3 convolutions (COALIGN_CONV_EMU) instead of the native-fp32 default")
    ap.add_argument("--no-opt-in", action="store_true", help="skip the extra timed passes of the opt-in convolution modes")
    ap.add_argument("--cpu-frames", type=int, default=3)
    ap.add_argument("--cpu-threads", type=int, default=16, help="torch CPU threads for the oracle (tiny batched matmuls "
                    "get slower, not faster, with one thread per core on a many-core host)")
    ap.add_argument("--cpu-budget-s", type=float, default=30.0)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # COALIGN_BENCH_BACKEND=gloo + COALIGN_BENCH_ONE_GPU=1: functional test of the multi-rank code path on a single
    # GPU (ranks share device 0, the exchange is staged through host memory).  Never used for reported numbers.
    backend = os.environ.get("COALIGN_BENCH_BACKEND", "nccl")
    if os.environ.get("COALIGN_BENCH_ONE_GPU") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    hypes = builtin_config(args.config)
    N = args.agents
    nx, ny, _ = [int(v) for v in hypes["model"]["args"]["point_pillar_scatter"]["grid_size"]]
    model = build_model(hypes)
    fill_parameters_(model, seed=0)
    model = model.to(dev).eval()
    pp = build_postprocessor(hypes["postprocess"], False)
    anchors = torch.from_numpy(pp.generate_anchor_box())
    ego_meta = {"ego": {"transformation_matrix": torch.eye(4, device=dev), "anchor_box": anchors}}

    # My frame (I am its ego): poses + the point clouds of the agents I encode this step.  In the ring each rank
    # encodes agent a of frame (rank - a) mod R; synthetic frames are i.i.d., so a rank simply generates N agents'
    # pillars plus its own frame's pose matrices -- same bytes, same work as the routed real thing.
    frame_cpu = make_frame(hypes, N, pillars_per_agent=args.pillars, seed=303 + rank, noise=(0.2, 0.2))
    frame = to_device(frame_cpu, dev)
    frame["record_len"] = frame_cpu["record_len"]        # host-side agent counts: no device->host sync per frame
    calibrate_cls_bias(model, pp, frame)
    if world > 1:   # identical weights everywhere
        for p in model.parameters():
            if backend == "nccl":
                dist.broadcast(p.data, 0)
            else:
                buf = p.data.cpu()
                dist.broadcast(buf, 0)
                p.data.copy_(buf)

    record = [N]
    import collections
    pending = collections.deque()          # post-process handles whose results have not been collected yet
    last = [(None, None)]

    def flush():
        while pending:
            last[0] = pending.popleft().result()
        return last[0]

    # Frames are independent, so consecutive frames go to alternating HIP streams ("lanes"): the tail of one frame's kernels
    # (partial last waves of every convolution launch, the small latency-bound fusion / head kernels) overlaps the other
    # frame's work.  Throughput device; every frame still completes inside the timed bracket (device-wide synchronise).
    n_lanes = max(1, args.lanes)
    result_lag = max(0, args.result_lag)
    pp.buffer_sets = result_lag + 2        # decode / NMS buffer sets: one per uncollected frame + the one being filled
    lanes = [torch.cuda.Stream(device=dev) for _ in range(n_lanes)] if n_lanes > 1 else [None]
    # one exchange buffer set per lane (a lane's all-to-all may still be in flight when the next lane packs); the collectives of
    # all lanes go to the one communicator in frame order on every rank, ProcessGroupNCCL serialises them on its own stream
    rings = [FrameRing(N) for _ in range(n_lanes)] if world > 1 else None
    counter = [0]

    def step():
        k = counter[0] % len(lanes)
        counter[0] += 1
        if lanes[k] is None:
            return _step(k)
        with torch.cuda.stream(lanes[k]):
            return _step(k)

    graphs = {}                              # lane -> (HIP graph of encode + fuse + heads, its static output dict)

    def capture(k):
        # same launches, recorded once on the lane's stream; replay re-issues them with one host call.  The frame's tensors and
        # the head outputs are static buffers of the graph; decode + NMS stay outside (their sizes are data-dependent).
        g = torch.cuda.CUDAGraph()
        with torch.no_grad():
            with torch.cuda.graph(g, stream=lanes[k]):
                feats, affine = model.encode(frame)
                out = model.fuse_and_head(feats, record, affine)
        graphs[k] = (g, out)

    def _step(k):
        ring = rings[k] if rings is not None else None
        with torch.no_grad():
            if k in graphs:
                g, out = graphs[k]
                g.replay()
                handle = pp.post_process_async(ego_meta, {"ego": out})
                pending.append(handle)
                while len(pending) > result_lag:
                    last[0] = pending.popleft().result()
                return last[0]
            with ops.timed("stage_encode(pillars+backbone)"):
                feats, affine = model.encode(frame)
            if ring is not None:
                with ops.timed("stage_exchange(all_to_all)"):
                    feats = ring.exchange(feats)
            with ops.timed("stage_fuse_and_heads"):
                out = model.fuse_and_head(feats, record, affine)
            with ops.timed("stage_post_process(enqueue)"):
                # decode + NMS run on a side stream and overlap the next frame's encoder; the previous frame's
                # results are collected `result_lag` frames later (software pipeline, flushed before the timed region closes)
                handle = pp.post_process_async(ego_meta, {"ego": out})
            pending.append(handle)
            while len(pending) > result_lag:       # collect the oldest frame's detections (host wait for that frame only)
                last[0] = pending.popleft().result()
            return last[0]

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # MIOpen picks each convolution's solver by measurement during the warm-up (torch's cudnn.benchmark = miopenFind*): on a
    # fresh machine the immediate-mode fallback, used when the user find-db is empty, is 10 % slower than what a second run gets
    torch.backends.cudnn.benchmark = not args.no_miopen_find
    from coalign_amd import backbone as backbone_mod
    backbone_mod.CONV_EMU_TERMS = args.conv_emu          # 0 unless asked for: the headline number is the native-fp32 path
    for _ in range(args.warmup):
        step()
    flush()
    sync()
    if args.graph and world == 1 and lanes[0] is not None:
        for k in range(len(lanes)):
            capture(k)
        for _ in range(len(lanes)):
            step()
        flush()
        sync()
    # the roofline kernel alone on the GPU (outside the timed region): with several frames in flight the live launch durations
    # below include time-sharing with the other lanes' kernels, so both figures are reported
    iso_ms = iso_pillar_ms = iso_fuse_ms = None
    if rank == 0:
        with torch.no_grad():
            gx = torch.randn(N, 64, ny // 2, nx // 2, device=dev)
            gw = ops.pack_conv3x3_weight(torch.randn(64, 64, 3, 3, device=dev) / 24.0)
            gb, gr = torch.randn(64, device=dev), torch.randn(N, 64, ny // 2, nx // 2, device=dev)
            for _ in range(3):
                ops.conv3x3_bias_act(gx, gw, gb, gr, True)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ops.conv3x3_bias_act(gx, gw, gb, gr, True)
            e1.record()
            torch.cuda.synchronize()
            iso_ms = e0.elapsed_time(e1) / 10
            del gx, gw, gb, gr
            # the HBM-bound pillar encoder (memset + cellmap + pillar_canvas) alone on the GPU, same bracket
            pl_in = dict(frame["processed_lidar"], record_len=frame["record_len"])
            for _ in range(3):
                model.pillar_vfe(dict(pl_in))
            e0.record()
            for _ in range(10):
                model.pillar_vfe(dict(pl_in))
            e1.record()
            torch.cuda.synchronize()
            iso_pillar_ms = e0.elapsed_time(e1) / 10
            try:
                # ... and the warp + attention fusion of all three scales (launched as the model launches them), same bracket
                # (the three launches + their stream fork / join cost more host time than GPU time, so the GPU-side duration is taken
                #  from ten replays of a HIP graph of exactly these launches)
                feats_iso, affine_iso = model.encode(frame)
                for _ in range(3):
                    model._fuse_scales(list(feats_iso), record, affine_iso)
                torch.cuda.synchronize()
                if world == 1:
                    gs = torch.cuda.Stream(device=dev)
                    gs.wait_stream(torch.cuda.current_stream(dev))
                    fg = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(fg, stream=gs):
                        fused_iso = model._fuse_scales(list(feats_iso), record, affine_iso)
                    torch.cuda.synchronize()
                    replay = fg.replay
                else:       # no stream capture next to a live RCCL communicator (its watchdog thread polls events): plain launches
                    replay = lambda: model._fuse_scales(list(feats_iso), record, affine_iso)
                replay()
                torch.cuda.synchronize()
                e0.record()
                for _ in range(10):
                    replay()
                e1.record()
                torch.cuda.synchronize()
                iso_fuse_ms = e0.elapsed_time(e1) / 10
                del feats_iso, affine_iso, replay
            except Exception:       # measurement aid only: the bench line must survive it
                iso_fuse_ms = None
                torch.cuda.synchronize()
    sync()
    ops.PROFILE = {}
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    t_issue = time.perf_counter() - t0   # host time to enqueue the K frames (the GPU may still be working)
    boxes, scores = flush()          # the last frame's detections: all K frames are complete inside the bracket
    sync()
    dt = time.perf_counter() - t0
    prof, ops.PROFILE = ops.PROFILE, None
    if graphs:      # graph replays bypass the per-op event hooks: take the per-kernel profile from ordinary steps right after the region
        saved = dict(graphs)
        graphs.clear()
        ops.PROFILE = {}
        for _ in range(2 * n_lanes):
            step()
        flush(); sync()
        prof, ops.PROFILE = ops.PROFILE, None
        graphs.update(saved)
    # the same bracket with the opt-in split-bf16 3x3 convolutions (reported beside `value`, never as `value`)
    opt_in = None
    if world == 1 and args.conv_emu == 0 and not args.no_opt_in:
        try:
            opt_in = {}
            for terms in (3, 2):
                backbone_mod.CONV_EMU_TERMS = terms
                had_graphs = bool(graphs)
                graphs.clear()
                for _ in range(max(args.warmup, n_lanes + 1)):
                    step()
                flush(); sync()
                if had_graphs:
                    for k in range(len(lanes)):
                        capture(k)
                    for _ in range(len(lanes)):
                        step()
                    flush(); sync()
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    step()
                ti = time.perf_counter() - t1
                flush(); sync()
                d = time.perf_counter() - t1
                opt_in[f"bf16x{terms}"] = {"value": round(args.steps / d, 3), "unit": "frames/s", "ms_per_step": round(d / args.steps * 1e3, 4),
                                           "host_enqueue_ms_per_step": round(ti / args.steps * 1e3, 4)}
            backbone_mod.CONV_EMU_TERMS = 0
            opt_in["note"] = ("COALIGN_CONV_EMU=3|2: every 3x3 convolution through coalign_conv3x3_emu_bias_act -- fp32 operands split "
                              "error-free into 3 (2) bf16 terms, 6 (3) cross products on v_mfma_f32_32x32x16_bf16, fp32 accumulation; "
                              "conv error vs fp64 1.1e-6..2.4e-6 (x3) / 2.7e-6..3.6e-6 (x2) of the output scale against 1.8e-6..3.6e-6 for "
                              "the native fp32-MFMA kernel; end-to-end head outputs within 5.7e-6 (x3) / 3.8e-5 (x2) of eager PyTorch "
                              "(north-star tolerance 1e-3).  Not the default: `value` keeps native fp32 products.")
        except Exception as e:      # a side report must never cost the headline line
            backbone_mod.CONV_EMU_TERMS = 0
            graphs.clear()
            opt_in = {"error": f"{type(e).__name__}: {str(e)[:200]}"}
            torch.cuda.synchronize()
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        ms_step = dt / args.steps * 1e3
        fps = world * args.steps / dt
        M = int(frame["processed_lidar"]["voxel_features"].shape[0])
        scales = [(64, ny // 2, nx // 2), (128, ny // 4, nx // 4), (256, ny // 8, nx // 8)]
        alg_bytes = {"pillar_vfe_scatter": M * (32 * 4 * 4 + 4 * 4 + 4 + 64 * 4) + N * 64 * ny * nx * 4}
        for C, H, W in scales:
            alg_bytes[f"warp_fuse_C{C}"] = (N + 1) * C * H * W * 4
        # the hand-written convolution serves the 64 -> 64 channel layers of the first ResNet stage: 2 * N * Cout * H * W * Cin * 9 flops
        alg_flops = {"conv3x3_bias_act": 2 * N * 64 * (ny // 2) * (nx // 2) * 64 * 9}
        kernels = []
        for name, pairs in sorted(prof.items()):
            ms = sum(s.elapsed_time(e) for s, e in pairs) / len(pairs)
            b, fl = alg_bytes.get(name), alg_flops.get(name)
            kernels.append({"name": name, "launches_timed": len(pairs), "avg_ms": round(ms, 5), "algorithmic_bytes": b,
                            "GBps": None if b is None else round(b / ms / 1e6, 1),
                            "frac_of_8TBps": None if b is None else round(b / ms / 1e6 / HBM_PEAK_GBPS, 4),
                            "algorithmic_flops": fl, "TFLOPs": None if fl is None else round(fl / ms / 1e9, 2),
                            "frac_of_157TFLOPs": None if fl is None else round(fl / ms / 1e9 / F32_MFMA_PEAK_TFLOPS, 4)})
        dom = max((k for k in kernels if k["algorithmic_bytes"]), key=lambda k: k["avg_ms"])
        dom_mfma = max((k for k in kernels if k["algorithmic_flops"]), key=lambda k: k["avg_ms"], default=None)
        # HBM traffic per launch from the committed rocprofv3 --pmc passes (FETCH_SIZE + WRITE_SIZE, tools/gpu_round_artifacts.sh);
        # it cannot be sampled from inside the process, so it is read from profiles/ and is null when that file is absent
        traffic, traffic_src = None, None
        pmc_kernels = {"pillar_vfe_scatter": ["pillar_canvas_kernel", "cellmap_kernel"], "warp_fuse_C64": ["warp_fuse_kernel<5, 8, true, 512, 8>"],
                       "warp_fuse_C128": ["warp_fuse_kernel<5, 8, true, 1024, 8>"], "warp_fuse_C256": ["warp_fuse_kernel<5, 16, true, 1024, 4>"]}
        pmc_path = os.path.join(ROOT, "profiles", "round1", "final_pmc_summary.json")
        if os.path.exists(pmc_path) and N == 5 and args.pillars == 8000 and args.config == "opv2v_coalign":
            pmc = json.load(open(pmc_path))
            names = pmc_kernels.get(dom["name"], [])
            if names and all(n in pmc and "hbm_bytes_raw" in pmc[n] for n in names):
                traffic = int(sum(pmc[n]["hbm_bytes_raw"] for n in names))
                traffic_src = "profiles/round1/final_pmc_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, uncorrected sum)"
        hbm_ms = iso_pillar_ms if (iso_pillar_ms and dom["name"] == "pillar_vfe_scatter") else dom["avg_ms"]
        roofline_hbm = {"kernel": dom["name"] + (" = memset + cellmap_kernel + pillar_canvas_kernel" if dom["name"] == "pillar_vfe_scatter" else ""),
                    "bound": "hbm", "achieved": round(dom["algorithmic_bytes"] / hbm_ms / 1e6, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": round(dom["algorithmic_bytes"] / hbm_ms / 1e6 / HBM_PEAK_GBPS, 4), "traffic": traffic, "traffic_source": traffic_src,
                    "algorithmic_bytes_per_launch": dom["algorithmic_bytes"], "avg_launch_ms": round(hbm_ms, 5),
                    "in_timed_region": {"avg_launch_ms": dom["avg_ms"], "achieved": dom["GBps"], "frac": dom["frac_of_8TBps"]},
                    "note": "avg_launch_ms / achieved: HIP events around 10 calls of the op alone on the GPU right before the timed region (the "
                            "figure comparable to the committed rocprofv3 kernel trace); in_timed_region: the same op inside the timed steps, "
                            "where it shares the GPU with the other frames in flight and the previous frame's decode + NMS"}
        if iso_fuse_ms and iso_pillar_ms and dom["name"] == "pillar_vfe_scatter":
            fuse_bytes = sum(alg_bytes[f"warp_fuse_C{C}"] for C, _, _ in scales)
            roofline_hbm["pillar_plus_warp_path"] = {
                "warp_fuse_all_scales": {"algorithmic_bytes": fuse_bytes, "ms": round(iso_fuse_ms, 5), "achieved": round(fuse_bytes / iso_fuse_ms / 1e6, 1),
                                         "frac": round(fuse_bytes / iso_fuse_ms / 1e6 / HBM_PEAK_GBPS, 4)},
                "combined": {"algorithmic_bytes": dom["algorithmic_bytes"] + fuse_bytes, "ms": round(iso_pillar_ms + iso_fuse_ms, 5),
                             "achieved": round((dom["algorithmic_bytes"] + fuse_bytes) / (iso_pillar_ms + iso_fuse_ms) / 1e6, 1),
                             "frac": round((dom["algorithmic_bytes"] + fuse_bytes) / (iso_pillar_ms + iso_fuse_ms) / 1e6 / HBM_PEAK_GBPS, 4)},
                "note": "north_star's pillar-scatter + warp path, each part alone on the GPU (HIP events around 10 calls before the timed region); "
                        "the three fusion scales are launched the way the model launches them (finest scale on the main stream, the other two on side streams), "
                        "replayed from a HIP graph so that host launch overhead does not enter the GPU-side duration"}
        conv_traffic = None
        if os.path.exists(pmc_path) and N == 5 and args.config == "opv2v_coalign":
            for k, v in json.load(open(pmc_path)).items():
                if k.startswith("conv3x3_kernel") and "hbm_bytes_raw" in v:
                    conv_traffic = int(v["hbm_bytes_raw"])
        # `roofline` = the hand-written kernel with the longest launches: the matrix-core convolution when the model routes layers
        # through it (per launch 2x the pillar encoder's time), else the HBM-bound pillar encoder; the other one rides along
        if dom_mfma is not None and dom_mfma["avg_ms"] >= dom["avg_ms"]:
            live = {"avg_launch_ms": dom_mfma["avg_ms"], "achieved": dom_mfma["TFLOPs"], "frac": dom_mfma["frac_of_157TFLOPs"],
                    "launches_timed": dom_mfma["launches_timed"]}
            use_ms = iso_ms if iso_ms else dom_mfma["avg_ms"]
            roofline = {"kernel": "conv3x3_bias_act (v_mfma_f32_32x32x2_f32 implicit GEMM, 64->64 channels at %dx%d, N=%d)" % (ny // 2, nx // 2, N),
                        "bound": "mfma", "achieved": round(dom_mfma["algorithmic_flops"] / use_ms / 1e9, 2), "peak": F32_MFMA_PEAK_TFLOPS,
                        "unit": "TFLOP/s", "frac": round(dom_mfma["algorithmic_flops"] / use_ms / 1e9 / F32_MFMA_PEAK_TFLOPS, 4),
                        "traffic": conv_traffic, "algorithmic_flops_per_launch": dom_mfma["algorithmic_flops"], "avg_launch_ms": round(use_ms, 5),
                        "in_timed_region": live,
                        "note": "fp32 matrix peak 157.3 TFLOP/s (MI355X_MICROARCH.md).  avg_launch_ms / achieved: HIP events around 10 launches of the "
                                "kernel alone on the GPU, taken inside bench.py right before the timed region -- the figure that matches the "
                                "kernel's duration in the committed rocprofv3 --kernel-trace summary.  in_timed_region: HIP events around the same "
                                "launches inside the timed steps; with several frames in flight on separate streams an event pair also spans the "
                                "time the launch waits for / shares compute units with the other lanes' kernels, so it is not the kernel's duration.  "
                                "traffic = FETCH_SIZE + WRITE_SIZE of the committed rocprofv3 --pmc pass (algorithmic: 45 MB in, 45 MB residual in, "
                                "45 MB out per launch)",
                        "hbm_bound_kernel": roofline_hbm}
        else:
            roofline = roofline_hbm
        result = {
            "metric": "frames_per_s_5agent_opv2v_synthetic", "value": round(fps, 3), "unit": "frames/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.conv_emu == 0 else f"f32 (3x3 convolutions: bf16x{args.conv_emu} split products, f32 accumulate)", "data": "synthetic",
            "config": {"workload": f"OPV2V PointPillar + CoAlign multiscale attention fusion ({args.config}.yaml, BASELINE configs[2] "
                                   f"geometry): {N} agents/frame, {args.pillars} pillars/agent, canvas {nx}x{ny}, 70400 anchors, "
                                   "full path incl. decode + rotated NMS",
                       "agents_per_frame": N, "pillars_per_agent": args.pillars, "frames_per_step": world, "frames_in_flight": n_lanes, "result_lag_frames": result_lag, "hip_graph": bool(graphs),
                       "parallelism": "single GPU" if world == 1 else f"agent-sharded frame ring x{world}, {'RCCL' if backend == 'nccl' else backend + ' (functional test)'} all-to-all",
                       "detections_last_frame": 0 if boxes is None else int(boxes.shape[0]),
                       "candidates_last_frame": pp.last_counts["candidates"]},
            "roofline": roofline, "kernels": kernels,
            "host_enqueue_ms_per_step": round(t_issue / args.steps * 1e3, 4),
        }
        if opt_in is not None:
            result["opt_in_conv_emu"] = opt_in
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(hypes, model, frame_cpu, anchors, args.cpu_frames, args.cpu_threads, args.cpu_budget_s)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(hypes, model, frame_cpu, anchors, n_frames, threads, budget_s):
    """The CPU oracle (numpy/torch-CPU restatement of the reference path, oracle/) on the SAME frame.
    Bounded sample: up to n_frames timed frames, stopping once `budget_s` seconds are spent (>= 1 frame)."""
    from oracle import coalign_oracle as oracle
    cores = max(1, min(threads, os.cpu_count() or 1))
    torch.set_num_threads(cores)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    margs = hypes["model"]["args"]
    done, t0 = 0, time.perf_counter()
    while done < n_frames and (done == 0 or time.perf_counter() - t0 < budget_s):
        with torch.no_grad():
            out = oracle.coalign_forward(sd, margs, frame_cpu)
            oracle.post_process([out], anchors, hypes["postprocess"])
        done += 1
    dt = time.perf_counter() - t0
    return {"value": round(done / dt, 4), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{done} frame(s) of the same 5-agent synthetic workload (no warm-up, {dt:.1f} s), torch CPU threads = {cores} "
                      f"of {os.cpu_count()} host cores"}


if __name__ == "__main__":
    main()
