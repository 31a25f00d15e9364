"""SURVEY §8(e) "Determinism", with the real kernels: two ranks sharing the one GPU (gloo, exchange staged through host memory --
RCCL refuses two ranks on one device) run ROUTED frames out of a common pool through ``FrameRing`` / ``AgentGather`` and must
reproduce the 1-rank run: gathered feature maps ``torch.equal``, head outputs equal, identical detections.  Also through the
product's ``FramePipeline`` with one ring + one communicator per lane (what ``bench.py --gpus N`` runs)."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

N_AGENTS, POOL, PILLARS = 3, 4, 3000


def _setup():
    from coalign_amd.config import builtin_config
    from coalign_amd.detector import build_model, to_device
    from coalign_amd.postprocess import build_postprocessor
    from coalign_amd.synthetic import calibrate_heads_, fill_parameters_, make_frame
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    h = builtin_config("opv2v_coalign")
    frames = []
    for g in range(POOL):
        f = to_device(make_frame(h, N_AGENTS, pillars_per_agent=PILLARS, seed=500 + g, noise=(0.2, 0.2)), dev)
        f["record_len"] = [N_AGENTS]
        frames.append(f)
    model = build_model(h)
    fill_parameters_(model, seed=0)
    model = model.to(dev).eval()
    pp = build_postprocessor(h["postprocess"], False)
    calibrate_heads_(model, frames[0], 0.2, 400)            # deterministic: same frame, same weights on every rank
    anchors = torch.from_numpy(pp.generate_anchor_box())
    meta = {"ego": {"transformation_matrix": torch.eye(4, device=dev), "anchor_box": anchors}}
    return h, frames, model, pp, anchors, meta


def _same(a, b):
    return (a is None and b is None) or (a is not None and b is not None and torch.equal(a, b))


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    fails = []
    try:
        from coalign_amd.pipeline import FramePipeline
        from coalign_amd.sharded import AgentGather, FrameRing, ring_batch, split_agents, stack_agents
        h, frames, model, pp, anchors, meta = _setup()
        by_agent = [split_agents(f) for f in frames]
        pair = [f["pairwise_t_matrix"] for f in frames]
        with torch.no_grad():
            # ---- the 1-rank reference of every pool frame, computed locally
            single = []
            for f in frames:
                feats, affine = model.encode(f)
                out = model.fuse_and_head(list(feats), [N_AGENTS], affine)
                single.append((feats, out, pp.post_process(meta, {"ego": out})))
            # ---- frame ring, step by step
            ring = FrameRing(N_AGENTS)
            for step in range(POOL // world):
                g = (step * world + rank) % POOL
                batch = ring_batch(by_agent, pair, rank, world, N_AGENTS, step)
                feats, affine = model.encode(batch)
                recv, rows = ring.exchange(feats)
                for k in range(3):
                    for a in range(N_AGENTS):
                        if not torch.equal(recv[k][rows[a]], single[g][0][k][a]):
                            fails.append(f"ring step {step} scale {k} agent {a}: features differ from the 1-rank run")
                out = model.fuse_and_head(recv, [N_AGENTS], affine, rows)
                for key in out:
                    if not torch.equal(out[key], single[g][1][key]):
                        fails.append(f"ring step {step}: {key} differs")
                b, s = pp.post_process(meta, {"ego": out})
                if not (_same(b, single[g][2][0]) and _same(s, single[g][2][1])):
                    fails.append(f"ring step {step}: detections differ")
            # ---- the same through the product pipeline: 2 lanes, one ring + one communicator per lane
            groups = [dist.new_group(backend="gloo") for _ in range(2)]
            rings = [FrameRing(N_AGENTS, group=gr) for gr in groups]
            pipe = FramePipeline(model, pp, anchors, lanes=2, result_lag=1, exchange=[r.exchange for r in rings])
            steps = [s % (POOL // world) for s in range(2 * (POOL // world))]
            res = pipe.run(ring_batch(by_agent, pair, rank, world, N_AGENTS, s) for s in steps)
            for s, (b, sc) in zip(steps, res):
                g = (s * world + rank) % POOL
                if not (_same(b, single[g][2][0]) and _same(sc, single[g][2][1])):
                    fails.append(f"pipeline ring step {s}: detections differ")
            # ---- latency mode (BASELINE configs[2] as worded): ONE frame, agents split over the ranks, all-gather, ego fuses
            ag = AgentGather(N_AGENTS)
            mine = list(ag.local_agents())
            sets = [by_agent[1][a] for a in mine] + [None] * (ag.per - len(mine))
            if mine:
                batch = {"processed_lidar": stack_agents(sets), "record_len": [ag.per], "pairwise_t_matrix": pair[1]}
                feats, affine = model.encode(batch)
            else:       # a rank without agents still takes part in the collective
                feats = [torch.zeros_like(t[: ag.per]) for t in single[1][0]]
            full = ag.gather(list(feats))
            for k in range(3):
                if not torch.equal(full[k], single[1][0][k]):
                    err = float((full[k] - single[1][0][k]).abs().max())
                    fails.append(f"agent-gather scale {k}: features differ from the 1-rank run (max {err:.3e})")
            if rank == 0:
                from coalign_amd.pose import normalize_pairwise_tfm
                affine = normalize_pairwise_tfm(pair[1], 200, 704, 0.4)
                out = model.fuse_and_head(full, [N_AGENTS], affine)
                b, s = pp.post_process(meta, {"ego": out})
                if not (_same(b, single[1][2][0]) and _same(s, single[1][2][1])):
                    fails.append("agent-gather: detections differ")
            n_det = sum(0 if x[2][0] is None else x[2][0].shape[0] for x in single)
            if n_det < 100:
                fails.append(f"only {n_det} detections in the pool: the test does not exercise NMS")
        torch.cuda.synchronize()
        q.put((rank, fails))
    except Exception as e:      # noqa: BLE001
        import traceback
        q.put((rank, [f"{type(e).__name__}: {e}\n{traceback.format_exc()}"]))
    finally:
        dist.destroy_process_group()


def test_two_ranks_on_one_gpu_reproduce_the_single_rank_run():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() * 17) % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(60)
    assert all(not res[r] for r in range(world)), res
