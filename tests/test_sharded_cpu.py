"""World-size-2/3 gloo tests of the agent-sharded schedules (CPU tensors, no kernels): after the exchange every rank must
hold exactly the agents of its own frame, addressable in agent order through the row table, bit for bit; the routing of real
frames out of a shared pool reaches every (frame, agent) pair exactly once.  (The same schedules with the real kernels on a GPU:
tests/test_sharded_gpu.py.)"""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from coalign_amd.sharded import (AgentGather, FrameRing, agent_blocks, control_agree, encode_assignments, negotiate_schedule, recv_plan, ring_batch,
                                 send_plan, split_agents, stack_agents)

SHAPES = [(4, 6, 8), (8, 3, 4), (16, 2, 2)]


def _tagged(frame, agent, shape):
    base = float(frame * 100 + agent)
    return torch.arange(int(torch.tensor(shape).prod()), dtype=torch.float32).reshape(shape) * 1e-3 + base


def _ring_worker(rank, world, n_agents, port, q, wire):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ring = FrameRing(n_agents, wire_dtype=wire)
        local = ring.assignments()
        assert local == encode_assignments(rank, world, n_agents)
        feats = [torch.stack([_tagged(f, a, s) for f, a in local]) for s in SHAPES]
        got, rows = ring.exchange(feats)
        ok = sorted(rows) == list(range(n_agents))
        for k, s in enumerate(SHAPES):
            for a in range(n_agents):
                want = _tagged(rank, a, s)
                if wire is not None:
                    want = want.to(wire).float()
                ok = ok and torch.equal(got[k][rows[a]], want)
        per_agent = sum(int(torch.tensor(s).prod()) for s in SHAPES) * (4 if wire is None else 2)
        remote = sum(1 for f, a in local if f != rank)
        ok = ok and ring.bytes_sent_last == remote * per_agent
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def _gather_worker(rank, world, n_agents, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ag = AgentGather(n_agents)
        mine = list(ag.local_agents())
        feats = []
        for s in SHAPES:
            rows = [_tagged(7, a, s) for a in mine] + [torch.full(s, -1.0)] * (ag.per - len(mine))      # unused slots: junk
            feats.append(torch.stack(rows))
        got = ag.gather(feats)
        ok = all(g.shape[0] == n_agents for g in got)
        ok = ok and all(torch.equal(got[k][a], _tagged(7, a, s)) for k, s in enumerate(SHAPES) for a in range(n_agents))
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def _spawn(target, world, args):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() * 13 + world * 7 + hash(str(args)) % 997) % 2000
    procs = [ctx.Process(target=target, args=(r, world) + tuple(args[:1]) + (port, q) + tuple(args[1:])) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = dict(q.get(timeout=10) for _ in range(world))
    assert all(res[r] for r in range(world)), res


@pytest.mark.parametrize("world,n_agents,wire", [(2, 5, None), (3, 2, None), (2, 1, None), (3, 5, torch.float16), (2, 5, torch.bfloat16)])
def test_frame_ring_exchange(world, n_agents, wire):
    _spawn(_ring_worker, world, (n_agents, wire))


@pytest.mark.parametrize("world,n_agents", [(2, 5), (3, 5), (2, 2), (3, 2)])
def test_agent_gather(world, n_agents):
    _spawn(_gather_worker, world, (n_agents,))


def test_plans_are_consistent():
    for world in (1, 2, 4, 5, 8):
        for n in (1, 2, 5, 8):
            total_sent = [[0] * world for _ in range(world)]
            for r in range(world):
                order, counts = send_plan(r, world, n)
                assert sorted(order) == list(range(n)) and sum(counts) == n
                dests = [(r - a) % world for a in order]
                assert dests == sorted(dests)                      # local slot order IS destination order: no pack copy
                for d in range(world):
                    total_sent[r][d] = counts[d]
            for r in range(world):
                agents, counts = recv_plan(r, world, n)
                assert sorted(agents) == list(range(n))
                assert counts == [total_sent[s][r] for s in range(world)]
            frames = {}
            for r in range(world):
                for f, a in encode_assignments(r, world, n):
                    frames.setdefault(f, []).append(a)
            assert all(sorted(v) == list(range(n)) for v in frames.values()) and len(frames) == world
            per, blocks = agent_blocks(world, n)
            assert [a for b in blocks for a in b] == list(range(n)) and all(len(b) <= per for b in blocks)


def _toy_frame(g, n_agents, pillars):
    gen = torch.Generator().manual_seed(g)
    coords = torch.cat([torch.stack([torch.full((pillars,), a), torch.zeros(pillars, dtype=torch.long), torch.arange(pillars), torch.arange(pillars) + g], 1)
                        for a in range(n_agents)]).int()
    m = coords.shape[0]
    return {"processed_lidar": {"voxel_features": torch.randn(m, 4, 4, generator=gen), "voxel_coords": coords,
                                "voxel_num_points": torch.randint(1, 5, (m,), generator=gen).int()},
            "record_len": torch.tensor([n_agents]), "pairwise_t_matrix": torch.full((1, 5, 5, 4, 4), float(g), dtype=torch.float64)}


def test_ring_batches_route_every_agent_of_every_pool_frame_once():
    """Real-frame routing: over one period of steps the ranks together encode agent a of pool frame g exactly once, each rank
    carries the pose matrices of the frame it is the ego of, and the local agent index is the slot."""
    n, pool = 5, 8
    frames = [_toy_frame(g, n, 3 + g) for g in range(pool)]
    by_agent = [split_agents(f) for f in frames]
    pair = [f["pairwise_t_matrix"] for f in frames]
    for a in range(n):                                             # split + stack round trip
        assert torch.equal(by_agent[2][a]["voxel_coords"][:, 0], torch.full((5,), a, dtype=torch.int32))
    back = stack_agents(by_agent[2])
    assert all(torch.equal(back[k], frames[2]["processed_lidar"][k]) for k in back)
    for world in (1, 2, 4, 8):
        seen = {}
        for step in range(pool // world):
            for r in range(world):
                b = ring_batch(by_agent, pair, r, world, n, step)
                assert float(b["pairwise_t_matrix"][0, 0, 0, 0, 0]) == (step * world + r) % pool       # my frame's poses
                pl = b["processed_lidar"]
                for slot, (f, a) in enumerate(encode_assignments(r, world, n)):
                    g = (step * world + f) % pool
                    sel = pl["voxel_coords"][:, 0] == slot
                    assert torch.equal(pl["voxel_features"][sel], by_agent[g][a]["voxel_features"])
                    assert torch.equal(pl["voxel_coords"][sel][:, 1:], by_agent[g][a]["voxel_coords"][:, 1:])
                    seen[(g, a)] = seen.get((g, a), 0) + 1
        assert seen == {(g, a): 1 for g in range(pool) for a in range(n)}


def test_exchange_preflight_every_world_size():
    """coalign_amd.sharded.preflight (what `bench.py --gpus N --dry-run` prints): plans of every rank consistent pairwise, every (frame, agent)
    encoded once, contiguous row views for NCHW and channels-last maps, 16-byte rows -- for world sizes 1..8 and 5 / 8 agents, both schedules;
    and it does fail on a broken configuration."""
    import pytest
    import torch
    from coalign_amd.sharded import preflight
    opv2v = [(64, 100, 352), (128, 50, 176), (256, 25, 88)]
    lss = [(64, 120, 120), (128, 60, 60), (256, 30, 30)]
    for world in range(1, 9):
        for n, shapes in ((5, opv2v), (8, lss), (2, opv2v)):
            for cl in (True, False):
                for mode in ("ring", "gather"):
                    rep = preflight(world, n, shapes, cl, None, 3, mode)
                    assert rep["communicators"] == 1 and rep["collectives_per_step"] == 3
                    if mode == "ring":
                        assert max(rep["peers_per_rank"]) <= min(world - 1, n) and all(sum(c) == n for c in rep["send_counts"])
                        if world > 1:      # all but the rows a rank keeps for itself cross a link, once
                            keep = [rep["send_counts"][r][r] for r in range(world)]
                            assert rep["bytes_sent_per_rank_per_step"] == [(n - k) * sum(rep["row_bytes_per_scale"]) for k in keep]
                    else:
                        assert rep["per"] * world >= n and sum(len(b) for b in rep["blocks"]) == n
    assert preflight(5, 5, opv2v, wire_dtype=torch.float16)["row_bytes_per_scale"][0] == 64 * 100 * 352 * 2
    with pytest.raises(ValueError):
        preflight(2, 5, [(3, 5, 7)])                    # 105 floats per row: not a multiple of 16 bytes
    with pytest.raises(ValueError):
        preflight(2, 5, opv2v, mode="scatter")


def _negotiate_worker(rank, world, failing, port, q):
    """``failing``: {schedule: ranks whose first exchanges raise}.  Every rank must end on the same schedule with the same fall-back list, and a rank
    that did NOT fail must not be left inside a collective the failing rank never entered (the all-reduce of the control group is the only rendezvous)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ctl = dist.new_group(backend="gloo")
        tried, reported = [], []

        def attempt(mode):
            tried.append(mode)
            if rank in failing.get(mode, ()):
                raise RuntimeError(f"injected failure of '{mode}' on rank {rank}")
            # (a real schedule enters its data-plane collective here; this worker only records the attempt: the control group's all-reduce is the rendezvous)

        if failing.get("ring") and failing.get("gather") and failing.get("replicas"):      # nothing runs: EVERY rank must raise, none may return "replicas"
            try:
                negotiate_schedule("ring", attempt, lambda ok: control_agree(ok, ctl), lambda m, e: reported.append(m))
                q.put((rank, False))
            except RuntimeError:
                q.put((rank, tried == ["ring", "gather", "replicas"]))
            return
        mode, fallbacks = negotiate_schedule("ring", attempt, lambda ok: control_agree(ok, ctl), lambda m, e: reported.append(m))
        out = [None] * world
        dist.all_gather_object(out, (mode, fallbacks, tried))
        want_mode = "ring" if not failing.get("ring") else "gather" if not failing.get("gather") else "replicas"
        ok = all(o == out[0] for o in out) and mode == want_mode and len(fallbacks) == ("ring", "gather", "replicas").index(want_mode)
        ok = ok and reported == [m for m in tried if rank in failing.get(m, ())]
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("failing", [{}, {"ring": (1,)}, {"ring": (0, 1), "gather": (0,)}, {"ring": (1,), "gather": (1,), "replicas": (1,)}])
def test_schedule_fallback_chain_is_taken_by_all_ranks_together(failing):
    """bench.py's RCCL bring-up: ring -> gather -> replicas, decided over a control group; world size 2, gloo."""
    _spawn(_negotiate_worker, 2, (failing,))


def test_negotiate_schedule_single_process_contract():
    calls = []
    def attempt(m):
        calls.append(m)
        if m == "ring":
            raise ValueError("no")
    assert negotiate_schedule("ring", attempt, lambda ok: ok) == ("gather", ["ring failed in its first exchanges -> gather"]) and calls == ["ring", "gather"]
    with pytest.raises(RuntimeError):      # terminal: nothing left to fall back to -- a failed 'replicas' attempt is an error, not a schedule that "ran"
        negotiate_schedule("replicas", lambda m: 1 / 0, lambda ok: ok)
    assert negotiate_schedule("replicas", lambda m: None, lambda ok: ok) == ("replicas", [])
    with pytest.raises(ValueError):
        negotiate_schedule("mesh", attempt, lambda ok: ok)



def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` WITHOUT a launcher starts two ranks by itself (torch.distributed.run on 127.0.0.1) and rank 0
    reports world 2; with more GPUs requested than visible it exits non-zero instead of printing a one-GPU line (no GPU in this container: --gpus 8)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-check"], capture_output=True, text=True, timeout=300, env=env)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, r.stderr[-800:]
    rep = json.loads(lines[0])
    assert rep == {"launch_check": True, "world": 2, "n_gpus": 2, "launcher": "self"}
    if not __import__("torch").cuda.is_available():
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8"], capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 2 and "not launching" in r.stderr and not any(l.startswith("{") for l in r.stdout.splitlines())
