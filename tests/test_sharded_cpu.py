"""World-size-2/3 gloo tests of the frame-ring exchange (CPU, no kernels): every rank must end up with exactly
the agents of its own frame, in agent order, bit for bit."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from coalign_amd.sharded import FrameRing, encode_assignments, recv_plan, send_plan


def _tagged(frame, agent, shape):
    base = float(frame * 100 + agent)
    return torch.arange(int(torch.tensor(shape).prod()), dtype=torch.float32).reshape(shape) * 1e-3 + base


def _worker(rank, world, n_agents, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ring = FrameRing(n_agents)
        shapes = [(4, 6, 8), (8, 3, 4), (16, 2, 2)]
        local = encode_assignments(rank, world, n_agents)
        feats = [torch.stack([_tagged(f, a, s) for f, a in local]) for s in shapes]
        got = ring.exchange(feats)
        ok = all(torch.equal(got[k][a], _tagged(rank, a, s)) for k, s in enumerate(shapes) for a in range(n_agents))
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_agents", [(2, 5), (3, 2), (2, 1)])
def test_frame_ring_exchange(world, n_agents):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + world * 7 + n_agents) % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, n_agents, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = dict(q.get(timeout=10) for _ in range(world))
    assert all(res[r] for r in range(world)), res


def test_plans_are_consistent():
    for world in (1, 2, 4, 5, 8):
        for n in (1, 2, 5, 8):
            total_sent = [[0] * world for _ in range(world)]
            for r in range(world):
                order, counts = send_plan(r, world, n)
                assert sorted(order) == list(range(n)) and sum(counts) == n
                for d in range(world):
                    total_sent[r][d] = counts[d]
                # every frame gets each agent exactly once
            for r in range(world):
                agents, counts = recv_plan(r, world, n)
                assert sorted(agents) == list(range(n))
                assert counts == [total_sent[s][r] for s in range(world)]
            frames = {}
            for r in range(world):
                for f, a in encode_assignments(r, world, n):
                    frames.setdefault(f, []).append(a)
            assert all(sorted(v) == list(range(n)) for v in frames.values()) and len(frames) == world
