"""Agent-sharded execution of the hot path across the GPUs of one node.

The reference has no multi-GPU inference at all (every agent of a frame is a row of one batch on one device);
this is the MI355X design: the per-agent stages (pillar encode -> canvas -> backbone) run where the agent's data
lives, the three multiscale feature maps (15.77 MB fp32 per agent at OPV2V size) cross xGMI once, and the ego runs
warp + fusion + heads + post-processing.

Schedule ("frame ring", weak scaling): with R ranks a step processes R frames of N agents.  Rank r is the EGO of
frame r and ENCODES agent a of frame (r - a) mod R, a = 0..N-1, so every rank does N encodes + 1 ego tail per step
whatever R is.  The exchange is one all-to-all per step (RCCL over xGMI; point-to-point links, each rank talks to
at most N-1 distinct peers, one 15.77 MB message per link): agent a encoded on rank r goes to rank (r - a) mod R;
frame r's agent a arrives from rank (r + a) mod R.  For R == 1 nothing is exchanged.
"""
from __future__ import annotations

from typing import Callable, List, Sequence, Tuple

import torch
import torch.distributed as dist


def encode_assignments(rank: int, world: int, n_agents: int) -> List[Tuple[int, int]]:
    """(frame, agent) pairs rank ``rank`` encodes in one step, in local slot order a = 0..n_agents-1."""
    return [((rank - a) % world, a) for a in range(n_agents)]


def send_plan(rank: int, world: int, n_agents: int):
    """-> (send_order, send_counts): local slots grouped by destination rank (ascending), and per-destination counts."""
    order, counts = [], [0] * world
    for d in range(world):
        for a in range(n_agents):
            if (rank - a) % world == d:
                order.append(a)
                counts[d] += 1
    return order, counts


def recv_plan(rank: int, world: int, n_agents: int):
    """-> (agent_of_recv_slot, recv_counts): which agent of MY frame each received slot holds (source-rank major)."""
    agents, counts = [], [0] * world
    for s in range(world):
        for a in range(n_agents):
            if (s - a) % world == rank:          # source s encoded agent a of frame (s - a) % world == rank
                agents.append(a)
                counts[s] += 1
    return agents, counts


class FrameRing:
    """Runs ``encode_fn`` on the local agents, exchanges the packed multiscale features, returns this rank's
    frame as per-scale tensors ``[n_agents, C_s, H_s, W_s]`` in agent order (agent 0 = ego)."""

    def __init__(self, n_agents: int, group=None):
        self.n = n_agents
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if self.world > 1 else 0
        self.send_order, self.send_counts = send_plan(self.rank, self.world, self.n)
        self.recv_agents, self.recv_counts = recv_plan(self.rank, self.world, self.n)
        self._send = self._recv = None

    def exchange(self, feats: Sequence[torch.Tensor]) -> List[torch.Tensor]:
        """feats: per scale [n_agents, C, H, W] in local slot order -> same shapes holding MY frame's agents."""
        if self.world == 1:
            return list(feats)
        shapes = [tuple(f.shape[1:]) for f in feats]
        sizes = [f[0].numel() for f in feats]
        per_agent = sum(sizes)
        dev, dt = feats[0].device, feats[0].dtype
        if self._send is None or self._send.shape != (self.n, per_agent) or self._send.device != dev:
            self._send = torch.empty((self.n, per_agent), dtype=dt, device=dev)
            self._recv = torch.empty((self.n, per_agent), dtype=dt, device=dev)
        idx = torch.as_tensor(self.send_order, device=dev)
        off = 0
        for f, sz in zip(feats, sizes):          # pack: destination-major rows, scales side by side
            self._send[:, off:off + sz] = f.reshape(self.n, sz).index_select(0, idx)
            off += sz
        if self._send.is_cuda and dist.get_backend(self.group) == "gloo":
            # functional-test route only (two ranks sharing one GPU cannot use RCCL): stage through host memory
            recv = torch.empty(self._recv.shape, dtype=dt)
            dist.all_to_all_single(recv, self._send.cpu(), output_split_sizes=self.recv_counts,
                                   input_split_sizes=self.send_counts, group=self.group)
            self._recv.copy_(recv)
        else:
            dist.all_to_all_single(self._recv, self._send, output_split_sizes=self.recv_counts,
                                   input_split_sizes=self.send_counts, group=self.group)
        inv = torch.empty(self.n, dtype=torch.long)
        for slot, a in enumerate(self.recv_agents):
            inv[a] = slot
        inv = inv.to(dev)
        out, off = [], 0
        for shp, sz in zip(shapes, sizes):
            out.append(self._recv[:, off:off + sz].index_select(0, inv).reshape((self.n,) + shp))
            off += sz
        return out

    def step(self, encode_fn: Callable[[], Sequence[torch.Tensor]], tail_fn: Callable[[List[torch.Tensor]], object]):
        return tail_fn(self.exchange(encode_fn()))
