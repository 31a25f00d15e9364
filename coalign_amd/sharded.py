"""Agent-sharded execution of the hot path across the GPUs of one node.

The reference has no multi-GPU inference at all (every agent of a frame is a row of one batch on one device; its only
distributed code is the DDP training wrapper, opencood/tools/train_ddp.py:104-109, multi_gpu_utils.py:32).  This is the
MI355X design: the per-agent stages (pillar encode -> canvas -> backbone) run where the agent's data lives, the three
multiscale feature maps (15.77 MB fp32 per agent at OPV2V size) cross xGMI once, and the ego runs warp + fusion + heads +
post-processing.  Softmax and weighted sum run at the ego in fixed agent order, so results are independent of the sharding.

Two schedules:

``FrameRing`` (throughput, weak scaling): with R ranks a step processes R frames of N agents.  Rank r is the EGO of frame r
and ENCODES agent a of frame (r - a) mod R, a = 0..N-1, so every rank does N encodes + 1 ego tail per step whatever R is.
One all-to-all per scale and step (RCCL over xGMI; point-to-point links, each rank talks to at most N-1 distinct peers,
15.77 MB per agent and link).  No pack / unpack copies: the local agents are stacked in DESTINATION order, so the backbone's
output tensors are the send buffers as they are, and the receive buffers (source-rank major) go to the fusion kernel together
with a row table ``rows[i]`` = "physical row holding logical agent i" (``coalign_warp_fuse_rows``).

``AgentGather`` (latency, BASELINE configs[2] as north_star words it): ONE frame, rank r owns a contiguous block of its
agents, one all-gather per scale, rank 0 (the ego) fuses.  With 5 ranks and 5 agents: one agent per GPU.

``wire_dtype`` (float16 / bfloat16): the feature maps are cast for the exchange and back (SURVEY §8f next-4 "compression on
the wire"); fp32 (default) is exact.  Relative error per element <= 2^-11 (fp16, inside its normal range) / 2^-8 (bf16).
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


# ------------------------------------------------------------------------------------------------ frame ring: plans
def send_plan(rank: int, world: int, n_agents: int):
    """-> (send_order, send_counts): the agents this rank encodes, grouped by destination rank (ascending), and the
    per-destination counts.  ``send_order`` IS the local slot order."""
    order, counts = [], [0] * world
    for d in range(world):
        for a in range(n_agents):
            if (rank - a) % world == d:
                order.append(a)
                counts[d] += 1
    return order, counts


def recv_plan(rank: int, world: int, n_agents: int):
    """-> (agent_of_recv_slot, recv_counts): which agent of MY frame each received row holds (source-rank major)."""
    agents, counts = [], [0] * world
    for s in range(world):
        for a in range(n_agents):
            if (s - a) % world == rank:          # source s encoded agent a of frame (s - a) % world == rank
                agents.append(a)
                counts[s] += 1
    return agents, counts


def encode_assignments(rank: int, world: int, n_agents: int) -> List[Tuple[int, int]]:
    """(frame, agent) pairs rank ``rank`` encodes in one step, in LOCAL SLOT order (= destination-rank major)."""
    order, _ = send_plan(rank, world, n_agents)
    return [((rank - a) % world, a) for a in order]


def _is_gloo_cuda(t: torch.Tensor, group) -> bool:
    return t.is_cuda and dist.get_backend(group) == "gloo"


def _rows_view(t: torch.Tensor) -> torch.Tensor:
    """[n, ...] -> [n, numel / n] view of the tensor's MEMORY (channels-last maps are flattened in their own (H, W, C) order)."""
    n = t.shape[0]
    if t.dim() == 4 and not t.is_contiguous() and t.is_contiguous(memory_format=torch.channels_last):
        return t.permute(0, 2, 3, 1).reshape(n, -1)
    assert t.is_contiguous(), "feature maps must be dense (NCHW or channels-last)"
    return t.reshape(n, -1)


class FrameRing:
    """``exchange(feats)``: per scale ``[n_agents, C, H, W]`` in local slot order -> ``(recv, rows)``: per scale
    ``[n_agents, C, H, W]`` holding MY frame's agents source-rank major, and ``rows[i]`` = the row of logical agent i
    (agent 0 = ego)."""

    def __init__(self, n_agents: int, group=None, wire_dtype: Optional[torch.dtype] = None, force_collective: bool = False):
        self.n = n_agents
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if self.world > 1 else 0
        # round 6: a world of ONE still runs the collective (a self all-to-all on the backend's own stream and buffers) -- how the RCCL path is exercised on a
        # 1-GPU box (tests/test_round6_gpu.py, bench.py --force-dist); without it a single rank hands its maps straight through
        self.force_collective = bool(force_collective) and dist.is_available() and dist.is_initialized()
        self.send_order, self.send_counts = send_plan(self.rank, self.world, self.n)
        self.recv_agents, self.recv_counts = recv_plan(self.rank, self.world, self.n)
        self.rows = [0] * self.n
        for slot, a in enumerate(self.recv_agents):
            self.rows[a] = slot
        self.wire_dtype = wire_dtype
        self._recv: Dict[tuple, torch.Tensor] = {}
        self.bytes_sent_last = 0

    def assignments(self) -> List[Tuple[int, int]]:
        return encode_assignments(self.rank, self.world, self.n)

    def exchange(self, feats: Sequence[torch.Tensor]) -> Tuple[List[torch.Tensor], List[int]]:
        if self.world == 1 and not self.force_collective:
            return list(feats), list(range(self.n))
        out, sent = [], 0
        for i, f in enumerate(feats):
            assert f.shape[0] == self.n
            send = f if self.wire_dtype is None else f.to(self.wire_dtype)
            key = (i, tuple(send.shape), tuple(send.stride()), send.dtype, str(send.device))
            recv = self._recv.get(key)
            if recv is None:
                recv = self._recv[key] = torch.empty_like(send)          # same memory format as the map (NCHW or channels-last)
            send2, recv2 = _rows_view(send), _rows_view(recv)
            if _is_gloo_cuda(send2, self.group):
                # functional-test route only (ranks sharing one GPU cannot use RCCL): staged through host memory
                host = torch.empty(recv2.shape, dtype=recv2.dtype)
                dist.all_to_all_single(host, send2.cpu(), output_split_sizes=self.recv_counts, input_split_sizes=self.send_counts,
                                       group=self.group)
                recv2.copy_(host)
            else:
                dist.all_to_all_single(recv2, send2, output_split_sizes=self.recv_counts, input_split_sizes=self.send_counts,
                                       group=self.group)
            sent += sum(c for d, c in enumerate(self.send_counts) if d != self.rank) * send2.shape[1] * send2.element_size()
            out.append(recv if self.wire_dtype is None else recv.to(f.dtype))
        self.bytes_sent_last = sent
        return out, list(self.rows)

    def step(self, encode_fn: Callable[[], Sequence[torch.Tensor]], tail_fn: Callable[[List[torch.Tensor], List[int]], object]):
        return tail_fn(*self.exchange(encode_fn()))


# ------------------------------------------------------------------------------------------------ one frame over R ranks
def agent_blocks(world: int, n_agents: int) -> Tuple[int, List[range]]:
    """Contiguous agent blocks of ``per = ceil(N / R)`` agents: rank r owns agents [r * per, min(N, (r + 1) * per))."""
    per = (n_agents + world - 1) // world
    return per, [range(min(n_agents, r * per), min(n_agents, (r + 1) * per)) for r in range(world)]


class AgentGather:
    """Latency mode: the agents of ONE frame are split over the ranks, ``gather(feats)`` all-gathers the per-scale maps
    (``[per, C, H, W]`` on every rank, unused rows arbitrary) into ``[n_agents, C, H, W]`` in agent order on every rank."""

    def __init__(self, n_agents: int, group=None, wire_dtype: Optional[torch.dtype] = None, force_collective: bool = False):
        self.n = n_agents
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if self.world > 1 else 0
        self.force_collective = bool(force_collective) and dist.is_available() and dist.is_initialized()      # (see FrameRing)
        self.per, self.blocks = agent_blocks(self.world, self.n)
        self.wire_dtype = wire_dtype
        self._recv: Dict[tuple, torch.Tensor] = {}
        self.bytes_sent_last = 0

    def local_agents(self) -> range:
        return self.blocks[self.rank]

    def gather(self, feats: Sequence[torch.Tensor]) -> List[torch.Tensor]:
        if self.world == 1 and not self.force_collective:
            return [f[: self.n] for f in feats]
        out, sent = [], 0
        for i, f in enumerate(feats):
            assert f.shape[0] == self.per
            send = f if self.wire_dtype is None else f.to(self.wire_dtype)
            cl = send.dim() == 4 and not send.is_contiguous() and send.is_contiguous(memory_format=torch.channels_last)
            key = (i, tuple(send.shape), cl, send.dtype, str(send.device))
            recv = self._recv.get(key)
            if recv is None:
                recv = self._recv[key] = torch.empty((self.world * self.per,) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device,
                                                     memory_format=torch.channels_last if cl else torch.contiguous_format)
            send2, recv2 = _rows_view(send), _rows_view(recv)
            if _is_gloo_cuda(send, self.group):
                parts = [torch.empty(send2.shape, dtype=send.dtype) for _ in range(self.world)]
                dist.all_gather(parts, send2.cpu(), group=self.group)
                recv2.copy_(torch.cat(parts))
            else:
                dist.all_gather_into_tensor(recv2, send2, group=self.group)
            sent += (self.world - 1) * send.numel() * send.element_size()
            full = recv[: self.n]                                # blocks are contiguous and in rank order: already agent order
            out.append(full if self.wire_dtype is None else full.to(f.dtype))
        self.bytes_sent_last = sent
        return out


# ------------------------------------------------------------------------------------------------ routing real frames
def split_agents(frame: dict) -> List[Dict[str, torch.Tensor]]:
    """Collated frame (one frame, N agents) -> per-agent pillar sets (rows with ``voxel_coords[:, 0] == a``)."""
    pl = frame["processed_lidar"]
    n = int(frame["record_len"][0]) if torch.is_tensor(frame["record_len"]) else int(frame["record_len"][0])
    agent = pl["voxel_coords"][:, 0]
    out = []
    for a in range(n):
        sel = (agent == a).nonzero(as_tuple=True)[0]
        out.append({k: pl[k].index_select(0, sel) for k in ("voxel_features", "voxel_coords", "voxel_num_points")})
    return out


def stack_agents(agent_sets: Sequence[Optional[Dict[str, torch.Tensor]]]) -> Dict[str, torch.Tensor]:
    """Per-agent pillar sets -> one ``processed_lidar`` whose agent index is the position in ``agent_sets``
    (``None`` = an empty slot: no pillars, all-zero canvas)."""
    live = [(i, s) for i, s in enumerate(agent_sets) if s is not None]
    feats = torch.cat([s["voxel_features"] for _, s in live])
    npts = torch.cat([s["voxel_num_points"] for _, s in live])
    coords = []
    for i, s in live:
        c = s["voxel_coords"].clone()
        c[:, 0] = i
        coords.append(c)
    return {"voxel_features": feats, "voxel_coords": torch.cat(coords), "voxel_num_points": npts}


def ring_batch(frames_by_agent: Sequence[Sequence[Dict[str, torch.Tensor]]], pairwise: Sequence[torch.Tensor], rank: int, world: int,
               n_agents: int, step: int) -> dict:
    """The batch rank ``rank`` works on in ring step ``step``: global frame ``step * world + f`` is frame f of the step; this
    rank encodes agent a of frame (rank - a) mod world (slots in destination order) and is the ego of frame ``rank``, whose
    pose matrices it carries.  ``frames_by_agent[g]`` = ``split_agents`` of pool frame g (indexed modulo the pool size)."""
    pool = len(frames_by_agent)
    sets = [frames_by_agent[(step * world + f) % pool][a] for f, a in encode_assignments(rank, world, n_agents)]
    return {"processed_lidar": stack_agents(sets), "record_len": [n_agents], "pairwise_t_matrix": pairwise[(step * world + rank) % pool]}


# ------------------------------------------------------------------------------------------------ pre-flight (no communicator needed)
def preflight(world: int, n_agents: int, feature_shapes: Sequence[Tuple[int, int, int]], channels_last: bool = True,
              wire_dtype: Optional[torch.dtype] = None, lanes: int = 1, mode: str = "ring") -> dict:
    """Everything about the exchange that can be checked WITHOUT a multi-GPU node: builds every rank's plans and (shape-only, ``meta`` device)
    buffers and validates them against what ``all_to_all_single`` / ``all_gather_into_tensor`` require on the NCCL (= RCCL) backend:

    * ring: rank s sends rank d exactly what d expects from s (split sizes match pairwise), every rank sends and receives ``n_agents`` rows,
      every (frame, agent) of a step is encoded exactly once, the row table is a permutation, each rank talks to at most ``n_agents - 1`` peers;
    * gather: every rank contributes the same number of rows (``per``), ``world * per >= n_agents``, blocks are contiguous and in rank order;
    * both: the send / receive views handed to the collective are CONTIGUOUS VIEWS of the feature maps (no hidden copy: channels-last maps are
      flattened in their own (H, W, C) order), one row = one agent, row size and every split boundary a multiple of 16 bytes.

    Raises ``ValueError`` on the first violation; returns a report (bytes per rank and step, peers per rank, buffers per lane).
    ``bench.py --gpus N --dry-run`` prints it; tests/test_sharded_cpu.py runs it for every world size 1..8."""
    if mode not in ("ring", "gather"):
        raise ValueError("mode must be 'ring' or 'gather'")
    dt = torch.float32 if wire_dtype is None else wire_dtype
    esize = torch.empty((), dtype=dt).element_size()

    def fail(msg):
        raise ValueError(f"exchange pre-flight ({mode}, world {world}, {n_agents} agents): {msg}")

    def views_ok(n_rows, C, H, W):
        t = torch.empty((n_rows, C, H, W), dtype=dt, device="meta", memory_format=torch.channels_last if channels_last else torch.contiguous_format)
        v = _rows_view(t)
        if not v.is_contiguous() or tuple(v.shape) != (n_rows, C * H * W) or v.stride(0) != C * H * W:
            fail(f"the row view of a [{n_rows}, {C}, {H}, {W}] map is not a dense [rows, row_size] matrix")
        if (C * H * W * esize) % 16:
            fail(f"row size {C * H * W * esize} B is not a multiple of 16 B")
        return C * H * W * esize

    row_bytes = [views_ok(max(n_agents, 1), C, H, W) for C, H, W in feature_shapes]
    report = {"mode": mode, "world": world, "n_agents": n_agents, "lanes": lanes, "wire_dtype": str(dt), "row_bytes_per_scale": row_bytes,
              "communicators": 1, "collectives_per_step": len(feature_shapes)}
    if mode == "ring":
        sends = [send_plan(r, world, n_agents) for r in range(world)]
        recvs = [recv_plan(r, world, n_agents) for r in range(world)]
        seen = set()
        for r in range(world):
            order, sc = sends[r]
            agents, rc = recvs[r]
            if sum(sc) != n_agents or sum(rc) != n_agents or len(order) != n_agents:
                fail(f"rank {r} sends {sum(sc)} / receives {sum(rc)} rows, expected {n_agents}")
            for d in range(world):
                if sc[d] != recvs[d][1][r]:
                    fail(f"rank {r} sends {sc[d]} rows to rank {d}, which expects {recvs[d][1][r]}")
            if sorted(agents) != list(range(n_agents)):
                fail(f"rank {r}'s received rows do not cover agents 0..{n_agents - 1} once: {agents}")
            rows = [0] * n_agents
            for slot, a in enumerate(agents):
                rows[a] = slot
            if sorted(rows) != list(range(n_agents)):
                fail(f"rank {r}'s row table is not a permutation: {rows}")
            for fa in encode_assignments(r, world, n_agents):
                if fa in seen:
                    fail(f"(frame, agent) {fa} is encoded twice")
                seen.add(fa)
        if len(seen) != world * n_agents:
            fail(f"{len(seen)} (frame, agent) pairs encoded per step, expected {world * n_agents}")
        peers = [sum(1 for d, c in enumerate(sends[r][1]) if c and d != r) for r in range(world)]
        if world > 1 and max(peers) > max(1, min(world - 1, n_agents)):
            fail(f"a rank talks to {max(peers)} peers")
        report.update({"peers_per_rank": peers, "send_counts": [sc for _, sc in sends], "recv_counts": [rc for _, rc in recvs],
                       "bytes_sent_per_rank_per_step": [sum(c for d, c in enumerate(sends[r][1]) if d != r) * sum(row_bytes) for r in range(world)],
                       "buffers_per_lane": {"send": "the backbone's output maps themselves", "recv_bytes": n_agents * sum(row_bytes)}})
    else:
        per, blocks = agent_blocks(world, n_agents)
        if per * world < n_agents:
            fail(f"{world} blocks of {per} agents do not cover {n_agents}")
        flat = [a for b in blocks for a in b]
        if flat != list(range(n_agents)):
            fail(f"agent blocks are not contiguous / in rank order: {blocks}")
        for C, H, W in feature_shapes:
            views_ok(per, C, H, W)
            views_ok(world * per, C, H, W)
        report.update({"per": per, "blocks": [list(b) for b in blocks], "bytes_sent_per_rank_per_step": [(world - 1) * per * sum(row_bytes)] * world,
                       "buffers_per_lane": {"recv_bytes": world * per * sum(row_bytes)}})
    return report


# ---- first contact with a collective backend, hardened (the reference's own bring-up: opencood/tools/multi_gpu_utils.py:31-37) ----

SCHEDULES = ("ring", "gather", "replicas")          # fall-back order: frame ring -> one-frame agent gather -> N independent replicas (no collective)


def control_agree(ok: bool, group=None) -> bool:
    """True iff EVERY rank reports ok.  ``group``: a control group that does not depend on the data-plane backend (gloo next to RCCL): a rank whose
    collective raised must not leave the others waiting inside the next one."""
    t = torch.tensor([1 if ok else 0], dtype=torch.int32)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return bool(int(t.item()))


def negotiate_schedule(start: str, attempt, agree, report=None) -> Tuple[str, List[str]]:
    """Run ``attempt(mode)`` -- the mode's set-up and first exchanges on THIS rank, raising on any failure -- for ``start`` and, whenever ANY rank failed
    (``agree(ok)`` = logical AND over the ranks), for the next schedule of ``SCHEDULES``; ``"replicas"`` is terminal: if it fails too, every rank raises.
    Every rank takes the same decisions.
    Returns (schedule that runs, list of fall-backs taken)."""
    if start not in SCHEDULES:
        raise ValueError(f"schedule must be one of {SCHEDULES}")
    mode, fallbacks = start, []
    while True:
        ok, err = True, None
        try:
            attempt(mode)
        except Exception as e:      # noqa: BLE001 -- whatever the backend throws
            ok, err = False, e
            if report is not None:
                report(mode, e)
        if agree(ok):
            return mode, fallbacks
        if mode == "replicas":      # nothing left to fall back to: every rank raises (the one that failed with its own error)
            raise RuntimeError(f"the terminal schedule 'replicas' failed{'' if err is None else ' on this rank: ' + repr(err)}") from err
        nxt = SCHEDULES[SCHEDULES.index(mode) + 1]
        fallbacks.append(f"{mode} failed in its first exchanges -> {nxt}")
        mode = nxt

