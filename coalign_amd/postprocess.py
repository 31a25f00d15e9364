"""Detection post-processing of the hot path, host side.

``VoxelPostprocessor`` keeps the reference's constructor, ``generate_anchor_box()`` and
``post_process(data_dict, output_dict)`` contract (opencood/data_utils/post_processor/voxel_postprocessor.py:25-81,
243-402); ``nms_rotated`` keeps box_utils.nms_rotated's (opencood/utils/box_utils.py:693-738).  Decode, sanity
filters, rotated NMS and the range filter run in the gfx950 kernels (``coalign_anchor_decode``,
``coalign_nms_rotated``, ``coalign_gather_in_range``) back to back on one stream; the only device->host traffic is
the final box count.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from . import ops

NMS_TOP = 1000  # box_utils.py:719


def nms_rotated(boxes: torch.Tensor, scores: torch.Tensor, threshold: float) -> np.ndarray:
    """boxes [N, 8, 3] or [N, 4, 2] (device), scores [N] -> np.int32 indices of the kept boxes in pick order."""
    if boxes.shape[0] == 0:
        return np.array([], dtype=np.int32)
    keep, cnt = ops.nms_rotated_device(boxes, scores, threshold, NMS_TOP)
    return keep[: int(cnt.item())].cpu().numpy().astype(np.int32)


class VoxelPostprocessor:
    def __init__(self, anchor_params: dict, train: bool):
        self.params = anchor_params
        self.train = train
        self.bbx_dict = {}
        self.anchor_num = self.params["anchor_args"]["num"]
        self._buffers: Dict[tuple, list] = {}
        self._anchor_cache: Dict[tuple, torch.Tensor] = {}

    # ------------------------------------------------------------------------------------------ anchors (host)
    def generate_anchor_box(self) -> np.ndarray:
        """[H/stride, W/stride, num, 7] float64 anchors; centres are ``linspace(min + v, max - v, n)`` (not cell
        centres), z = -1, yaw from ``r`` in degrees (voxel_postprocessor.py:30-81)."""
        aa = self.params["anchor_args"]
        assert self.anchor_num == len(aa["r"])
        stride = aa.get("feature_stride", 2)
        rng = aa["cav_lidar_range"]
        xs = np.linspace(rng[0] + aa["vw"], rng[3] - aa["vw"], aa["W"] // stride)
        ys = np.linspace(rng[1] + aa["vh"], rng[4] - aa["vh"], aa["H"] // stride)
        out = np.empty((len(ys), len(xs), self.anchor_num, 7), dtype=np.float64)
        out[..., 0] = xs[None, :, None]
        out[..., 1] = ys[:, None, None]
        out[..., 2] = -1.0
        if self.params["order"] == "hwl":
            dims = (aa["h"], aa["w"], aa["l"])
        elif self.params["order"] == "lhw":
            dims = (aa["l"], aa["h"], aa["w"])
        else:
            raise ValueError("Unknown bbx order.")
        for k, v in enumerate(dims):
            out[..., 3 + k] = v
        out[..., 6] = np.array([math.radians(r) for r in aa["r"]])[None, None, :]
        return out

    # ------------------------------------------------------------------------------------------ decode + NMS (device)
    def _anchors_f32(self, anchor_box, device) -> torch.Tensor:
        a = anchor_box if torch.is_tensor(anchor_box) else torch.from_numpy(np.asarray(anchor_box))
        key = (a.data_ptr(), tuple(a.shape), str(a.dtype), str(device))
        hit = self._anchor_cache.get(key)
        if hit is None:
            hit = a.to(device=device).reshape(-1, 7).float().contiguous()   # anchors.view(-1, 7).float(), :431
            if len(self._anchor_cache) > 8:
                self._anchor_cache.clear()
            self._anchor_cache[key] = hit
        return hit

    def post_process(self, data_dict: dict, output_dict: dict) -> Tuple[Optional[torch.Tensor], Optional[torch.Tensor]]:
        """-> (pred_box3d [K', 8, 3], scores [K']) on the device, or (None, None) when nothing passes the score
        threshold.  ``data_dict`` holds one entry per cav (only 'ego' for early / intermediate fusion)."""
        return self.post_process_async(data_dict, output_dict, side_stream=False).result()

    def decode_buffers(self, output_dict: dict, n_cavs: int = 1) -> "ops.DecodeBuffers":
        """A caller-owned buffer set sized for these head outputs (``FramePipeline`` keeps one per lane for its HIP graphs)."""
        first = next(iter(output_dict.values()))
        cls0 = first["cls_preds"] if "cls_preds" in first else first["psm"]
        A, H, W = cls0.shape[-3:]
        return ops.DecodeBuffers(A * H * W * n_cavs, A, H, W, NMS_TOP, cls0.device)

    def enqueue(self, data_dict: dict, output_dict: dict, buf: "ops.DecodeBuffers", record_on: Optional[torch.cuda.Stream] = None) -> None:
        """Enqueue decode + NMS + range filter + the four result scalars' copy to pinned host memory on the CURRENT stream, into
        ``buf``.  Static launch geometry, no host synchronisation, no allocation once the anchors / transforms are on the device:
        the sequence can be captured into a HIP graph.  ``record_on``: the head outputs are marked as used by that stream."""
        cavs = list(data_dict.keys())
        if len(cavs) + 1 > buf.counts.numel():
            raise ValueError("too many cavs for one post_process call")
        thr = self.params["target_args"]["score_threshold"]
        da = self.params.get("dir_args", {})
        if not cavs:
            buf.reset_frame()                             # (nothing to decode: the fill launch still opens the frame; otherwise the first decode call clears the words)
        for slot, cav_id in enumerate(cavs):
            assert cav_id in output_dict
            out = output_dict[cav_id]
            cls = out["cls_preds"] if "cls_preds" in out else out["psm"]
            reg = out["reg_preds"] if "reg_preds" in out else out["rm"]
            dirp = out.get("dir_preds", out.get("dm"))
            if reg.dim() != 4:
                raise NotImplementedError("anchor-free heads are outside the CoAlign hot path")
            content = data_dict[cav_id]
            device = cls.device
            T = torch.as_tensor(content["transformation_matrix"]).to(device=device, dtype=torch.float32)
            anchors = self._anchors_f32(content["anchor_box"], device)
            if record_on is not None:                     # keep the head outputs alive until the side stream is done
                for t in (cls, reg, dirp):
                    if t is not None:
                        t.record_stream(record_on)
            ops.anchor_decode(buf, slot, cls, reg, dirp, anchors, thr, da.get("dir_offset", 0.0), da.get("num_bins", 2),
                              self.params["order"], T, clear_frame=slot == 0)
            if "iou_preds" in out:
                # IoU-head rescoring (voxel_postprocessor.py:335-339): scores *= ((clamp(sigmoid(iou), 0, 1) + 1) / 2) ** 4 on the candidates
                # this cav just appended (rows counts[slot] .. counts[slot + 1], flat anchor index in cand_index); element-wise torch ops on
                # the same values the reference forms, no host synchronisation (row bounds stay on the device)
                fac = torch.pow((torch.clamp(torch.sigmoid(out["iou_preds"].permute(0, 2, 3, 1).contiguous()).reshape(-1), min=0.0, max=1.0) + 1) * 0.5, 4)
                rows = torch.arange(buf.capacity, device=device, dtype=torch.int32)
                sel = (rows >= buf.counts[slot]) & (rows < buf.counts[slot + 1])
                raw = buf.cand_index.long()
                # rows outside this cav's range hold stale indices and only need SOME valid address; an index of a SELECTED row outside the head's anchors means
                # the candidate layout and the head disagree: that is reported through the status word (bit 1), not hidden by the clamp
                bad = sel & ((raw < 0) | (raw >= fac.numel()))
                buf.status.bitwise_or_(bad.any().to(torch.int32).reshape(1) * 2)
                idx = raw.clamp(0, fac.numel() - 1)
                buf.cand_score.copy_(torch.where(sel, buf.cand_score * fac[idx], buf.cand_score))
        total_dev = buf.counts[len(cavs): len(cavs) + 1]
        if NMS_TOP <= 1024:        # rank, bitmask, walk + in-range gather: three launches (coalign_nms_rotated_gather)
            ops.nms_rotated_gather(buf.cand_corners, buf.cand_score, self.params["nms_thresh"], NMS_TOP, buf.cand_keep, total_dev, buf.keep, buf.keep_count,
                                   self.params["gt_range"], buf.out_corners, buf.out_scores, buf.out_count, buf.nms_ws)
        else:
            ops.nms_rotated_device(buf.cand_corners, buf.cand_score, self.params["nms_thresh"], NMS_TOP, valid=buf.cand_keep,
                                   k_dev=total_dev, keep=buf.keep, keep_count=buf.keep_count, ws=buf.nms_ws)
            ops.gather_in_range(buf.cand_corners, buf.cand_score, buf.keep, buf.keep_count, self.params["gt_range"],
                                buf.out_corners, buf.out_scores, buf.out_count)
        # the frame's scalars (chained candidate totals, status, kept, final) are one 268-byte block: ONE async copy to pinned host memory behind the kernels
        buf.n_cavs = len(cavs)
        buf.host.copy_(buf.frame_words, non_blocking=True)

    def post_process_async(self, data_dict: dict, output_dict: dict, side_stream: bool = True) -> "PostProcessHandle":
        """Enqueue decode + NMS + range filter and return immediately.  With ``side_stream`` the kernels run on a
        second HIP stream (ordered after the head outputs), so the next frame's encoder overlaps these small,
        latency-bound launches; ``handle.result()`` waits for this frame only.  ``self.buffer_sets`` (default 2) buffer sets
        rotate, i.e. at most that many frames may be in flight (set it before the first call)."""
        cavs = list(data_dict.keys())
        first = output_dict[cavs[0]]
        cls0 = first["cls_preds"] if "cls_preds" in first else first["psm"]
        device = cls0.device
        A, H, W = cls0.shape[-3:]
        capacity = A * H * W * len(cavs)
        key = (str(device), A, H, W, capacity)
        ring = self._buffers.get(key)
        want = max(2, int(getattr(self, "buffer_sets", 2)))
        if ring is None or len(ring) < want:
            ring = self._buffers[key] = (ring or []) + [ops.DecodeBuffers(capacity, A, H, W, NMS_TOP, device) for _ in range(want - len(ring or []))]
        self._turn = getattr(self, "_turn", 0) + 1
        buf = ring[self._turn % len(ring)]
        main = torch.cuda.current_stream(device)
        stream = main
        if side_stream:
            if getattr(self, "_side", None) is None or self._side.device != device:
                self._side = torch.cuda.Stream(device=device)
            stream = self._side
            stream.wait_stream(main)
        with torch.cuda.device(device), torch.cuda.stream(stream):
            self.enqueue(data_dict, output_dict, buf, record_on=stream if side_stream else None)
            done = torch.cuda.Event()
            done.record(stream)
        return PostProcessHandle(self, buf, done)

    # ------------------------------------------------------------------------------------------ ground truth (host)
    def generate_gt_bbx(self, data_dict: dict) -> torch.Tensor:
        """Ground-truth corners [G, 8, 3] for evaluation (base_postprocessor.py:46-106): every agent's masked object
        centres become corners, are projected with ``transformation_matrix_clean``, de-duplicated by object id (first
        occurrence wins) and kept only when all eight corners -- z included -- fall inside ``gt_range``.  At most a few
        hundred boxes per frame, once per frame, outside the timed path: plain torch on whatever device the labels are on."""
        order = self.params["order"]
        boxes, ids = [], []
        for cav in data_dict.values():
            centre = cav["object_bbx_center"][cav["object_bbx_mask"] == 1].float()
            T = cav["transformation_matrix_clean"].to(centre.device, torch.float32)
            b = centre[:, [0, 1, 2, 5, 4, 3, 6]] if order == "hwl" else centre
            signs = centre.new_tensor([[1, -1, -1], [1, 1, -1], [-1, 1, -1], [-1, -1, -1],
                                       [1, -1, 1], [1, 1, 1], [-1, 1, 1], [-1, -1, 1]]) / 2
            local = b[:, None, 3:6] * signs[None]
            c, s = torch.cos(b[:, 6])[:, None], torch.sin(b[:, 6])[:, None]
            x = local[..., 0] * c - local[..., 1] * s + b[:, None, 0]
            y = local[..., 0] * s + local[..., 1] * c + b[:, None, 1]
            z = local[..., 2] + b[:, None, 2]
            homo = torch.stack((x, y, z, torch.ones_like(x)), dim=1)               # [n, 4, 8]
            boxes.append(torch.matmul(T, homo)[:, :3, :].transpose(1, 2))
            ids += list(cav["object_ids"])
        boxes = torch.vstack(boxes)
        boxes = boxes[[ids.index(i) for i in set(ids)]]
        lo = torch.tensor(self.params["gt_range"][0:3], dtype=torch.float64, device=boxes.device)
        hi = torch.tensor(self.params["gt_range"][3:6], dtype=torch.float64, device=boxes.device)
        b64 = boxes.double()
        return boxes[((b64 >= lo) & (b64 <= hi)).all(dim=2).all(dim=1)]

    @staticmethod
    def delta_to_boxes3d(deltas: torch.Tensor, anchors: torch.Tensor) -> torch.Tensor:
        """Dense decode of every anchor, [N, 7A, H, W] -> [N, H*W*A, 7] (voxel_postprocessor.py:405-450); plain
        tensor algebra on whatever device ``deltas`` lives on (the fused kernel decodes only passing anchors)."""
        N = deltas.shape[0]
        d = deltas.permute(0, 2, 3, 1).contiguous().view(N, -1, 7)
        a = anchors.to(d.device).view(-1, 7).float()
        diag = torch.sqrt(a[:, 4] ** 2 + a[:, 5] ** 2)
        xy = d[..., 0:2] * diag[None, :, None] + a[None, :, 0:2]
        z = d[..., 2:3] * a[None, :, 3:4] + a[None, :, 2:3]
        hwl = torch.exp(d[..., 3:6]) * a[None, :, 3:6]
        yaw = d[..., 6:7] + a[None, :, 6:7]
        return torch.cat([xy, z, hwl, yaw], dim=-1)


class PostProcessHandle:
    """Result of :meth:`VoxelPostprocessor.post_process_async`; ``result()`` performs the frame's only host sync."""

    def __init__(self, owner: VoxelPostprocessor, buf: ops.DecodeBuffers, done: torch.cuda.Event):
        self.owner, self.buf, self.done = owner, buf, done

    def result(self) -> Tuple[Optional[torch.Tensor], Optional[torch.Tensor]]:
        self.done.synchronize()
        words = self.buf.host.tolist()
        n_out, n_cand, n_keep, status = int(words[66]), int(words[self.buf.n_cavs]), int(words[65]), int(words[64])
        if status & 1:
            raise RuntimeError("post_process: candidate buffer overflow")
        if status & 2:
            raise RuntimeError("post_process: iou_preds rescoring met a candidate index outside the head's anchors (candidate layout and head disagree)")
        self.owner.last_counts = {"candidates": n_cand, "kept": n_keep, "final": n_out}
        if n_cand == 0:
            return None, None
        flat, top = self.buf.out_flat.clone(), self.buf.top          # (one copy launch for both arrays: the buffers are rewritten by the lane's next frame)
        return flat[: top * 24].view(top, 8, 3)[:n_out], flat[top * 24:][:n_out]


class UncertaintyVoxelPostprocessor(VoxelPostprocessor):
    """opencood/data_utils/post_processor/uncertainty_voxel_postprocessor.py:26-112: the post-processor of the stage-1 models."""

    def post_process_stage1(self, stage1_output_dict: dict, anchor_box):
        """Per-agent detections *in each agent's own frame* for box alignment: score threshold, box decode, direction fix,
        corners, rotated NMS per agent -- no projection, no size / z / range filters.  Returns three lists with one entry per
        agent: corners [K_i, 8, 3], boxes [K_i, 7], uncertainty [K_i, uncertainty_dim] (raw ``unc_preds`` of the kept anchors);
        ``(None, None, None)`` when no anchor of any agent passes the threshold.  Same kernels as ``post_process``: the decode
        kernel with an identity transform, the NMS without the sanity mask."""
        cls, reg, unc = stage1_output_dict["cls_preds"], stage1_output_dict["reg_preds"], stage1_output_dict["unc_preds"]
        dirp = stage1_output_dict.get("dir_preds")
        device = cls.device
        n_agents, A, H, W = cls.shape
        udim = unc.shape[1] // A
        anchors = self._anchors_f32(anchor_box, device)
        key = ("stage1", str(device), A, H, W)
        ring = self._buffers.get(key)
        if ring is None:
            ring = self._buffers[key] = [ops.DecodeBuffers(A * H * W, A, H, W, NMS_TOP, device)]
        buf = ring[0]
        thr = self.params["target_args"]["score_threshold"]
        da = self.params.get("dir_args", {})
        corners, boxes, uncertainty, any_box = [], [], [], False
        for i in range(n_agents):
            ops.anchor_decode(buf, 0, cls[i], reg[i], None if dirp is None else dirp[i], anchors, thr, da.get("dir_offset", 0.0),
                              da.get("num_bins", 2), self.params["order"], None, clear_frame=True)
            k_dev = buf.counts[1:2]
            ops.nms_rotated_device(buf.cand_corners, buf.cand_score, self.params["nms_thresh"], NMS_TOP, valid=None, k_dev=k_dev,
                                   keep=buf.keep, keep_count=buf.keep_count, ws=buf.nms_ws)
            n_keep, n_cand = int(buf.keep_count.item()), int(k_dev.item())
            any_box = any_box or n_cand > 0
            keep = buf.keep[:n_keep].long()
            flat = buf.cand_index[keep].long()                        # (h, w, anchor) order, like the reference's mask
            hw, an = torch.div(flat, A, rounding_mode="floor"), flat % A
            u = unc[i].reshape(A, udim, H * W)[an, :, hw]              # unc_preds.permute(0, 2, 3, 1).view(-1, udim)[mask]
            corners.append(buf.cand_corners[keep].clone())
            boxes.append(buf.cand_box7[keep].clone())
            uncertainty.append(u.reshape(-1, udim))
        if not any_box:
            return None, None, None
        return corners, boxes, uncertainty


POSTPROCESSORS = {"VoxelPostprocessor": VoxelPostprocessor, "UncertaintyVoxelPostprocessor": UncertaintyVoxelPostprocessor}


def build_postprocessor(anchor_cfg: dict, train: bool) -> VoxelPostprocessor:
    """opencood/data_utils/post_processor/__init__.py:build_postprocessor for the families on the hot path and its stage 1."""
    name = anchor_cfg["core_method"]
    if name not in POSTPROCESSORS:
        raise KeyError(f"post-processor '{name}' is outside the CoAlign hot path (available: {sorted(POSTPROCESSORS)})")
    return POSTPROCESSORS[name](anchor_params=anchor_cfg, train=train)
