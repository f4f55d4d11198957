"""Tile-sharded multi-GPU rendering: one process (rank) per GPU, scene replicated, the reference's own 20x20
block grid (raytracer/render.rs:103-110) dealt round-robin to the ranks, ONE gather of the RGB8 strips.

The reference has no multi-device path; its unit of parallel work is the block (render.rs:109-133, rayon
``par_bridge`` over blocks).  Blocks are independent, so there is no data-path collective besides the final
gather of finished pixels (RCCL over xGMI when the process group is ``nccl``).

Host-side helpers (``tiling``, ``owned_blocks``, ``strip_from_frame``, ``assemble_host``) restate the block/strip
layout of ``trr_render_shard_device`` / ``trr_assemble_device`` (include/trayracer_hip.h) in numpy for callers that
hold strips in host memory and for the world_size-2 gloo tests.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple

import numpy as np


def tiling(width: int, height: int) -> Tuple[int, int, int, int]:
    """(block_w, block_h, blocks_x, blocks_y) of render.rs:103-110; ZeroDivisionError below 20 px like render.rs:109."""
    bw, bh = width // 20, height // 20
    if bw == 0 or bh == 0:
        raise ZeroDivisionError("resolution below 20 pixels: the reference divides by zero (render.rs:109)")
    return bw, bh, width // bw, height // bh


def owned_blocks(width: int, height: int, shard: int, n_shards: int) -> List[int]:
    """Row-major block ids rendered by ``shard``: b % n_shards == shard."""
    _, _, nbx, nby = tiling(width, height)
    return list(range(shard, nbx * nby, n_shards))


def strip_bytes(width: int, height: int, n_shards: int) -> int:
    bw, bh, nbx, nby = tiling(width, height)
    return ((nbx * nby + n_shards - 1) // n_shards) * bw * bh * 3


def strip_from_frame(frame: np.ndarray, shard: int, n_shards: int) -> np.ndarray:
    """Cuts the strip ``trr_render_shard_device`` would produce for ``shard`` out of a full frame [h, w, 3] u8."""
    h, w = frame.shape[:2]
    bw, bh, nbx, _ = tiling(w, h)
    out = np.zeros(strip_bytes(w, h, n_shards), np.uint8)
    for j, b in enumerate(owned_blocks(w, h, shard, n_shards)):
        xb, yb = b % nbx, b // nbx
        blk = frame[yb * bh:(yb + 1) * bh, xb * bw:(xb + 1) * bw]
        out[j * bw * bh * 3:(j + 1) * bw * bh * 3] = blk.reshape(-1)
    return out


def assemble_host(strips: np.ndarray, width: int, height: int, n_shards: int) -> np.ndarray:
    """``trr_assemble_device`` in numpy: n_shards contiguous strips -> row-major frame; the remainder strip
    (render.rs:100-102) stays 0,0,0."""
    bw, bh, nbx, nby = tiling(width, height)
    sb = strip_bytes(width, height, n_shards)
    strips = np.asarray(strips, np.uint8).reshape(n_shards, sb)
    frame = np.zeros((height, width, 3), np.uint8)
    for b in range(nbx * nby):
        shard, j = b % n_shards, b // n_shards
        xb, yb = b % nbx, b // nbx
        blk = strips[shard, j * bw * bh * 3:(j + 1) * bw * bh * 3].reshape(bh, bw, 3)
        frame[yb * bh:(yb + 1) * bh, xb * bw:(xb + 1) * bw] = blk
    return frame


def gather_strips_host(strip: np.ndarray, rank: int, world: int, group=None) -> Optional[np.ndarray]:
    """One ``torch.distributed.gather`` of equal-size host strips to rank 0 (gloo).  Returns [world, strip_bytes] on
    rank 0, None elsewhere."""
    import torch
    import torch.distributed as dist

    t = torch.from_numpy(np.ascontiguousarray(strip))
    if world == 1:
        return t.numpy().reshape(1, -1)
    outs = [torch.empty_like(t) for _ in range(world)] if rank == 0 else None
    dist.gather(t, outs, dst=0, group=group)
    return torch.stack(outs).numpy() if rank == 0 else None


class ShardedRenderer:
    """One rank of a tile-sharded render.  ``raytracer`` is this rank's ``Raytracer`` (scene already uploaded to this
    rank's GPU).  ``render()`` queues shard render -> gather -> (root) assemble and returns the frame tensor
    [h, w, 3] u8 on the root's GPU (None on the other ranks); nothing is synchronised.

    ``frames_in_flight`` (default 2) sets of buffers and render streams alternate, so that a sequence of frames
    (an animation, a progressive render) keeps the GPU busy across frame boundaries: frame k+1's first block batches
    start while frame k's last persistent trace kernels drain their few long rays.  A frame tensor returned by
    ``render()`` is overwritten by the ``frames_in_flight``-th call after it.  The gathers and the assemble kernels
    of all frames are issued on ONE stream, in frame order, so every rank issues its collectives in the same order."""

    def __init__(self, raytracer, rank: int = 0, world: int = 1, group=None, use_dist: bool = None,
                 frames_in_flight: int = 2, transport: str = None):
        """``transport``: "device" = the strips are gathered GPU to GPU by the process group (nccl = RCCL over xGMI),
        "host" = every rank copies its strip to host memory and the group gathers host buffers (gloo; render() then waits
        for the frame).  Default: by the group's backend."""
        import torch
        import torch.distributed as dist

        self.rt = raytracer
        self.rank, self.world, self.group = int(rank), int(world), group
        self.use_dist = (self.world > 1) if use_dist is None else bool(use_dist)
        if self.world > 1 and not self.use_dist:
            raise ValueError("world > 1 needs the process group (use_dist): the root assembles the strips of every rank")
        if transport is None:
            transport = "host" if (self.use_dist and dist.is_initialized() and dist.get_backend(group) == "gloo") else "device"
        if transport not in ("device", "host"):
            raise ValueError("transport must be 'device' or 'host'")
        self.transport = transport
        self.device = torch.device("cuda", raytracer.ctx.device)
        self.sb = raytracer.strip_bytes(self.world)
        self.is_root = self.rank == 0
        self.n_slots = max(1, int(frames_in_flight))
        self.slots = []
        for _ in range(self.n_slots):
            sl = {"strip": torch.zeros(self.sb, dtype=torch.uint8, device=self.device),
                  # side streams of torch's: the library renders on them (torch's default stream has handle 0, which
                  # the C ABI reads as "the context's own stream")
                  "stream": torch.cuda.Stream(self.device),
                  "rendered": torch.cuda.Event(), "consumed": torch.cuda.Event()}
            if self.is_root:
                sl["strips"] = torch.zeros((self.world, self.sb), dtype=torch.uint8, device=self.device)
                sl["frame"] = torch.zeros((raytracer.height, raytracer.width, 3), dtype=torch.uint8, device=self.device)
                sl["views"] = [sl["strips"][i] for i in range(self.world)]
            self.slots.append(sl)
        # gather + assemble of every frame, in frame order
        self.comm_stream = torch.cuda.Stream(self.device)
        self.frame_no = 0
        torch.cuda.synchronize(self.device)

    def render(self, stats=None, counters: bool = False, defer_stats: bool = False):
        import torch
        import torch.distributed as dist

        sl = self.slots[self.frame_no % self.n_slots]
        self.frame_no += 1
        rs = sl["stream"]
        rs.wait_event(sl["consumed"])          # the strip's previous contents have been gathered / assembled
        with torch.cuda.stream(rs):
            self.rt.render_shard_device(self.rank, self.world, sl["strip"].data_ptr(), self.sb, stream=rs.cuda_stream,
                                        stats=stats, counters=counters, defer_stats=defer_stats, pipelined=self.n_slots >= 2)
            sl["rendered"].record(rs)
        cs = self.comm_stream
        cs.wait_event(sl["rendered"])
        frame = None
        with torch.cuda.stream(cs):
            if self.use_dist and self.transport == "host":
                # host buffers through the group (gloo has no device gather); same strips, same assemble
                sl["rendered"].synchronize()
                host = sl["strip"].cpu()
                outs = [torch.empty_like(host) for _ in range(self.world)] if self.is_root else None
                dist.gather(host, outs, dst=0, group=self.group)
                if self.is_root:
                    sl["strips"].copy_(torch.stack(outs), non_blocking=False)
                src = sl["strips"] if self.is_root else None
            elif self.use_dist:
                # the single exchange of the path: finished RGB8 strips to the root (RCCL over xGMI with backend nccl)
                dist.gather(sl["strip"], sl["views"] if self.is_root else None, dst=0, group=self.group)
                src = sl["strips"] if self.is_root else None
            else:
                src = sl["strip"]
            if self.is_root:
                self.rt.assemble_device(self.world, src.data_ptr(), sl["frame"].data_ptr(), stream=cs.cuda_stream)
                frame = sl["frame"]
            sl["consumed"].record(cs)
        return frame
