"""Data-parallel rendering of ONE frame over the GPUs of a node.

The path shards over pixels: tile t of the frame belongs to rank t % nranks (static,
interleaved, so every rank gets a spatially uniform sample of cheap and expensive
regions).  The scene and the sample table are replicated.  Each rank renders its tiles
into a compact buffer [n_local_tiles][tile_h][tile_w][3]; the ONE exchange step of the
path is the gather of those buffers on rank 0 (RCCL over xGMI when the process group's
backend is "nccl"), followed by rt_assemble_async, which scatters them into the frame.

Frames are pipelined: the gather of frame k runs on the collective's stream while the
trace/shade kernel of frame k+1 runs on the render stream (double-buffered tile
buffers), so at N ranks the per-frame cost is max(render/N, gather) rather than their sum.

torch is used for device memory, streams and torch.distributed only.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def partition_layout(n_tiles: int, tile_w: int, tile_h: int, nranks: int) -> Tuple[int, int]:
    """(tile_floats, stride_floats): every rank's compact buffer is padded to the size of
    the largest share, ceil(n_tiles / nranks) tiles, so the gather is regular."""
    tile_floats = tile_w * tile_h * 3
    return tile_floats, ((n_tiles + nranks - 1) // nranks) * tile_floats


def gather_tiles(mine: torch.Tensor, gathered: Optional[torch.Tensor], dst: int = 0,
                 async_op: bool = False):
    """Gather the ranks' compact tile buffers (all `mine.numel()` floats) on `dst`, where
    `gathered` is a flat tensor of world_size * mine.numel() floats, rank-major -- the
    layout rt_assemble_async expects.  Backend-agnostic (nccl = RCCL on GPUs, gloo on CPU)."""
    world = dist.get_world_size()
    glist: Optional[List[torch.Tensor]] = None
    if dist.get_rank() == dst:
        assert gathered is not None and gathered.numel() == world * mine.numel()
        glist = list(gathered.split(mine.numel()))
    return dist.gather(mine, glist, dst=dst, async_op=async_op)


class DistributedFrame:
    """One rank's side of the tiled, gathered, pipelined frame render.

    renderer: an rt.Renderer with world and samples already set, on `device`.
    After `finish()`, rank 0 holds the newest frame in `self.frame` ([H*W*3] float32,
    render's pixel order)."""

    def __init__(self, renderer, rank: int, nranks: int, device: torch.device, pipeline: int = 2):
        self.r, self.rank, self.nranks, self.device = renderer, rank, nranks, device
        renderer.set_partition(rank, nranks)
        til = renderer.tiling()
        self.tile_floats, self.stride = partition_layout(til.n_tiles, til.tile_w, til.tile_h, nranks)
        w = renderer.world
        self.is_root = rank == 0
        self.frame = torch.zeros(w.height * w.width * 3, dtype=torch.float32, device=device) \
            if self.is_root else None
        self.render_stream = torch.cuda.Stream(device=device)
        self.k = 0
        if nranks == 1:
            return
        self.depth = max(1, pipeline)
        self.mine = [torch.zeros(self.stride, dtype=torch.float32, device=device)
                     for _ in range(self.depth)]
        self.gathered = [torch.zeros(nranks * self.stride, dtype=torch.float32, device=device)
                         for _ in range(self.depth)] if self.is_root else [None] * self.depth
        self.works = [None] * self.depth
        self.asm_stream = torch.cuda.Stream(device=device)
        self.asm_done = [None] * self.depth

    def submit(self, flags: int = 0) -> None:
        """Enqueue one frame; returns without waiting for the GPU."""
        rs = self.render_stream
        if self.nranks == 1:
            self.r.render_async(self.frame.data_ptr(), 0, flags, rs.cuda_stream)
            self.k += 1
            return
        slot = self.k % self.depth
        with torch.cuda.stream(rs):
            if self.works[slot] is not None:
                self.works[slot].wait()          # mine[slot] is free once its gather is done
            if self.asm_done[slot] is not None:
                rs.wait_event(self.asm_done[slot])  # gathered[slot] is free once assembled
            self.r.render_tiles_async(self.mine[slot].data_ptr(), flags, rs.cuda_stream)
            # the collective's stream picks up after the kernel just enqueued on rs
            self.works[slot] = gather_tiles(self.mine[slot], self.gathered[slot], 0, async_op=True)
        if self.is_root:
            with torch.cuda.stream(self.asm_stream):
                self.works[slot].wait()
                self.r.assemble_async(self.gathered[slot].data_ptr(), self.nranks, self.stride,
                                      self.frame.data_ptr(), self.asm_stream.cuda_stream)
                ev = torch.cuda.Event()
                ev.record(self.asm_stream)
                self.asm_done[slot] = ev
        self.k += 1

    def finish(self) -> None:
        """Wait until every submitted frame is rendered, gathered and assembled."""
        if self.nranks > 1:
            for w in self.works:
                if w is not None:
                    w.wait()
        torch.cuda.synchronize(self.device)
