"""Data-parallel rendering of ONE frame over the GPUs of a node.

The path shards over pixels: tile t of the frame belongs to rank t % nranks (static,
interleaved, so every rank gets a spatially uniform sample of cheap and expensive
regions).  The scene and the sample table are replicated.  Each rank renders its tiles
into a compact buffer [n_local_tiles][tile_h][tile_w][3]; the ONE exchange step of the
path is the gather of those buffers on rank 0 (RCCL over xGMI when the process group's
backend is "nccl"), followed by rt_assemble_async, which scatters them into the frame.

Frames are pipelined: `in_flight` contexts (rt_ctx) render alternate frames on their own
streams, so the kernel of frame k+1 overlaps both the tail of frame k's kernel and its
gather (which runs on the collective's stream); at N ranks the per-frame cost tends to
max(average wave time, gather) rather than slowest wave + gather.

torch is used for device memory, streams and torch.distributed only.
"""
from __future__ import annotations

import os
import sys
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def partition_layout(n_tiles: int, tile_w: int, tile_h: int, nranks: int) -> Tuple[int, int]:
    """(tile_floats, stride_floats): every rank's compact buffer is padded to the size of
    the largest share, ceil(n_tiles / nranks) tiles, so the gather is regular."""
    tile_floats = tile_w * tile_h * 3
    return tile_floats, ((n_tiles + nranks - 1) // nranks) * tile_floats


def compact_floats(tiling) -> int:
    """Floats rt_render_tiles_async writes for a context whose rt_get_tiling is `tiling`:
    n_local_tiles whole tiles.  The tile grid is padded to whole tiles, so with one rank this
    EXCEEDS img_w * img_h * 3 whenever the image sides are not multiples of the tile sides
    (800x600 at 1 spp: 25 x 19 tiles of 32 x 32 pixels = 1 459 200 floats for a 1 440 000-float
    frame) -- every buffer handed to render_tiles_async is sized with this, never from W * H."""
    return tiling.n_local_tiles * tiling.tile_w * tiling.tile_h * 3


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


class _Gpu:
    """The stream / event layer DistributedFrame is written against: torch.cuda on a GPU."""

    def __init__(self, device: torch.device):
        self.device = device

    def wrap_stream(self, handle):          # a context's own hipStream_t
        return torch.cuda.ExternalStream(handle, device=self.device)

    def new_stream(self):
        return torch.cuda.Stream(device=self.device)

    def new_event(self):
        return torch.cuda.Event()

    def use(self, stream):                  # context manager: `stream` is current inside
        return torch.cuda.stream(stream)

    def synchronize(self) -> None:
        torch.cuda.synchronize(self.device)


class HostStreams:
    """The same layer for a host-only rehearsal (tests/test_distributed_gloo.py drives
    DistributedFrame.submit / finish over gloo with a stub renderer whose "kernels" run
    synchronously): every operation completes when it is called, so streams and events are
    placeholders -- what runs for real is the slot / work / event BOOKKEEPING and the collective."""

    class _Stream:
        cuda_stream = 0

        def wait_event(self, ev) -> None:
            pass

    class _Event:
        def record(self, stream=None) -> None:
            pass

    class _Use:
        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

    def __init__(self, device: torch.device = torch.device("cpu")):
        self.device = device

    def wrap_stream(self, handle):
        return HostStreams._Stream()

    def new_stream(self):
        return HostStreams._Stream()

    def new_event(self):
        return HostStreams._Event()

    def use(self, stream):
        return HostStreams._Use()

    def synchronize(self) -> None:
        pass


class DistributedFrame:
    """One rank's side of the tiled, gathered, pipelined frame render.

    make_renderer: callable returning an rt.Renderer (one rt_ctx) with world and samples set,
    on `device`.  `in_flight` contexts render alternate frames on their own streams: a
    context owns every per-frame buffer (pixel sums, ray queues, ticket counters), so frame
    k+1's kernel may start while frame k's is still running, and the hardware places its
    blocks into the wave slots that frame k's tail (a few deep ray trees) leaves empty.
    After `finish()`, rank 0 holds the newest frames in `self.frames` ([H*W*3] float32 each,
    render's pixel order)."""

    def __init__(self, make_renderer, rank: int, nranks: int, device: torch.device, in_flight: int = 3,
                 force_collective: bool = False, share_gpu: str = "auto",
                 rehearse_share: Optional[Tuple[int, int]] = None, tune_worlds=None, streams=None):
        """rehearse_share=(r, n): timing rehearsal on one GPU of what rank r of an n-rank job
        computes (its tiles only, no collective, the frame buffers then hold compact tiles).
        tune_worlds: callable k -> World for jobs that upload a new scene per frame (animation);
        the start-up timing then does the same, since the upload serialises a context's frames.
        in_flight: the most contexts (frames in flight) to use.  share_gpu: "auto" times a few
        frames at start-up for a handful of (frames in flight F, grid fraction 1/d) pairs -- each
        context launches 1/d of the GPU's wave slots (rt_set_concurrency) -- and keeps the
        fastest; which one wins depends on the frame's size (a small share of a frame wants
        fewer, longer-lived waves: less start-up and queue-drain cost per ray) and on how the
        runtime maps the contexts' streams onto hardware queues.  "yes" forces (in_flight,
        1/in_flight), "no" (in_flight, whole GPU).  The choice never changes a pixel."""
        self.rank, self.nranks, self.device = rank, nranks, device
        self.gpu = streams if streams is not None else _Gpu(device)   # (a HostStreams in the CPU rehearsal)
        self.depth = max(1, in_flight)
        self.renderers = [make_renderer() for _ in range(self.depth)]
        for r in self.renderers:
            r.set_timing(False)   # throughput contexts: four stream operations per frame, not eight
        self.rehearsal = rehearse_share is not None
        self.tune_worlds = tune_worlds
        for r in self.renderers:
            r.set_partition(*(rehearse_share if self.rehearsal else (rank, nranks)))
        self.mode = self._choose_mode(share_gpu)
        if nranks > 1 and not self.rehearsal and dist.is_available() and dist.is_initialized():
            # ONE mode for the job: rank 0's choice (each rank timed its own share; the shares are
            # alike, but "within a per cent of each other" must not leave the ranks with different
            # numbers of frames in flight)
            t = torch.tensor(list(self.mode), dtype=torch.int64, device=device)
            dist.broadcast(t, src=0)
            depth, div = int(t[0].item()), int(t[1].item())
            self.depth = max(1, min(depth, len(self.renderers)))
            for r in self.renderers:
                r.set_concurrency(div)
            self.mode = (self.depth, div)
        self.r = self.renderers[0]
        til = self.r.tiling()
        self.tile_floats, self.stride = partition_layout(til.n_tiles, til.tile_w, til.tile_h, nranks)
        w = self.r.world
        self.is_root = rank == 0
        # (a rehearsal renders its share as compact tiles into the frame buffers: whole tiles)
        frame_floats = max(w.height * w.width * 3, compact_floats(til) if self.rehearsal else 0)
        self.frames = [torch.zeros(frame_floats, dtype=torch.float32, device=device)
                       for _ in range(self.depth)] if self.is_root else [None] * self.depth
        # every context renders on ITS OWN stream (created by rt_init, one hardware queue each);
        # torch only wraps it, so that events and the collective's stream can be ordered
        # against it.  (Two streams from torch's pool were measured to land on one hardware
        # queue: no overlap at all.)
        self.streams = [self.gpu.wrap_stream(r.stream) for r in self.renderers]
        self.k = 0
        # force_collective: run the gather/assemble path even with one rank (self-test on a
        # one-GPU box of exactly the code the 8-GPU job runs)
        self.collective = nranks > 1 or force_collective
        self.poison = force_collective
        if not self.collective:
            return
        self.mine = [torch.zeros(self.stride, dtype=torch.float32, device=device)
                     for _ in range(self.depth)]
        self.gathered = [torch.zeros(nranks * self.stride, dtype=torch.float32, device=device)
                         for _ in range(self.depth)] if self.is_root else [None] * self.depth
        self.works = [None] * self.depth
        self.asm_stream = self.gpu.new_stream()
        self.asm_done = [None] * self.depth
        self.own_copied = [self.gpu.new_event() for _ in range(self.depth)]

    def _choose_mode(self, share_gpu: str):
        """(frames in flight, grid divisor) -- local timing only, no collective."""
        if self.depth == 1:
            return (1, 1)
        if share_gpu in ("yes", "no"):
            div = self.depth if share_gpu == "yes" else 1
            for r in self.renderers:
                r.set_concurrency(div)
            return (self.depth, div)
        import time
        scratch = [torch.zeros(max(64, compact_floats(self.renderers[0].tiling())), dtype=torch.float32,
                               device=self.device) for _ in range(self.depth)]
        candidates = [(1, 1)]
        for f in range(2, self.depth + 1):
            # (three and more whole-GPU grids in flight never won anywhere: not timed)
            candidates += [(f, f - 1), (f, f)] if f > 2 else [(f, 1), (f, f)]
        if self.depth >= 4:
            # headroom: the contexts' persistent waves leave a fifth of the wave slots to the small
            # kernels between the frames (probe, order, resolve), which otherwise queue for slots
            candidates += [(self.depth, self.depth + 1)]
        best = None
        for depth, div in candidates:
            for r in self.renderers:
                r.set_concurrency(div)
            times = []
            n_frames = 4 * self.depth
            for rep in range(4):                 # the first repetition warms up and is dropped
                self.gpu.synchronize()
                t0 = time.perf_counter()
                for k in range(n_frames):
                    j = k % depth
                    if self.tune_worlds is not None:
                        self.renderers[j].set_world(self.tune_worlds(k))
                    self.renderers[j].render_tiles_async(scratch[j].data_ptr(), 0, 0)
                self.gpu.synchronize()
                times.append(time.perf_counter() - t0)
            t = min(times[1:])
            if os.environ.get("RT_TUNE_DEBUG"):
                print(f"[tune] in flight {depth}, grid 1/{div}: " + " ".join(f"{x / n_frames * 1e3:.3f}" for x in times) + " ms/frame",
                      file=sys.stderr, flush=True)
            if best is None or t < best[0] * 0.99:   # prefer the simpler mode unless 1 % faster
                best = (t, depth, div)
        _, depth, div = best
        for r in self.renderers:
            r.set_concurrency(div)
        self.depth = depth
        return (depth, div)

    @property
    def frame(self):
        """The most recently submitted frame (valid after finish())."""
        return self.frames[(self.k - 1) % self.depth] if self.k else self.frames[0]

    def submit(self, flags: int = 0, world=None, refit: bool = False, device_build: bool = False) -> None:
        """Enqueue one frame; returns without waiting for the GPU.  `world` (animation): the
        frame's own World, uploaded to the frame's context first (rt_upload_scene frees the
        previous scene, which waits for that context's frame in flight); with `refit` the context
        keeps its tree and only refits the boxes to the world's objects on the GPU (rt_refit_async);
        with `device_build` the world's own tree is ignored and one is built on the GPU (rt_build_scene:
        a tree-changing mode, see include/rt_mi355x.h)."""
        slot = self.k % self.depth
        rs, r = self.streams[slot], self.renderers[slot]
        if world is not None:
            if device_build:
                r.build_world(world)     # the tree built on the GPU from the world's objects (rt_build_scene)
            elif refit:
                r.refit(world.scene)
            else:
                r.set_world(world)
        if not self.collective:
            if self.rehearsal:
                r.render_tiles_async(self.frames[slot].data_ptr(), flags, rs.cuda_stream)
            else:
                r.render_async(self.frames[slot].data_ptr(), 0, flags, rs.cuda_stream)
            self.k += 1
            return
        with self.gpu.use(rs):
            if self.works[slot] is not None:
                self.works[slot].wait()          # mine[slot] is free once its gather is done
            if self.asm_done[slot] is not None:
                rs.wait_event(self.asm_done[slot])  # gathered[slot]/frames[slot] free once assembled
            if self.poison:
                # self-test: whatever is read before it was written this frame shows as NaN (a still
                # frame rendered twice would otherwise hide a premature read behind the old copy)
                self.mine[slot].fill_(float("nan"))
                if self.is_root:
                    self.gathered[slot].fill_(float("nan"))
                    self.frames[slot].fill_(float("nan"))
            r.render_tiles_async(self.mine[slot].data_ptr(), flags, rs.cuda_stream)
            # the collective's stream picks up after the kernel just enqueued on rs
            self.works[slot] = gather_tiles(self.mine[slot], self.gathered[slot], 0, async_op=True)
            if self.is_root:
                # The root's OWN tiles reach gathered[slot] by a device copy that the process group
                # enqueues on the stream that is current at the call (rs), not on the collective's:
                # waiting for the collective's work does not cover it.  (Seen with one rank, 6 frames
                # in flight, first use of a slot: 2 % of the frame still zero when assembled.)
                self.own_copied[slot].record(rs)
        if self.is_root:
            with self.gpu.use(self.asm_stream):
                self.works[slot].wait()
                self.asm_stream.wait_event(self.own_copied[slot])
                r.assemble_async(self.gathered[slot].data_ptr(), self.nranks, self.stride,
                                 self.frames[slot].data_ptr(), self.asm_stream.cuda_stream)
                ev = self.gpu.new_event()
                ev.record(self.asm_stream)
                self.asm_done[slot] = ev
        self.k += 1

    def finish(self) -> None:
        """Wait until every submitted frame is rendered, gathered and assembled."""
        if self.collective:
            for w in self.works:
                if w is not None:
                    w.wait()
        self.gpu.synchronize()

    def close(self) -> None:
        for r in self.renderers:
            r.close()
