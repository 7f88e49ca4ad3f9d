"""N > 1 path on CPU: two (and three) gloo ranks share one frame by the static interleaved
tile partition, gather their compact tile buffers with the PRODUCT's gather_tiles on rank 0,
and the assembled frame must be byte-identical to the single-rank frame.

Without a GPU the pixels cannot come from the HIP kernels, so each rank fills its tiles
from the CPU oracle (test infrastructure) and rank 0 assembles with a numpy mirror of
assemble_kernel (rt_api.hip); what is under test is the partition/layout/gather logic of
ray-tracer_amd/distributed.py.  The GPU side of the same path (rt_render_tiles_async +
rt_assemble_async) is covered by test_gpu_parity.py::test_partition_tiles_assemble_to_full_frame."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, RPP, SEED = 72, 40, 4, 77


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def pack_tiles(rt, full, w, h, spp, rank, nranks):
    """Compact buffer of `rank`: [n_local_tiles][tile_h][tile_w][3], zero outside the image
    -- the layout rt_render_tiles_async writes (include/rt_mi355x.h)."""
    tiles, (tw, th, tx, ty) = rt.tiles_of_rank(w, h, spp, rank, nranks)
    img = full.reshape(h, w, 3)
    out = np.zeros((len(tiles), th, tw, 3), np.float32)
    for l, t in enumerate(tiles):
        x0, y0 = (t % tx) * tw, (t // tx) * th
        blk = img[y0:y0 + th, x0:x0 + tw]
        out[l, :blk.shape[0], :blk.shape[1]] = blk
    return out.reshape(-1), (tw, th, tx, ty)


def assemble_numpy(gathered, nranks, stride, w, h, tw, th, tx, ty):
    """numpy mirror of assemble_kernel (ray-tracer_amd/csrc/rt_api.hip)."""
    out = np.zeros((h, w, 3), np.float32)
    for t in range(tx * ty):
        rank, l = t % nranks, t // nranks
        src = gathered[rank * stride + l * tw * th * 3: rank * stride + (l + 1) * tw * th * 3]
        blk = src.reshape(th, tw, 3)
        x0, y0 = (t % tx) * tw, (t // tx) * th
        hh, ww = min(th, h - y0), min(tw, w - x0)
        out[y0:y0 + hh, x0:x0 + ww] = blk[:hh, :ww]
    return out.reshape(-1, 3)


def _worker(rank, nranks, port, result_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import rt_amd as rt
    from conftest import make_world
    from oracle import binding as ob
    distributed = __import__("importlib").import_module("ray-tracer_amd.distributed")

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank,
                            world_size=nranks)
    try:
        world = make_world(rt, "bench1", W, H, RPP)
        grids = rt.generate_grids(SEED, W + 10, world.camera.aa)
        # this rank's pixels (oracle stands in for the kernels): render only the rows its
        # tiles touch, then keep only its tiles
        tiles, (tw, th, tx, ty) = rt.tiles_of_rank(W, H, world.spp, rank, nranks)
        full = np.zeros((H * W, 3), np.float32)
        for t in tiles:
            y0 = (t // tx) * th
            rgb, _, _ = ob.render(world, grids, ob.FLAG_CULL_ZERO_WEIGHT,
                                  rows=(y0, min(H, y0 + th)), threads=1, want_hit_ids=False)
            x0 = (t % tx) * tw
            a = full.reshape(H, W, 3)
            a[y0:y0 + th, x0:x0 + tw] = rgb.reshape(H, W, 3)[y0:y0 + th, x0:x0 + tw]
        mine_np, _ = pack_tiles(rt, full, W, H, world.spp, rank, nranks)
        tile_floats, stride = distributed.partition_layout(tx * ty, tw, th, nranks)
        assert mine_np.size <= stride
        mine = torch.zeros(stride, dtype=torch.float32)
        mine[:mine_np.size] = torch.from_numpy(mine_np)
        gathered = torch.zeros(nranks * stride, dtype=torch.float32) if rank == 0 else None
        distributed.gather_tiles(mine, gathered, dst=0)
        if rank == 0:
            frame = assemble_numpy(gathered.numpy(), nranks, stride, W, H, tw, th, tx, ty)
            ref, _, _ = ob.render(world, grids, ob.FLAG_CULL_ZERO_WEIGHT, want_hit_ids=False)
            np.save(result_path, np.array([np.array_equal(frame, ref), float(np.abs(frame - ref).max())]))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("nranks", [2, 3])
def test_tiles_gathered_over_gloo_equal_single_rank_frame(tmp_path, nranks):
    result = os.path.join(tmp_path, "ok.npy")
    mp.spawn(_worker, args=(nranks, _free_port(), result), nprocs=nranks, join=True)
    ok, maxdiff = np.load(result)
    assert ok == 1.0, f"assembled frame differs from the single-rank frame by {maxdiff}"


def test_partition_layout():
    import importlib
    sys.path.insert(0, ROOT)
    d = importlib.import_module("ray-tracer_amd.distributed")
    assert d.partition_layout(10, 8, 8, 1) == (192, 1920)
    assert d.partition_layout(10, 8, 8, 3) == (192, 4 * 192)
    assert d.partition_layout(32400, 8, 8, 8) == (192, 4050 * 192)


# ---- DistributedFrame itself with N > 1: slots, works, events, mode broadcast -------------------

class _StubRenderer:
    """What DistributedFrame asks of an rt.Renderer, computed on the host: a frame is a pattern
    of (frame number, global tile, pixel in tile); "kernels" complete inside the call.  Only the
    bookkeeping of DistributedFrame and the collective are real here."""
    W, H, SPP = 72, 40, 4

    def __init__(self, rt, counter):
        from types import SimpleNamespace
        self.rt = rt
        self.world = SimpleNamespace(width=self.W, height=self.H)
        self.stream = 0
        self.rank, self.nranks = 0, 1
        self.counter = counter          # frames rendered by ALL contexts of this rank, in submit order
        self.concurrency = None
        self.rendered = []

    def set_timing(self, on): pass
    def set_concurrency(self, n): self.concurrency = n
    def close(self): pass

    def set_partition(self, rank, nranks):
        self.rank, self.nranks = rank, nranks

    def tiling(self):
        from types import SimpleNamespace
        tiles, (tw, th, tx, ty) = self.rt.tiles_of_rank(self.W, self.H, self.SPP, self.rank, self.nranks)
        return SimpleNamespace(tile_w=tw, tile_h=th, tiles_x=tx, tiles_y=ty, n_tiles=tx * ty, n_local_tiles=len(tiles))

    @staticmethod
    def _view(ptr, n):
        import ctypes
        return np.ctypeslib.as_array((ctypes.c_float * n).from_address(ptr))

    @staticmethod
    def pattern(frame_no, tile, tw, th):
        px = np.arange(tw * th, dtype=np.float32).reshape(th, tw, 1)
        return (np.float32(frame_no * 1000 + tile) + px / np.float32(4096.0)) * np.ones((1, 1, 3), np.float32)

    def render_tiles_async(self, ptr, flags, stream):
        til = self.tiling()
        tiles, _ = self.rt.tiles_of_rank(self.W, self.H, self.SPP, self.rank, self.nranks)
        out = self._view(ptr, til.n_local_tiles * til.tile_w * til.tile_h * 3).reshape(len(tiles), til.tile_h, til.tile_w, 3)
        frame_no = self.counter[0]
        self.counter[0] += 1
        self.rendered.append(frame_no)
        for l, t in enumerate(tiles):
            out[l] = self.pattern(frame_no, t, til.tile_w, til.tile_h)

    def assemble_async(self, gathered_ptr, nranks, stride, frame_ptr, stream):
        til = self.tiling()
        g = self._view(gathered_ptr, nranks * stride)
        f = self._view(frame_ptr, self.W * self.H * 3)
        f[:] = assemble_numpy(g, nranks, stride, self.W, self.H, til.tile_w, til.tile_h, til.tiles_x, til.tiles_y).reshape(-1)


def _frame_worker(rank, nranks, port, result_path):
    sys.path.insert(0, ROOT)
    import rt_amd as rt
    distributed = __import__("importlib").import_module("ray-tracer_amd.distributed")
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=nranks)
    try:
        counter = [0]
        dev = torch.device("cpu")
        # share_gpu "auto": every rank times the modes on its own; rank 0's choice must become the job's
        df = distributed.DistributedFrame(lambda: _StubRenderer(rt, counter), rank, nranks, dev, in_flight=3,
                                          share_gpu="auto", streams=distributed.HostStreams(dev))
        counter[0] = 0                                   # (the start-up timing rendered frames of its own)
        modes = [None] * nranks
        dist.all_gather_object(modes, tuple(df.mode))
        ok = len(set(modes)) == 1 and all(r.concurrency == df.mode[1] for r in df.renderers)
        n_frames = 7                                      # slots are reused: 7 frames over <= 3 slots
        seen = []
        for k in range(n_frames):
            df.submit(0)
            if k % 2 == 1:                                # finish in the middle of the sequence as well
                df.finish()
                if rank == 0:
                    seen.append((k, df.frame.clone()))
        df.finish()
        if rank == 0:
            seen.append((n_frames - 1, df.frame.clone()))
            til = df.r.tiling()
            for k, fr in seen:
                want = np.zeros((_StubRenderer.H, _StubRenderer.W, 3), np.float32)
                for t in range(til.n_tiles):
                    x0, y0 = (t % til.tiles_x) * til.tile_w, (t // til.tiles_x) * til.tile_h
                    blk = _StubRenderer.pattern(k, t, til.tile_w, til.tile_h)
                    hh, ww = min(til.tile_h, _StubRenderer.H - y0), min(til.tile_w, _StubRenderer.W - x0)
                    want[y0:y0 + hh, x0:x0 + ww] = blk[:hh, :ww]
                ok = ok and np.array_equal(fr.numpy()[:want.size], want.reshape(-1))
            np.save(result_path, np.array([float(ok), float(df.mode[0]), float(df.mode[1])]))
        dist.barrier()
        df.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("nranks", [2, 3])
def test_distributed_frame_pipeline_over_gloo(tmp_path, nranks):
    """DistributedFrame.submit / finish with N > 1 ranks (gloo, host streams, a stub renderer): frame
    k of the sequence must arrive whole on rank 0 from every rank's k-th render, through reused
    slots, and all ranks must run rank 0's (frames in flight, grid fraction)."""
    result = os.path.join(tmp_path, "ok.npy")
    mp.spawn(_frame_worker, args=(nranks, _free_port(), result), nprocs=nranks, join=True)
    ok, depth, div = np.load(result)
    assert ok == 1.0, f"frames or modes differ (mode {int(depth)}, 1/{int(div)})"
