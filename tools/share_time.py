#!/usr/bin/env python3
"""Per-frame time of rank 0's share of an N-way partition on ONE GPU, for the three pipelining
modes DistributedFrame chooses between (1 frame in flight; F frames, whole-GPU grids; F frames
sharing the wave slots).  No collective: compute side of the strong-scaling story only.
  tools/share_time.py [workload ...]      (N = 1 and 8; F frames in flight from env F, default 2)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rt_amd as rt
from bench import make_world, SEED
F = int(os.environ.get('F', '2'))
for wl in (sys.argv[1:] or ["c2"]):
    world = make_world(wl)
    grids = rt.generate_grids(SEED, world.width + 10, world.camera.aa)
    rs = []
    for i in range(F):
        r = rt.Renderer(0); r.set_world(world); r.set_samples(grids); rs.append(r)
    bufs = [torch.zeros(world.height * world.width * 3 + 4096, dtype=torch.float32, device="cuda:0") for _ in rs]
    for n in (1, 8):
        for r in rs:
            r.set_partition(0, n)
        line = []
        for depth, share in ((1, False), (F, False), (F, True)):
            for r in rs:
                r.set_concurrency(depth if share else 1)
            best = 1e9
            for rep in range(4):
                torch.cuda.synchronize()
                K = 16
                t0 = time.perf_counter()
                for k in range(K):
                    rs[k % depth].render_tiles_async(bufs[k % depth].data_ptr(), 0, 0)
                torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - t0) / K * 1e3)
            line.append(f"{depth} in flight{' shared' if share else ''}: {best:.3f} ms")
        print(f"{wl} 1/{n} of the frame | " + " | ".join(line), flush=True)
    for r in rs:
        r.close()
