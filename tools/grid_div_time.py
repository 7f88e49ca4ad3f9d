#!/usr/bin/env python3
"""Frame time of rank 0's 1/N share against the fraction of the GPU's wave slots each context
launches (rt_set_concurrency), F contexts in flight.  tools/grid_div_time.py [workload] [N]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rt_amd as rt
from bench import make_world, SEED
wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
world = make_world(wl)
grids = rt.generate_grids(SEED, world.width + 10, world.camera.aa)
rs = []
for i in range(4):
    r = rt.Renderer(0); r.set_world(world); r.set_samples(grids); r.set_partition(0, n); rs.append(r)
bufs = [torch.zeros(world.height * world.width * 3 + 4096, dtype=torch.float32, device="cuda:0") for _ in rs]
for F in (1, 2, 3):
    line = []
    for div in (1, 2, 3, 4, 6, 8):
        for r in rs:
            r.set_concurrency(div)
        best = 1e9
        for rep in range(4):
            torch.cuda.synchronize()
            K = 24
            t0 = time.perf_counter()
            for k in range(K):
                rs[k % F].render_tiles_async(bufs[k % F].data_ptr(), 0, 0)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / K * 1e3)
        line.append(f"1/{div}: {best:.3f}")
    print(f"{wl} 1/{n} share, {F} in flight, ms/frame by grid fraction | " + " | ".join(line), flush=True)
for r in rs:
    r.close()
