#!/usr/bin/env python3
"""Per-rank share of a frame on ONE GPU: time rank 0's tiles of an N-way partition
(N = 1,2,4,8).  t(N=1) / (N * t(N)) estimates the compute-side strong-scaling efficiency
(fixed per-launch overheads and tail effects; the gather is not included)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rt_amd as rt
from bench import make_world, SEED
wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
world = make_world(wl)
grids = rt.generate_grids(SEED, world.width + 10, world.camera.aa)
r = rt.Renderer(0); r.set_world(world); r.set_samples(grids)
buf = torch.zeros(world.height * world.width * 3 + 4096, dtype=torch.float32, device="cuda:0")
base = None
for n in (1, 8):
    for rank in ((0,) if n == 1 else (0, n - 1)):
        r.set_partition(rank, n)
        for _ in range(3):
            r.render_tiles_async(buf.data_ptr()); r.collect()
        ts = []
        for _ in range(8):
            t0 = time.perf_counter(); r.render_tiles_async(buf.data_ptr()); st = r.collect(); ts.append((time.perf_counter() - t0) * 1e3)
        ms = st.ms_launch
        if base is None: base = ms
        print(f"{wl} N={n} rank={rank}: launch {ms:.3f} ms (host wall {min(ts):.3f}) rays {st.rays_total}  eff {base / (n * ms):.3f}", flush=True)
r.close()
