#!/usr/bin/env python3
"""Throughput with F frames in flight (F contexts on F streams, one GPU): the next frame's
blocks fill the wave slots that the current frame's tail leaves empty."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rt_amd as rt
from bench import make_world, SEED
wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
nr = int(sys.argv[2]) if len(sys.argv) > 2 else 1
world = make_world(wl)
grids = rt.generate_grids(SEED, world.width + 10, world.camera.aa)
for F in (2, 4):
    rs, bufs, streams = [], [], []
    for i in range(F):
        r = rt.Renderer(0); r.set_world(world); r.set_samples(grids); r.set_partition(0, nr)
        rs.append(r)
        bufs.append(torch.zeros(world.height * world.width * 3 + 4096, dtype=torch.float32, device="cuda:0"))
        streams.append(torch.cuda.Stream())
    def run(k):
        for i in range(k):
            j = i % F
            (rs[j].render_async(bufs[j].data_ptr(), 0, 0, streams[j].cuda_stream) if os.environ.get("FULL") else rs[j].render_tiles_async(bufs[j].data_ptr(), 0, streams[j].cuda_stream))
        torch.cuda.synchronize()
    run(6)
    K = 30
    t0 = time.perf_counter(); run(K); dt = time.perf_counter() - t0
    st = rs[0].collect(streams[0].cuda_stream)
    print(f"{wl} ranks={nr} frames in flight {F}: {dt/K*1e3:.3f} ms/frame  {st.rays_total*K/dt/1e6:.0f} Mrays/s", flush=True)
    same = all(torch.equal(bufs[0], b) for b in bufs[1:])
    print("   frames identical:", same)
    for r in rs: r.close()
