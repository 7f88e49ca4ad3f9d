#!/usr/bin/env python3
"""Frames per second through the one-process C ABI (rt_multi_submit / rt_multi_wait) with K rt_multi
in flight on one GPU, 8-bit frames handed to host memory (GPU box).
usage: tools/multi_pipeline.py [workload] [ranks sharing the GPU]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import numpy as np
import rt_amd as rt
wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
nr = int(sys.argv[2]) if len(sys.argv) > 2 else 1
world = bench.make_world(wl)
grids = rt.generate_grids(bench.SEED, world.width + 10, world.camera.aa)
for K, div in ((1, 1), (2, 1), (2, 2), (3, 3), (4, 3), (4, 4)):
    ms = [rt.MultiRenderer([0] * nr) for _ in range(K)]
    for m in ms:
        m.set_world(world); m.set_samples(grids); m.set_concurrency(div)
    for m in ms:
        m.submit(0, want_float=False, want_8bit=True)
    n = 0
    frames = 24
    for warm in (True, False):
        t0 = time.perf_counter()
        for k in range(frames):
            m = ms[k % K]
            _, rgb8, st = m.wait()
            m.submit(0, want_float=False, want_8bit=True)
        dt = time.perf_counter() - t0
    for m in ms:
        m.wait()
    print(f"{wl}  ranks {nr}  rt_multi in flight {K}, 1/{div} of the wave slots each: {dt / frames * 1e3:.3f} ms per frame (8-bit frame in host memory)", flush=True)
    for m in ms:
        m.close()
