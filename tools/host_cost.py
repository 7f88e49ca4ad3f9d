#!/usr/bin/env python3
"""Host cost of submitting one frame (rt_render_async on K contexts, then rt_collect on all): is a small
workload bound by the host's launches or by the GPU?  usage: python tools/host_cost.py [workload] [K]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import rt_amd as rt

wl = sys.argv[1] if len(sys.argv) > 1 else "c1"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 8
world = bench.make_world(wl)
grids = rt.generate_grids(bench.SEED, world.width + 10, world.camera.aa)
rs = [rt.Renderer(0) for _ in range(K)]
bufs = [torch.empty(world.height * world.width * 3, dtype=torch.float32, device="cuda:0") for _ in range(K)]
for r in rs:
    r.set_world(world); r.set_samples(grids); r.set_concurrency(K); r.set_timing(False)
for rep in range(4):
    t0 = time.perf_counter()
    for r, b in zip(rs, bufs):
        r.render_async(b.data_ptr())
    t1 = time.perf_counter()
    for r in rs:
        r.collect()
    t2 = time.perf_counter()
    print(f"{wl}: submit {1e6 * (t1 - t0) / K:.1f} us per frame; all {K} frames done {1e6 * (t2 - t0):.0f} us after the first submit")
