#!/usr/bin/env python3
"""Host rebuild + upload against GPU refit for a large animated mesh (run on the GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
import rt_amd as rt

world = bench.make_world("c3")
grids = rt.generate_grids(1, world.width + 10, world.camera.aa)
r = rt.Renderer(0)
r.set_world(world); r.set_samples(grids)
a, _, st = r.render(0)
t0 = time.perf_counter(); s2 = rt.Scene.builtin("mesh:100000"); t1 = time.perf_counter()
w2 = rt.config_to_world(bench.rt.default_config("mesh"), s2)
t2 = time.perf_counter(); r.set_world(world); r.collect(); t3 = time.perf_counter()
ts = []
for _ in range(5):
    t4 = time.perf_counter(); r.refit(world.scene); r.collect(); ts.append(time.perf_counter() - t4)
b, _, st2 = r.render(0)
print(f"mesh:100000  host sahBVH build {1e3*(t1-t0):.1f} ms, rt_upload_scene {1e3*(t3-t2):.1f} ms, rt_refit_async (copy + kernels, waited for) "
      f"first {1e3*ts[0]:.2f} ms then {1e3*min(ts[1:]):.2f} ms; frame after refit identical: {np.array_equal(a.view(np.uint32), b.view(np.uint32))}")
r.close()
