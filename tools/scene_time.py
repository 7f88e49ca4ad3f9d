#!/usr/bin/env python3
"""Frame time of any built-in scene (GPU box):  tools/scene_time.py bench3 [width height rays_per_pixel]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rt_amd as rt
name = sys.argv[1] if len(sys.argv) > 1 else "bench3"
W, H, rpp = (int(x) for x in (sys.argv[2:5] + ["1920", "1080", "4"][len(sys.argv[2:5]):]))
cfg = rt.default_config(name if not name.startswith("mesh:") else "mesh")
cfg.image_width, cfg.image_height, cfg.anti_aliasing = W, H, rpp
if W * 9 == H * 16 and not name.startswith("mesh"):
    cfg.view_width, cfg.view_height, cfg.view_distance = 16, 9, 10
world = rt.config_to_world(cfg, rt.Scene.builtin(name))
grids = rt.generate_grids(7, W + 10, world.camera.aa)
r = rt.Renderer(0); r.set_world(world); r.set_samples(grids)
buf = torch.zeros(H * W * 3, dtype=torch.float32, device="cuda:0")
for _ in range(3):
    r.render_async(buf.data_ptr(), 0, 0); st = r.collect()
ts = []
for _ in range(8):
    r.render_async(buf.data_ptr(), 0, 0); st = r.collect(); ts.append(st.ms_launch)
print(f"{name} {W}x{H} {world.spp} spp: {min(ts):.3f} ms/frame (single frame), {st.rays_total / min(ts) / 1e3:.0f} Mrays/s, "
      f"rays {st.rays_total}", flush=True)
r.close()
