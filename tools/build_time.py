#!/usr/bin/env python3
"""rt_build_scene (tree built on the GPU) against the host path (sahBVH + rt_upload_scene), and the
frame each tree renders (GPU box):  tools/build_time.py [scene] [width height rays_per_pixel]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rt_amd as rt
name = sys.argv[1] if len(sys.argv) > 1 else "mesh:100000"
W, H, rpp = (int(x) for x in (sys.argv[2:5] + ["1920", "1080", "4"][len(sys.argv[2:5]):]))
cfg = rt.default_config(name if not name.startswith("mesh:") else "mesh")
cfg.image_width, cfg.image_height, cfg.anti_aliasing = W, H, rpp
cfg.view_width, cfg.view_height, cfg.view_distance = 16, 9, 10
t0 = time.perf_counter()
scene = rt.Scene.builtin(name)
t_scene = time.perf_counter() - t0
t0 = time.perf_counter()
scene.build_bvh("sah")
t_sah = time.perf_counter() - t0
world = rt.config_to_world(cfg, scene)
grids = rt.generate_grids(7, W + 10, world.camera.aa)
r = rt.Renderer(0)
r.set_samples(grids)
buf = torch.zeros(H * W * 3, dtype=torch.float32, device="cuda:0")


def frame_ms():
    for _ in range(3):
        r.render_async(buf.data_ptr(), 0, 0); st = r.collect()
    ts = []
    for _ in range(8):
        r.render_async(buf.data_ptr(), 0, 0); st = r.collect(); ts.append(st.ms_launch)
    return min(ts), st


def timed(f, n=5):
    f()
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    return min(ts) * 1e3


v = world.scene.view
t_up = timed(lambda: r.set_world(world))
ms_sah, st_sah = frame_ms()
t_build = timed(lambda: r.build_world(world))
ms_lbvh, st_lbvh = frame_ms()
print(f"{name}: {v.n_spheres} spheres, {v.n_triangles} triangles, {v.n_lights} lights, frame {W}x{H} at {world.spp} spp")
print(f"  host: sahBVH {t_sah * 1e3:.1f} ms + rt_upload_scene {t_up:.2f} ms   -> frame {ms_sah:.3f} ms (single frame), rays {st_sah.rays_total}")
print(f"  GPU : rt_build_scene {t_build:.2f} ms (upload of the objects included) -> frame {ms_lbvh:.3f} ms (single frame), rays {st_lbvh.rays_total}")
r.close()
