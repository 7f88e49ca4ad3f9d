import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctypes as C, numpy as np
import rt_amd as rt
from oracle import binding as ob
f32 = C.c_float
names = ["glass", "diamond", "greenGlass"]
mats = [rt.material(m) for m in names] + [rt.material("whiteDull"), rt.material("redDull")]
sph = [rt.Sphere((f32 * 3)(-4 + 4 * i, 2.0, 0.0), -1.5, i, i) for i in range(3)]
tris = [rt.make_triangle((-40, 0, -40), (-40, 0, 40), (40, 0, 40), 3), rt.make_triangle((-40, 0, -40), (40, 0, 40), (40, 0, -40), 3),
        rt.make_triangle((-40, 0, -10), (40, 0, -10), (40, 40, -10), 4), rt.make_triangle((-40, 0, -10), (40, 40, -10), (-40, 40, -10), 4)]
for k, t in enumerate(tris): t.object_id = 3 + k
order = [0, 1, 2] + [rt.REF_TRIANGLE | k for k in range(4)]
lights = [rt.Light((f32 * 3)(0, 30, 10), (f32 * 3)(2, 0, 0), (f32 * 3)(0, 0, 2), (f32 * 3)(0.8, 0.8, 0.8))]
sc = rt.Scene.create(sph, tris, order, mats, lights, bvh="none")
cfg = rt.Config(image_width=96, image_height=64, view_width=12, view_height=8, view_distance=10,
                reflection_depth=6, anti_aliasing=4, eye=(0.0, 3.0, 14.0), look_at=(0.0, 2.0, 0.0))
w = rt.config_to_world(cfg, sc)
grids = rt.generate_grids(5, 106, 2)
r = rt.Renderer(0); r.set_world(w); r.set_samples(grids)
for gflags, oflags in ((0, 1), (6, 0)):
    g_rgb, g_ids, g_st = r.render(gflags, want_hit_ids=True)
    o_rgb, o_ids, o_st = ob.render(w, grids, oflags)
    g8, o8 = rt.quantize(g_rgb), rt.quantize(o_rgb)
    d = np.abs(g8.astype(int) - o8.astype(int)); d[np.isnan(o_rgb)] = 0
    bad = np.argwhere(d > 1)
    print("flags", gflags, "bad", len(bad))
    for p, c in bad[:12]:
        print(" px", p, "ch", c, "oracle", o_rgb[p], "gpu", g_rgb[p], "ids", o_ids[p])
