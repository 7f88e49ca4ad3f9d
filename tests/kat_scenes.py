"""Hand-derived known answers for getIndirectColor (src/RayTracer.hs:109-156 of the reference):
one ray, one refractive surface, a red wall for the reflected ray and a blue floor for the
refracted one.  No lights, so every surface shows ambient * diffuse (getDirectColor [] , :91;
ambient = 0.1, Main.hs:117):  redDull -> (0.1, 0.01, 0.01),  blueDull -> (0.01, 0.01, 0.1)
(Materials.hs:10-15, 34-43).  The light-sample pair is (0.5, 0.5), so the glossy jitter
ju = -(a/2) + 0.5*a is exactly 0 (:125-127) and the reflected ray is `reflect dir n` itself.
The walls' own (always traced, :129-131) reflection rays are multiplied by reflectionIndex 0.

Each case: (scene, Config for a 1x1 image whose only ray is the KAT ray, expected colour, note).
Used by tests/test_oracle_kat.py (oracle, CPU) and tests/test_gpu_parity.py (1x1 GPU render)."""
import ctypes as C
import math

import numpy as np

f32 = C.c_float
RED = np.array([0.1, 0.01, 0.01])     # ambient * diffuse of redDull
BLUE = np.array([0.01, 0.01, 0.1])    # ... of blueDull


def _custom(rt, diffuse, specular, phong, refl, refr_ix, atten):
    """makeMaterial (Surfaces.hs:22-24): the attenuation colour is stored as its log."""
    la = [float(np.log(np.float32(a))) for a in atten]
    return rt.Material((f32 * 3)(*diffuse), (f32 * 3)(*specular), phong, refl, refr_ix, (f32 * 3)(*la))


def _ids(tris, first):
    for k, t in enumerate(tris):
        t.object_id = first + k
    return tris


def head_on_glass_sphere(rt, radius=1.0, atten=None):
    """Ray (0,0,5) -> (0,0,-1) onto a sphere |r| = 1 at the origin, nt = 1.5.
    radius > 0 (`glass`, Materials.hs:81-82: black, white, 10000, 1, 1.5, white):
      t = 4, pt = (0,0,1), n = (0,0,1), dn = -1 < 0: c = 1, k = (1,1,1), refract gives (0,0,-1)
      r_0 = (0.5)^2 / (2.5)^2 = 0.04, R = 0.04 + 0.96 * (1-1)^5 = 0.04
      reflected ray (0,0,1) -> red wall at z = 6; refl_c = 1 * (white * RED) = RED
      refracted ray from (0,0,1) down: the sphere itself is missed (t_b = (1-1)/1 = 0, not > 1e-8,
      and the far root is never reported, :350-352) -> blue floor at z = -3: BLUE
      colour = 0 (direct: black diffuse) + k * (R * RED + (1-R) * BLUE)
             = (0.004 + 0.0096, 0.0004 + 0.0096, 0.0004 + 0.096) = (0.0136, 0.01, 0.0964)
    radius < 0 (the same sphere inside out: getNormal = (pt - c)/r = (0,0,-1), :76):
      dn = +1 >= 0 -> the leaving-the-medium branch (:140-145): t = refract dir (-n) (1/nt) = (0,0,-1),
      c = t.n = 1 -> R = 0.04; k = exp(-log atten) = 1/atten per channel (Beer's law, path-length free);
      reflect dir n = (0,0,-1) - (0,0,-1)*2 = (0,0,1): the same two children
      colour = (1/atten) * (0.0136, 0.01, 0.0964); atten = (0.5, 0.25, 1) -> (0.0272, 0.04, 0.0964)"""
    if atten is None:
        glass = rt.material("glass")
        k = np.ones(3)
    else:
        glass = _custom(rt, (0, 0, 0), (1, 1, 1), 10000.0, 1.0, 1.5, atten)
        k = 1.0 / np.array(atten)
    mats = [glass, rt.material("redDull"), rt.material("blueDull")]
    sph = [rt.Sphere((f32 * 3)(0, 0, 0), radius, 0, 0)]
    tris = _ids([rt.make_triangle((-9, -9, 6), (-9, 18, 6), (18, -9, 6), 1),      # faces -z
                 rt.make_triangle((-9, -9, -3), (18, -9, -3), (-9, 18, -3), 2)], 1)  # faces +z
    assert list(tris[0].n) == [0.0, 0.0, -1.0] and list(tris[1].n) == [0.0, 0.0, 1.0]
    # noBVH: one leaf spanned by the two triangles (a negative radius turns the sphere's toBox inside out)
    sc = rt.Scene.create(sph, tris, [0, rt.REF_TRIANGLE | 0, rt.REF_TRIANGLE | 1], mats, [], bvh="none")
    cfg = rt.Config(image_width=1, image_height=1, view_width=1, view_height=1, view_distance=5,
                    reflection_depth=3, anti_aliasing=1, eye=(0.0, 0.0, 5.0), look_at=(0.0, 0.0, 0.0))
    expected = k * (0.04 * RED + 0.96 * BLUE)
    return sc, cfg, expected


def oblique_glass_pane(rt):
    """Ray from (2*sqrt 3, 2, 0) towards the origin: 60 degrees to the normal (0,1,0) of a `glass`
    pane in y = 0 (Materials.hs:81-82, nt = 1.5).
      dn = -1/2: c = 1/2, r_0 = 0.04, R = 0.04 + 0.96 * (1/2)^5 = 0.07
      reflected ray (-sqrt3/2, +1/2, 0) -> red ceiling y = 3 (faces down): RED
      refract: sin = (sqrt3/2)/1.5 -> refracted ray goes on downwards (the pane itself: t = 0 < 1e-8)
      -> blue floor y = -3: BLUE
      colour = 0.07 * RED + 0.93 * BLUE = (0.0163, 0.01, 0.0937)"""
    mats = [rt.material("glass"), rt.material("redDull"), rt.material("blueDull")]
    tris = _ids([rt.make_triangle((-9, 0, -9), (-9, 0, 18), (18, 0, -9), 0),       # pane, faces +y
                 rt.make_triangle((-30, 3, -30), (30, 3, -30), (-30, 3, 60), 1),   # ceiling, faces -y
                 rt.make_triangle((-30, -3, -30), (-30, -3, 60), (30, -3, -30), 2)], 0)  # floor, faces +y
    assert list(tris[0].n) == [0.0, 1.0, 0.0] and list(tris[1].n) == [0.0, -1.0, 0.0] and list(tris[2].n) == [0.0, 1.0, 0.0]
    sc = rt.Scene.create([], tris, [rt.REF_TRIANGLE | k for k in range(3)], mats, [], bvh="none")
    cfg = rt.Config(image_width=1, image_height=1, view_width=1, view_height=1, view_distance=4,
                    reflection_depth=3, anti_aliasing=1, eye=(2 * math.sqrt(3.0), 2.0, 0.0), look_at=(0.0, 0.0, 0.0))
    return sc, cfg, 0.07 * RED + 0.93 * BLUE


def total_internal_reflection(rt):
    """The same ray onto a pane with nt = 1/2, diffuse 0.2, specular (1, 0.5, 0.25), reflectionIndex 0.8:
      refract: x = 1 - (1 - 1/4) / (1/2) / (1/2) = -2 < 0 -> Nothing (:170-172); dn < 0 -> k = (1,1,1)
      colour = ambient * diffuse + k * refl_c = 0.02 + 0.8 * ((1, 0.5, 0.25) * RED)
             = (0.02 + 0.08, 0.02 + 0.004, 0.02 + 0.002) = (0.1, 0.024, 0.022)          (:151)
    No refraction ray is traced."""
    pane = _custom(rt, (0.2, 0.2, 0.2), (1.0, 0.5, 0.25), 10000.0, 0.8, 0.5, (1.0, 1.0, 1.0))
    mats = [pane, rt.material("redDull"), rt.material("blueDull")]
    tris = _ids([rt.make_triangle((-9, 0, -9), (-9, 0, 18), (18, 0, -9), 0),
                 rt.make_triangle((-30, 3, -30), (30, 3, -30), (-30, 3, 60), 1),
                 rt.make_triangle((-30, -3, -30), (-30, -3, 60), (30, -3, -30), 2)], 0)
    sc = rt.Scene.create([], tris, [rt.REF_TRIANGLE | k for k in range(3)], mats, [], bvh="none")
    cfg = rt.Config(image_width=1, image_height=1, view_width=1, view_height=1, view_distance=4,
                    reflection_depth=3, anti_aliasing=1, eye=(2 * math.sqrt(3.0), 2.0, 0.0), look_at=(0.0, 0.0, 0.0))
    return sc, cfg, np.array([0.1, 0.024, 0.022])


CASES = {
    "schlick_head_on_sphere": lambda rt: head_on_glass_sphere(rt),
    "schlick_60_degrees_pane": oblique_glass_pane,
    "total_internal_reflection": total_internal_reflection,
    "leaving_the_medium_beer": lambda rt: head_on_glass_sphere(rt, radius=-1.0, atten=(0.5, 0.25, 1.0)),
}

CENTRE_SAMPLE = np.array([[[0.5, 0.5, 0.5, 0.5, 0.5, 0.5]]] * 11, np.float32)   # [W+10][1][F6]: pixel centre, (s0,s1) = (0.5,0.5)
