"""Known-answer tests that pin the CPU oracle (oracle/rt_oracle.c).

The reference ships no tests and no golden vectors, so these answers are
derived BY HAND from the reference source; each test cites the lines it follows (paths
relative to /root/reference).  They are the pin of the restatement ("parity unpinned"
otherwise: no Haskell toolchain exists in the image to run the reference itself).
"""
import ctypes as C
import math

import numpy as np
import pytest

f32 = C.c_float


def v3(x):
    return (f32 * 3)(*[float(t) for t in x])


def f(x):
    return float(np.float32(x))


def tri(rt, a, b, c):
    return rt.make_triangle(tuple(a), tuple(b), tuple(c), 0)


def hit_triangle(oracle, base, d, t):
    out = f32()
    ok = oracle.lib().rto_hit_triangle(v3(base), v3(d), C.byref(t), C.byref(out))
    return (ok, out.value)


def hit_sphere(oracle, rt, base, d, center, radius):
    s = rt.Sphere((f32 * 3)(*center), radius, 0, 0)
    out = f32()
    ok = oracle.lib().rto_hit_sphere(v3(base), v3(d), C.byref(s), C.byref(out))
    return (ok, out.value)


# --- hit, triangles of the reference's own micro-benchmark (benchmarks/Main.hs:10-19) ---
def test_microbench_triangles(rt, oracle):
    """ray = (10,10,10) -> normalize(-1,-1,-1).  Triangle 0 and 2 face away (theta = dir.n
    >= 0: back-face culled, RayTracer.hs:377), triangle 1 lies in x+y+z = 1 and is hit at
    t = (30 - 1)/sqrt(3)."""
    n = 1 / math.sqrt(3)
    d = (-n, -n, -n)
    base = (10, 10, 10)
    t0 = tri(rt, (0, 1, 0), (1, 0, 0), (0, 0, 1))
    t1 = tri(rt, (1, 0, 0), (0, 1, 0), (0, 0, 1))
    t2 = tri(rt, (4, 0, 0), (4, 0, 1), (4, 1, 0))
    assert hit_triangle(oracle, base, d, t0)[0] == 0
    ok, t = hit_triangle(oracle, base, d, t1)
    assert ok == 1 and t == pytest.approx(29 / math.sqrt(3), rel=2e-6)
    assert hit_triangle(oracle, base, d, t2)[0] == 0


def test_calc_normal(rt):
    """calcNormal a b c = normalize ((b-a) x (c-a)) (Objects.hs:13-15)."""
    t = tri(rt, (1, 0, 0), (0, 1, 0), (0, 0, 1))
    n = 1 / math.sqrt(3)
    assert np.allclose(list(t.n), [n, n, n], rtol=1e-6)
    t = tri(rt, (0, 0, 0), (0, 0, 1), (1, 0, 0))  # (0,0,1) x (1,0,0) = (0,1,0)
    assert list(t.n) == [0.0, 1.0, 0.0]


def test_triangle_edges_and_epsilon(rt, oracle):
    """beta/gamma limits and the t < epsilon cut (RayTracer.hs:378-380)."""
    t = tri(rt, (0, 0, 0), (0, 0, 1), (1, 0, 0))  # in y = 0, normal +y
    down = (0, -1, 0)
    ok, tt = hit_triangle(oracle, (0.25, 2, 0.25), down, t)
    assert ok == 1 and tt == 2.0
    assert hit_triangle(oracle, (0.75, 2, 0.75), down, t)[0] == 0      # beta + gamma > 1
    assert hit_triangle(oracle, (-0.1, 2, 0.25), down, t)[0] == 0      # outside
    assert hit_triangle(oracle, (0.25, -2, 0.25), down, t)[0] == 0     # behind: t = -2 < epsilon
    assert hit_triangle(oracle, (0.25, -2, 0.25), (0, 1, 0), t)[0] == 0  # from below: back face
    # vertices and edges are inside (the comparisons are strict: gamma < 0, beta < 0, > 1)
    assert hit_triangle(oracle, (0, 2, 0), down, t)[0] == 1
    assert hit_triangle(oracle, (0.5, 2, 0.5), down, t)[0] == 1


# --- hit, sphere (RayTracer.hs:339-353) ---
def test_sphere_front_hit_and_never_far_root(rt, oracle):
    ok, t = hit_sphere(oracle, rt, (0, 0, 5), (0, 0, -1), (0, 0, 0), 1.0)
    assert ok == 1 and t == 4.0
    # origin inside: t = min t_a t_b < 0 -> Nothing; the far root is never returned (:350-352)
    assert hit_sphere(oracle, rt, (0, 0, 0), (0, 0, -1), (0, 0, 0), 1.0)[0] == 0
    # tangent line: discriminant == 0 is a miss (`discriminant > 0`, :352)
    assert hit_sphere(oracle, rt, (1, 0, 5), (0, 0, -1), (0, 0, 0), 1.0)[0] == 0
    # sphere behind the origin
    assert hit_sphere(oracle, rt, (0, 0, 5), (0, 0, 1), (0, 0, 0), 1.0)[0] == 0


def test_sphere_non_unit_direction(rt, oracle):
    """t is in units of |dir| (the formula divides by dir.dir, :349-350)."""
    ok, t = hit_sphere(oracle, rt, (0, 0, 5), (0, 0, -2), (0, 0, 0), 1.0)
    assert ok == 1 and t == 2.0


# --- hitsBox (BoundingVolumeHierarchy.hs:191-204): a LINE test ---
def box(rt, b):
    return rt.BvhNode(*[float(x) for x in b], -1, 1)


def hits_box(oracle, rt, base, d, b):
    n = box(rt, b)
    return oracle.lib().rto_hits_box(v3(base), v3(d), C.byref(n))


def test_hits_box_is_a_line_test(rt, oracle):
    unit = (-1, 1, -1, 1, -1, 1)
    assert hits_box(oracle, rt, (0.2, 0.3, 5), (0.01, 0.02, -1), unit) == 1
    # box BEHIND the origin still "hits": there is no t > 0 test in the reference
    assert hits_box(oracle, rt, (0.2, 0.3, 5), (0.01, 0.02, 1), unit) == 1
    assert hits_box(oracle, rt, (3, 3, 5), (0.01, 0.02, -1), unit) == 0
    # negative direction components swap the slab ends (:194-199)
    assert hits_box(oracle, rt, (5, 5, 5), (-1, -1, -1), unit) == 1
    assert hits_box(oracle, rt, (5, 5, 5), (-1, -1, -0.1), unit) == 0


def test_hits_box_nan_counts_as_hit(rt, oracle):
    """0/0 = NaN makes every comparison false, so `not (or ...)` is True (:200-204)."""
    b = (0, 1, 0, 1, 0, 1)
    # origin on the slab plane x = 0 with dx = 0: (0 - 0)/0 = NaN
    assert hits_box(oracle, rt, (0, 0.5, 5), (0, 0, -1), b) == 1


# --- reflect / refract / orthonormal (RayTracer.hs:162-179, 453-459) ---
def test_reflect(oracle):
    out = v3((0, 0, 0))
    oracle.lib().rto_reflect(v3((1, -1, 0)), v3((0, 1, 0)), out)
    assert list(out) == [1.0, 1.0, 0.0]


def test_refract_straight_and_tir(oracle):
    out = v3((0, 0, 0))
    assert oracle.lib().rto_refract(v3((0, -1, 0)), v3((0, 1, 0)), f32(1.5), out) == 1
    assert list(out) == [0.0, -1.0, 0.0]
    # 45 degrees into glass: sin(t) = sin(45)/1.5
    s = math.sqrt(0.5)
    assert oracle.lib().rto_refract(v3((s, -s, 0)), v3((0, 1, 0)), f32(1.5), out) == 1
    assert out[0] == pytest.approx(s / 1.5, rel=1e-6)
    assert out[1] == pytest.approx(-math.sqrt(1 - (s / 1.5) ** 2), rel=1e-6)
    # leaving glass (nt = 1/1.5) at 60 degrees: x < 0 -> Nothing (total internal reflection)
    d = (math.sin(math.radians(60)), -math.cos(math.radians(60)), 0)
    assert oracle.lib().rto_refract(v3(d), v3((0, 1, 0)), f32(1 / 1.5), out) == 0


def test_orthonormal(oracle):
    u, v = v3((0, 0, 0)), v3((0, 0, 0))
    oracle.lib().rto_orthonormal(v3((0, 0, 1)), u, v)   # t = (1,0,0): u = t x w = (0,-1,0)
    assert list(u) == [0.0, -1.0, 0.0] and list(v) == [1.0, 0.0, 0.0]
    oracle.lib().rto_orthonormal(v3((1, 0, 0)), u, v)   # the special case: t = (0,1,0)
    assert list(u) == [0.0, 0.0, -1.0] and list(v) == [0.0, 1.0, 0.0]
    oracle.lib().rto_orthonormal(v3((-1, 0, 0)), u, v)  # t x w = 0 -> normalize -> NaN (:457)
    assert all(math.isnan(x) for x in u)


# --- getRay (RayTracer.hs:183-205) ---
def test_get_ray_pinhole_centre(rt, oracle):
    """lens = 0: base = eye; the centre of the view plane maps to -w."""
    cfg = rt.Config(image_width=8, image_height=6, view_width=8, view_height=6, view_distance=7,
                    eye=(10.0, 0.0, 0.0), look_at=(0.0, 0.0, 0.0))
    cam = rt.config_to_world(cfg, rt.Scene.builtin("kat_tris")).camera
    assert list(cam.w) == [1.0, 0.0, 0.0]                     # getCam: w = normalize (eye - lookAt)
    assert list(cam.u) == [0.0, 0.0, -1.0]                    # u = normalize (up x w)
    assert list(cam.v) == [0.0, 1.0, 0.0]                     # v = w x u
    base, d = v3((0, 0, 0)), v3((0, 0, 0))
    oracle.lib().rto_get_ray(C.byref(cam), f32(4.0), f32(3.0), f32(0.3), f32(0.9), base, d)
    assert list(base) == [10.0, 0.0, 0.0]
    assert list(d) == [-1.0, 0.0, 0.0]
    # the corner (0,0): u_world = -4, v_world = -3, dir = w*(-7) + (u*(-4) + v*(-3))
    oracle.lib().rto_get_ray(C.byref(cam), f32(0.0), f32(0.0), f32(0.0), f32(0.0), base, d)
    exp = np.array([-7.0, -3.0, 4.0])
    assert np.allclose(list(d), exp / np.linalg.norm(exp), rtol=1e-6)


def test_get_ray_lens_offsets_base_not_focal_point(rt, oracle):
    """Thin lens: base moves in the (u,v) plane by (r - 1/2)*lens, the ray still passes
    through the same view-plane point (:188-190, 203)."""
    cfg = rt.Config(image_width=8, image_height=6, view_width=8, view_height=6, view_distance=7,
                    lens=0.5, eye=(10.0, 0.0, 0.0), look_at=(0.0, 0.0, 0.0))
    cam = rt.config_to_world(cfg, rt.Scene.builtin("kat_tris")).camera
    base, d = v3((0, 0, 0)), v3((0, 0, 0))
    oracle.lib().rto_get_ray(C.byref(cam), f32(4.0), f32(3.0), f32(1.0), f32(0.0), base, d)
    # ru = u*(1*0.5) - u*0.25 = 0.25 u = (0,0,-0.25);  rv = v*0 - v*0.25 = (0,-0.25,0)
    assert np.allclose(list(base), [10.0, -0.25, -0.25])
    focal = np.array([3.0, 0.0, 0.0])                         # eye + (-7) w
    exp = focal - np.array(list(base))
    assert np.allclose(list(d), exp / np.linalg.norm(exp), rtol=1e-6)


# --- hitBVH tie rules (RayTracer.hs:383-412) ---
def _two_coincident(rt, first_material, second_material, bvh):
    """Two identical triangles (so both are hit at the same t) with different materials."""
    mats = [rt.material("redDull"), rt.material("blueDull")]
    a, b, c = (0, 0, 0), (0, 0, 1), (1, 0, 0)
    tris = [rt.make_triangle(a, b, c, first_material), rt.make_triangle(a, b, c, second_material)]
    order = [rt.REF_TRIANGLE | 0, rt.REF_TRIANGLE | 1]
    return rt.Scene.create([], tris, order, mats, [], bvh=bvh)


def test_leaf_tie_first_object_wins(rt, oracle):
    """hits keeps the earlier object on equal t (strict `<`, RayTracer.hs:396-401)."""
    sc = _two_coincident(rt, 0, 1, "none")
    t, oid = f32(), C.c_uint32()
    ok = oracle.lib().rto_hit_bvh(C.byref(sc.view), v3((0.25, 2, 0.25)), v3((0, -1, 0)),
                                  C.byref(t), C.byref(oid))
    assert ok == 1 and t.value == 2.0 and oid.value == 0


def test_node_tie_right_subtree_wins(rt, oracle):
    """Node: max (hitBVH l) (hitBVH r) with compare flipped on t: on equal t `max x y`
    returns y, the RIGHT subtree's record (RayTracer.hs:383-387, 410)."""
    sc = _two_coincident(rt, 0, 1, "none")
    # hand-built tree: root -> (leaf {object 0}, leaf {object 1}), same boxes
    nodes = (rt.BvhNode * 3)(rt.BvhNode(0, 1, 0, 0, 0, 1, 1, 2), rt.BvhNode(0, 1, 0, 0, 0, 1, -1, 1),
                             rt.BvhNode(0, 1, 0, 0, 0, 1, -2, 1))
    refs = (C.c_uint32 * 2)(rt.REF_TRIANGLE | 0, rt.REF_TRIANGLE | 1)
    v = sc.view
    view = rt.SceneView(nodes, 3, refs, 2, v.spheres, v.n_spheres, v.triangles, v.n_triangles,
                        v.materials, v.n_materials, v.lights, v.n_lights)
    t, oid = f32(), C.c_uint32()
    ok = oracle.lib().rto_hit_bvh(C.byref(view), v3((0.25, 2, 0.25)), v3((0, -1, 0)), C.byref(t),
                                  C.byref(oid))
    assert ok == 1 and t.value == 2.0 and oid.value == 1


# --- raytrace / getDirectColor (RayTracer.hs:55-106) on a one-triangle world ---
def test_direct_colour_lambert_phong_by_hand(rt, oracle):
    """Floor triangle (normal +y, whiteDull: diffuse = specular = 1, phong 1, no reflection),
    one point light straight above the hit point, viewed straight down:
      l_dir = n = v = h = (0,1,0): s = 1, (n.h)**1 = 1
      colour = diffuse*l_c*1 + l_c*spec*1 + ambient*diffuse = 0.5 + 0.5 + 0.1 = 1.1
    and the always-traced reflection ray (straight up) hits nothing -> + 0."""
    mats = [rt.material("whiteDull")]
    tris = [rt.make_triangle((-5, 0, -5), (-5, 0, 5), (5, 0, 5), 0)]
    lights = [rt.Light((f32 * 3)(1, 4, 2), (f32 * 3)(0, 0, 0), (f32 * 3)(0, 0, 0),
                       (f32 * 3)(0.5, 0.5, 0.5))]
    sc = rt.Scene.create([], tris, [rt.REF_TRIANGLE | 0], mats, lights)
    cam = rt.config_to_world(rt.Config(), sc).camera        # ambient 0.1 (Main.hs:117)
    rgb = v3((0, 0, 0))
    oracle.lib().rto_raytrace(C.byref(sc.view), C.byref(cam), 3, f32(0.5), f32(0.5), v3((1, 3, 2)),
                              v3((0, -1, 0)), 0, rgb)
    assert np.allclose(list(rgb), [1.1, 1.1, 1.1], rtol=1e-6)
    # depth 0 -> black (RayTracer.hs:56)
    oracle.lib().rto_raytrace(C.byref(sc.view), C.byref(cam), 0, f32(0.5), f32(0.5), v3((1, 3, 2)),
                              v3((0, -1, 0)), 0, rgb)
    assert list(rgb) == [0.0, 0.0, 0.0]


def test_shadowed_point_gets_ambient_only(rt, oracle):
    """A second triangle between the point and the light: inShadow (t < |l|, :231-235)
    leaves ambient*diffuse = 0.1."""
    mats = [rt.material("whiteDull")]
    tris = [rt.make_triangle((-5, 0, -5), (-5, 0, 5), (5, 0, 5), 0),
            # occluder at y = 2 facing DOWN (a shadow ray going up must see its front face)
            rt.make_triangle((-5, 2, -5), (5, 2, 5), (-5, 2, 5), 0)]
    assert list(tris[1].n) == [0.0, -1.0, 0.0]
    lights = [rt.Light((f32 * 3)(-2, 4, 2), (f32 * 3)(0, 0, 0), (f32 * 3)(0, 0, 0),
                       (f32 * 3)(0.5, 0.5, 0.5))]
    sc = rt.Scene.create([], tris, [rt.REF_TRIANGLE | 0, rt.REF_TRIANGLE | 1], mats, lights)
    cam = rt.config_to_world(rt.Config(), sc).camera
    rgb = v3((0, 0, 0))
    # come in from the side, below the occluder
    d = np.array([-1.0, -1.0, 0.0]) / math.sqrt(2)
    oracle.lib().rto_raytrace(C.byref(sc.view), C.byref(cam), 1, f32(0.5), f32(0.5),
                              v3((-1, 1, 2)), v3(d), 0, rgb)
    assert np.allclose(list(rgb), [0.1, 0.1, 0.1], rtol=1e-6)


def test_default_ray_policy_never_changes_a_finite_pixel(rt, oracle):
    """The GPU's default policy, mirrored by the oracle's FLAG_CULL_ZERO_WEIGHT: reflection rays
    of weight (0,0,0) and inShadow rays of lights whose diffuse and specular factors are both
    zero (max 0 (l.n) = 0 and max 0 (n.h) = 0, RayTracer.hs:101-105) are not traced.  They can
    only contribute 0, so every finite pixel and every primary hit id is bit-identical to the
    reference's full accounting; only the ray counts drop."""
    from conftest import make_world
    for name, rpp in (("bench1", 4), ("bench2", 1), ("mesh:3000", 1), ("bench5", 4)):
        w = make_world(rt, name, 72, 40, rpp)
        grids = rt.generate_grids(31, w.width + 10, w.camera.aa)
        a_rgb, a_ids, a_st = oracle.render(w, grids, 0)
        b_rgb, b_ids, b_st = oracle.render(w, grids, oracle.FLAG_CULL_ZERO_WEIGHT)
        assert np.array_equal(a_ids, b_ids)
        fin = np.isfinite(a_rgb) & np.isfinite(b_rgb)
        assert np.array_equal(a_rgb[fin], b_rgb[fin])
        assert b_st.rays_primary == a_st.rays_primary
        assert b_st.rays_shadow <= a_st.rays_shadow and b_st.rays_reflection <= a_st.rays_reflection
        assert b_st.rays_total < a_st.rays_total


# --- getIndirectColor (RayTracer.hs:109-156): Schlick, TIR, Beer's law, both children ---
@pytest.mark.parametrize("case", ["schlick_head_on_sphere", "schlick_60_degrees_pane",
                                  "total_internal_reflection", "leaving_the_medium_beer"])
def test_indirect_colour_by_hand(rt, oracle, case):
    """tests/kat_scenes.py derives each colour by hand from the reference source; the oracle must
    reproduce it (whole-frame render of the 1x1 image whose only ray is the KAT ray)."""
    import kat_scenes
    sc, cfg, expected = kat_scenes.CASES[case](rt)
    w = rt.config_to_world(cfg, sc)
    rgb, ids, st = oracle.render(w, kat_scenes.CENTRE_SAMPLE, 0)
    assert np.allclose(rgb[0], expected, rtol=2e-5, atol=1e-7), (rgb[0], expected)
    assert ids[0, 0] == 0                                    # the refractive surface is object 0
    if case == "total_internal_reflection":
        assert st.rays_refraction == 0 and st.tir == 0       # TIR on ENTERING (nt < 1): not the exit branch
    else:
        assert st.rays_refraction == 1
    # (the walls' own zero-weight reflection rays come back to the sphere, so the branch count is
    #  3 with the inside-out sphere: the KAT ray's hit and two grandchildren)
    assert (st.exit_branch >= 1) == (case == "leaving_the_medium_beer")
