"""Animation driver (BASELINE config 5): scene description with object
coordinates as expressions of the frame number t.  The reference's parser for this is dead
code, so these tests pin OUR reading of the documented dialect
(SceneParser.hs:60-73, 196-209, 291-395; orbits as in the sample `sdl`:46-52)."""
import importlib
import math

import numpy as np
import pytest

SDL = """
CONFIG
  EYE {15 2 15}
  LOOKAT {-1 -1 -1}
  UP {0 1 0}
  VIEWPLANE {8 6 4}
  AMBIENT Color 0.1 0.1 0.1
END

LIGHTS
  Light  {50 20 0} bright
  Light  {3 2 20} Color 0.2 0.2 0.2
END

COLORS
  Color bright = 0.5 0.5 0.5
  Color  purple = 0.5 0.2 0.5
  Color  white = 1.0 1.0 1.0
  Color  black = 0.0 0.0 0.0
  Color  grey = 0.6 0.6 0.6
  Color  red = 1.0 0.0 0.0
  Color  glass = 0.1 0.2 0.4
END

MATERIAL
  Material  m_glass = glass glass grey 10.0 white 1.2 white 0.0
  Material  m_red   = red red grey 1000.0 grey 0.0 black 0.0
  Material  m_dull  = purple purple white 100.0 black 0.0 black 0.0
END

SHAPES
  Sphere s1 = {0 0 0} 3 m_glass
  Sphere s2 = {sin(t*3.14/50)*7 0 cos(t*3.14/50)*7} 1 m_red
  Sphere s3 = {0 sin(t*3.14/50)*8 cos(t*3.14/50)*8} 1+t/100 m_dull
  Triangle t1 = {-20 -4 -20} {-20 -4 20} {20 -4 20} m_dull
END
"""


@pytest.fixture(scope="module")
def anim():
    import rt_amd  # noqa: F401  (makes the package importable under its hyphenated name)
    return importlib.import_module("ray-tracer_amd.animation")


def test_expression_language(anim):
    E = anim.Expr
    assert E("1+2*3")(0) == 7.0
    assert E("(1+2)*3")(0) == 9.0
    assert E("8/2/2")(0) == 2.0                     # left associative
    assert E("-3+5")(0) == 2.0
    assert E("t*2")(21) == 42.0
    f32 = np.float32
    assert E("sin(t*3.14/50)*7")(25) == pytest.approx(7 * math.sin(float(f32(25) * f32(3.14) / f32(50))), rel=1e-6)
    assert E("cos(0)")(3) == 1.0
    with pytest.raises(Exception):
        E("1+")
    with pytest.raises(Exception):
        E("(1+2")


def test_scene_description_and_motion(rt, anim):
    d = anim.parse_scene_description(SDL)
    assert d.eye == (15.0, 2.0, 15.0) and d.view == (8, 6, 4)
    assert len(d.lights) == 2 and list(d.lights[1].color) == pytest.approx([0.2, 0.2, 0.2])
    assert list(d.lights[0].edge_a) == [0, 0, 0]              # point lights
    g = d.materials[d.material_index["m_glass"]]
    assert g.refr_ix == pytest.approx(1.2) and g.refl == pytest.approx(1.0) and g.phong == 10.0
    assert list(g.log_atten) == [0.0, 0.0, 0.0]               # log of white
    r = d.materials[d.material_index["m_red"]]
    assert r.refl == pytest.approx(0.6) and list(r.log_atten) == [-math.inf] * 3
    s0, s25 = d.scene_at(0), d.scene_at(25)
    assert s0.n_objects == 4 and s0.view.n_spheres == 3 and s0.view.n_triangles == 1
    assert list(s0.view.spheres[1].center) == [0.0, 0.0, 7.0]
    c = list(s25.view.spheres[1].center)
    assert c[0] == pytest.approx(7.0, abs=1e-3) and abs(c[2]) < 0.01      # quarter orbit
    assert s25.view.spheres[2].radius == pytest.approx(1.25)
    assert len(s25.nodes()) >= 1
    w = d.world_at(3, rt.Config(image_width=64, image_height=48, anti_aliasing=4, reflection_depth=4))
    assert w.camera.view_d == 4.0 and w.camera.aa == 2 and list(w.camera.eye) == [15.0, 2.0, 15.0]


def test_frames_differ_in_the_oracle(rt, oracle, anim):
    d = anim.parse_scene_description(SDL)
    cfg = rt.Config(image_width=48, image_height=36, anti_aliasing=1, reflection_depth=3)
    grids = rt.generate_grids(1, 58, 1)
    a, _, _ = oracle.render(d.world_at(0, cfg), grids)
    b, _, _ = oracle.render(d.world_at(20, cfg), grids)
    assert np.isfinite(a).all() and not np.array_equal(a, b)


@pytest.mark.gpu
def test_sequence_matches_oracle_frame_by_frame(rt, oracle, anim):
    d = anim.parse_scene_description(SDL)
    cfg = rt.Config(image_width=96, image_height=72, anti_aliasing=4, reflection_depth=5)
    grids = rt.generate_grids(5, 106, 2)
    r = rt.Renderer(0)
    try:
        n = 0
        for t, rgb, st in anim.render_sequence(r, d, cfg, grids, [0, 10, 33]):
            o_rgb, _, o_st = oracle.render(d.world_at(t, cfg), grids, oracle.FLAG_CULL_ZERO_WEIGHT,
                                           want_hit_ids=False)
            assert st.rays_total == o_st.rays_total
            diff = np.abs(rt.quantize(rgb).astype(int) - rt.quantize(o_rgb).astype(int))
            assert diff.max() <= 1
            n += 1
        assert n == 3
    finally:
        r.close()


def _refitted_view(rt, base_scene, moved_scene):
    """The tree of `base_scene` with the boxes sahBVH would attach to it for `moved_scene`'s objects
    (leaf = fold of toBox, node = union of children: BoundingVolumeHierarchy.hs:50-51), on the host in
    numpy -- what rt_refit_async must produce on the GPU."""
    import ctypes as C
    bv, mv = base_scene.view, moved_scene.view
    nodes = [rt.BvhNode(*[bv.nodes[i].xmin, bv.nodes[i].xmax, bv.nodes[i].ymin, bv.nodes[i].ymax, bv.nodes[i].zmin,
                          bv.nodes[i].zmax, bv.nodes[i].left, bv.nodes[i].right]) for i in range(bv.n_nodes)]

    def obj_box(ref):
        if ref & rt.REF_TRIANGLE:
            t = mv.triangles[ref & 0x7fffffff]
            pts = np.array([list(t.a), list(t.b), list(t.c)], np.float32)
            lo, hi = pts.min(axis=0), pts.max(axis=0)
        else:
            sp = mv.spheres[ref]
            c, r = np.array(list(sp.center), np.float32), np.float32(sp.radius)
            lo, hi = c - r, c + r
        return np.array([lo[0], hi[0], lo[1], hi[1], lo[2], hi[2]], np.float32)

    def union(a, b):
        return np.array([min(a[0], b[0]), max(a[1], b[1]), min(a[2], b[2]), max(a[3], b[3]), min(a[4], b[4]),
                         max(a[5], b[5])], np.float32)

    def fit(i):
        n = nodes[i]
        if n.left < 0:
            first = -1 - n.left
            b = obj_box(bv.leaf_refs[first])
            for k in range(1, n.right):
                b = union(b, obj_box(bv.leaf_refs[first + k]))
        else:
            b = union(fit(n.left), fit(n.right))
        n.xmin, n.xmax, n.ymin, n.ymax, n.zmin, n.zmax = [float(x) for x in b]
        return b
    fit(0)
    arr = (rt.BvhNode * len(nodes))(*nodes)
    view = rt.SceneView(arr, len(nodes), bv.leaf_refs, bv.n_refs, mv.spheres, mv.n_spheres, mv.triangles,
                        mv.n_triangles, mv.materials, mv.n_materials, mv.lights, mv.n_lights)
    view._keep = (arr, base_scene, moved_scene)
    return view


@pytest.mark.gpu
def test_gpu_refit_equals_the_refitted_tree_and_the_rebuilt_one(rt, oracle, anim):
    """rt_refit_async: upload frame 0, then refit to later frames on the GPU.
    (i) The refit frame is BITWISE the frame of an upload of the same topology with boxes refitted
    on the host, and equals the oracle on that very tree (ids, ray counts, 8-bit <= 1): the device
    refit computes exactly sahBVH's boxes for the topology.  (ii) Against the oracle on a tree
    REBUILT by sahBVH for the moved objects (what the reference's host would do every frame) ids
    and ray counts are still identical on this scene (no coincident surfaces, so no tie depends on
    the leaf order) and the 8-bit frame is within 1."""
    from types import SimpleNamespace
    d = anim.parse_scene_description(anim.ORBIT_DESCRIPTION)
    cfg = rt.Config(image_width=160, image_height=96, anti_aliasing=4, reflection_depth=6)
    grids = rt.generate_grids(5, 170, 2)
    w0 = d.world_at(0.0, cfg)
    r, r2 = rt.Renderer(0), rt.Renderer(0)
    try:
        r.set_world(w0)
        r.set_samples(grids)
        r2.set_samples(grids)
        base, _, _ = r.render(0)
        r.refit(w0.scene)                                    # nothing moved: the very same frame
        again, _, _ = r.render(0)
        assert np.array_equal(base.view(np.uint32), again.view(np.uint32))
        for t in (7.0, 23.0, 41.0):
            wt = d.world_at(t, cfg)                          # rebuilt by sahBVH on the host
            r.refit(wt.scene)
            g_rgb, g_ids, g_st = r.render(0, want_hit_ids=True)
            assert not np.array_equal(g_rgb, base)
            view = _refitted_view(rt, w0.scene, wt.scene)
            wr = SimpleNamespace(scene=SimpleNamespace(view=view), camera=wt.camera, width=160, height=96, spp=4)
            r2.set_world(wr)
            h_rgb, h_ids, h_st = r2.render(0, want_hit_ids=True)
            assert np.array_equal(g_rgb.view(np.uint32), h_rgb.view(np.uint32)) and np.array_equal(g_ids, h_ids)
            assert g_st.rays_total == h_st.rays_total
            for world in (wr, wt):
                o_rgb, o_ids, o_st = oracle.render(world, grids, oracle.FLAG_CULL_ZERO_WEIGHT)
                assert int((g_ids != o_ids).sum()) == 0 and g_st.rays_total == o_st.rays_total
                assert np.abs(rt.quantize(g_rgb).astype(int) - rt.quantize(o_rgb).astype(int)).max() <= 1
        # the sequence driver: rebuild every third frame, refit in between
        n = 0
        for t, rgb, st in anim.render_sequence(r, d, cfg, grids, [0, 5, 10, 15, 20], rebuild_every=3):
            o_rgb, _, o_st = oracle.render(d.world_at(t, cfg), grids, oracle.FLAG_CULL_ZERO_WEIGHT, want_hit_ids=False)
            assert st.rays_total == o_st.rays_total
            assert np.abs(rt.quantize(rgb).astype(int) - rt.quantize(o_rgb).astype(int)).max() <= 1
            n += 1
        assert n == 5
        with pytest.raises(rt.RtError, match="the uploaded scene has"):
            r.refit(rt.Scene.builtin("bench1"))
    finally:
        r.close()
        r2.close()
