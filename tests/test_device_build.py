"""The tree built on the GPU (rt_build_scene): a TREE-CHANGING mode -- a linear BVH,
not the reference's sahBVH -- so the parity statement is: the GPU frame equals what the reference's
algorithm (the oracle) returns when it walks THE SAME tree, which rt_download_tree hands back in
rt_upload_scene's input format.  Hit ids and ray counts identical, frame to the parity bar, in the
pruned default mode and in the reference-visit mode; and the downloaded tree, uploaded through the
host path, gives bitwise the frame of the device-built context."""
from types import SimpleNamespace

import numpy as np
import pytest

from conftest import make_world
from test_gpu_parity import check_image

pytestmark = pytest.mark.gpu


def tree_world(w, view):
    return SimpleNamespace(scene=SimpleNamespace(view=view), camera=w.camera, width=w.width, height=w.height, spp=w.spp)


def check_tree(rt, view):
    """Every object in exactly one leaf (one per leaf), children inside their parents, leaf boxes =
    their object's toBox."""
    n = view.n_nodes
    n_obj = view.n_spheres + view.n_triangles
    assert view.n_refs == n_obj and n == (2 * n_obj - 1 if n_obj else 0)
    if n == 0:
        return 0
    seen_nodes = np.zeros(n, bool)
    seen_refs = set()
    deepest = 0
    stack = [(0, 1)]
    while stack:
        i, d = stack.pop()
        assert not seen_nodes[i]
        seen_nodes[i] = True
        deepest = max(deepest, d)
        nd = view.nodes[i]
        box = (nd.xmin, nd.xmax, nd.ymin, nd.ymax, nd.zmin, nd.zmax)
        if nd.left < 0:
            assert nd.right == 1
            r = int(view.leaf_refs[-1 - nd.left])
            assert r not in seen_refs
            seen_refs.add(r)
            if r & rt.REF_TRIANGLE:
                t = view.triangles[r & ~rt.REF_TRIANGLE]
                pts = np.array([list(t.a), list(t.b), list(t.c)], np.float32)
                want = [pts[:, 0].min(), pts[:, 0].max(), pts[:, 1].min(), pts[:, 1].max(), pts[:, 2].min(), pts[:, 2].max()]
            else:
                s = view.spheres[r]
                c, rad = np.array(list(s.center), np.float32), np.float32(s.radius)
                want = [c[0] - rad, c[0] + rad, c[1] - rad, c[1] + rad, c[2] - rad, c[2] + rad]
            assert list(np.array(box, np.float32)) == list(np.array(want, np.float32))
        else:
            for ch in (nd.left, nd.right):
                cb = view.nodes[ch]
                assert cb.xmin >= nd.xmin and cb.xmax <= nd.xmax and cb.ymin >= nd.ymin and cb.ymax <= nd.ymax and \
                       cb.zmin >= nd.zmin and cb.zmax <= nd.zmax
                stack.append((ch, d + 1))
            l, r = view.nodes[nd.left], view.nodes[nd.right]
            assert box == (min(l.xmin, r.xmin), max(l.xmax, r.xmax), min(l.ymin, r.ymin), max(l.ymax, r.ymax),
                           min(l.zmin, r.zmin), max(l.zmax, r.zmax))
    assert seen_nodes.all() and len(seen_refs) == n_obj
    return deepest


@pytest.mark.parametrize("name,w,h,rpp", [("bench1", 160, 120, 4), ("bench5", 96, 72, 4), ("bench3", 128, 96, 4),
                                           ("bench4", 128, 96, 4), ("mesh:3000", 128, 72, 4)])
def test_device_built_tree_renders_what_the_reference_returns_on_it(rt, oracle, name, w, h, rpp):
    world = make_world(rt, name, w, h, rpp)
    grids = rt.generate_grids(17, w + 10, world.camera.aa)
    r = rt.Renderer(0)
    r2 = rt.Renderer(0)
    try:
        r.build_world(world)
        r.set_samples(grids)
        view = r.download_tree()
        depth = check_tree(rt, view)
        assert depth <= 64
        tw = tree_world(world, view)
        for gflags, oflags in ((0, oracle.FLAG_CULL_ZERO_WEIGHT), (rt.FLAG_NO_PRUNE | rt.FLAG_TRACE_ZERO_WEIGHT, 0)):
            g_rgb, g_ids, g_st = r.render(gflags, want_hit_ids=True)
            o_rgb, o_ids, o_st = oracle.render(tw, grids, oflags)
            assert int((g_ids != o_ids).sum()) == 0
            assert (g_st.rays_primary, g_st.rays_shadow, g_st.rays_reflection, g_st.rays_refraction) == \
                   (o_st.rays_primary, o_st.rays_shadow, o_st.rays_reflection, o_st.rays_refraction)
            check_image(rt, g_rgb, o_rgb)
        # the same tree through the host path (rt_upload_scene repacks and collapses it on the CPU)
        d_rgb, d_ids, d_st = r.render(0, want_hit_ids=True)
        r2.set_world(tw)
        r2.set_samples(grids)
        u_rgb, u_ids, u_st = r2.render(0, want_hit_ids=True)
        assert np.array_equal(d_rgb.view(np.uint32), u_rgb.view(np.uint32)) and np.array_equal(d_ids, u_ids)
        assert d_st.rays_total == u_st.rays_total
        # and the frame is the host-sahBVH frame up to ties: same objects hit except where distances tie
        r2.set_world(world)
        s_rgb, s_ids, _ = r2.render(0, want_hit_ids=True)
        assert (s_ids != d_ids).mean() < 0.01
        # a device-built tree keeps no refit tables
        with pytest.raises(rt.RtError, match="build again"):
            r.refit(world.scene)
        # ... and a host tree uploaded afterwards takes over again
        r.set_world(world)
        b_rgb, _, _ = r.render(0)
        assert np.array_equal(b_rgb.view(np.uint32), s_rgb.view(np.uint32))
        with pytest.raises(rt.RtError, match="not built by rt_build_scene"):
            r.download_tree()
    finally:
        r.close()
        r2.close()


def test_device_build_of_tiny_and_degenerate_scenes(rt, oracle):
    """No object, one object, two coincident objects (equal Morton codes: the object index makes the
    keys distinct)."""
    base = make_world(rt, "bench5", 64, 48, 1)
    v = base.scene.view
    grids = rt.generate_grids(3, 74, base.camera.aa)
    r = rt.Renderer(0)
    try:
        for n_s, dup in ((0, False), (1, False), (2, True)):
            sph = (rt.Sphere * max(1, n_s))(*[v.spheres[0] for _ in range(n_s)]) if dup else \
                  (rt.Sphere * max(1, n_s))(*[v.spheres[i] for i in range(n_s)])
            sv = rt.SceneView(None, 0, None, 0, sph, n_s, v.triangles, 0, v.materials, v.n_materials, v.lights, v.n_lights)
            sv._keep = sph
            w = tree_world(base, sv)
            r.build_world(w)
            r.set_samples(grids)
            view = r.download_tree()
            check_tree(rt, view)
            g_rgb, g_ids, g_st = r.render(0, want_hit_ids=True)
            o_rgb, o_ids, o_st = oracle.render(tree_world(base, view), grids, oracle.FLAG_CULL_ZERO_WEIGHT)
            assert int((g_ids != o_ids).sum()) == 0 and g_st.rays_total == o_st.rays_total
            check_image(rt, g_rgb, o_rgb)
            if n_s == 0:
                assert g_st.primary_hits == 0
    finally:
        r.close()


def test_device_build_at_mesh_size(rt, oracle):
    """The 100 000-triangle mesh: built on the GPU, frame at 480x270 against the oracle on the
    downloaded tree (stripes), and the build is repeatable (same tree twice)."""
    world = make_world(rt, "mesh:100000", 480, 270, 4, view=(16, 9, 10))
    grids = rt.generate_grids(5, 490, world.camera.aa)
    r = rt.Renderer(0)
    try:
        r.build_world(world)
        r.set_samples(grids)
        view = r.download_tree()
        a = np.frombuffer(bytes(view._keep[0]), np.uint8).copy()
        r.build_world(world)
        view2 = r.download_tree()
        assert np.array_equal(a, np.frombuffer(bytes(view2._keep[0]), np.uint8))
        assert view.n_nodes == 2 * view.n_triangles - 1 > 150000
        g_rgb, g_ids, g_st = r.render(0, want_hit_ids=True)
        tw = tree_world(world, view)
        W = 480
        for r0 in (40, 130, 200):
            o_rgb, o_ids, _ = oracle.render(tw, grids, oracle.FLAG_CULL_ZERO_WEIGHT, rows=(r0, r0 + 6))
            sl = slice(r0 * W, (r0 + 6) * W)
            assert int((g_ids[sl] != o_ids[sl]).sum()) == 0
            check_image(rt, g_rgb[sl], o_rgb[sl])
    finally:
        r.close()
