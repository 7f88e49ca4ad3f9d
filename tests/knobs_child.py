#!/usr/bin/env python3
"""Child process of tests/test_gpu_parity.py::test_exact_verdicts_catch_conservative_errors.

Runs under RT_MI355X_LIB=<the RT_DEBUG_KNOBS build> with RT_DEBUG_BOX_SLOP / RT_DEBUG_UNSAFE_MOD /
RT_DEBUG_NO_RANGE_PRUNE / RT_DEBUG_TRUST_TREE set (rt_init reads them, and the library is bound once
per process: hence a process of its own), or under the shipped library without any of them.
Renders a scene whose LEAF boxes have been shrunk so that they no longer bound their primitives
and compares the GPU with the oracle walking the same tree; prints one JSON line.

    python tests/knobs_child.py <scene> <width> <height> <rays per pixel>
"""
import json
import os
import sys
from types import SimpleNamespace

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import rt_amd as rt                      # noqa: E402
from oracle import binding as oracle     # noqa: E402
from conftest import make_world          # noqa: E402


def shrunk_leaves(w, keep=0.6):
    """The world with every leaf box cut down to `keep` of its extent on its widest axis (from the
    upper side).  Children still lie inside their parents; leaves no longer bound their objects, so
    the reference tests an object only for rays whose line meets what is left of the box."""
    v = w.scene.view
    nodes = (rt.BvhNode * v.n_nodes)(*[v.nodes[i] for i in range(v.n_nodes)])
    for n in nodes:
        if n.left >= 0:
            continue
        ext = [(n.xmax - n.xmin, "x"), (n.ymax - n.ymin, "y"), (n.zmax - n.zmin, "z")]
        e, a = max(ext)
        if e > 0:
            lo = getattr(n, a + "min")
            setattr(n, a + "max", np.float32(lo + np.float32(keep) * np.float32(e)))
    view = rt.SceneView(nodes, v.n_nodes, v.leaf_refs, v.n_refs, v.spheres, v.n_spheres, v.triangles, v.n_triangles,
                        v.materials, v.n_materials, v.lights, v.n_lights)
    view._keep = nodes
    return SimpleNamespace(scene=SimpleNamespace(view=view), camera=w.camera, width=w.width, height=w.height, spp=w.spp)


def main():
    name, width, height, rpp = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    w = make_world(rt, name, width, height, rpp)
    ws = shrunk_leaves(w)
    grids = rt.generate_grids(5, w.width + 10, w.camera.aa)
    r = rt.Renderer(0)
    try:
        r.set_world(w)
        r.set_samples(grids)
        _, ids_intact, _ = r.render(0, want_hit_ids=True)
        r.set_world(ws)
        g_rgb, g_ids, g_st = r.render(0, want_hit_ids=True)
        _, _, c_st = r.render(rt.FLAG_COUNT_TESTS)
    finally:
        r.close()
    o_rgb, o_ids, o_st = oracle.render(ws, grids, oracle.FLAG_CULL_ZERO_WEIGHT)
    finite = np.isfinite(o_rgb).all(axis=1)
    g8, o8 = rt.quantize(g_rgb)[finite], rt.quantize(o_rgb)[finite]
    d = np.abs(g8.astype(np.int32) - o8.astype(np.int32))
    print(json.dumps({
        "lib": os.path.basename(rt.DEVICE_LIB_PATH),
        "ids_changed_by_shrinking": int((g_ids != ids_intact).sum()),
        "id_mismatches": int((g_ids != o_ids).sum()),
        "rays_gpu": [int(g_st.rays_primary), int(g_st.rays_shadow), int(g_st.rays_reflection), int(g_st.rays_refraction)],
        "rays_oracle": [int(o_st.rays_primary), int(o_st.rays_shadow), int(o_st.rays_reflection), int(o_st.rays_refraction)],
        "max_delta8": int(d.max()), "pixels_off": int((d.max(axis=1) > 1).sum()),
        "box_tests": int(c_st.box_tests), "object_tests": int(c_st.sphere_tests + c_st.triangle_tests)}))


if __name__ == "__main__":
    main()
