#!/usr/bin/env python3
"""Longer run of tests/test_gpu_parity.py's random-soup fuzz: seeds A..B-1, both ray policies, GPU
against the oracle (hit ids, ray counts, 8-bit frame).  usage: python3 tools/fuzz_soups.py [A] [B] [--device-build]
--device-build: the scene's tree is built on the GPU (rt_build_scene) and the oracle walks the downloaded tree."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
rt = importlib.import_module("ray-tracer_amd")
from oracle import binding as oracle  # noqa: E402
import test_gpu_parity as T  # noqa: E402


def main():
    device_build = "--device-build" in sys.argv
    args = [x for x in sys.argv[1:] if not x.startswith("--")]
    a = int(args[0]) if len(args) > 0 else 0
    b = int(args[1]) if len(args) > 1 else 200
    oracle.lib()
    r = rt.Renderer(0)
    bad = 0
    for seed in range(a, b):
        w = T._random_soup(rt, seed)
        for gflags, oflags in ((0, oracle.FLAG_CULL_ZERO_WEIGHT), (rt.FLAG_NO_PRUNE | rt.FLAG_TRACE_ZERO_WEIGHT, 0)):
            if device_build:
                from types import SimpleNamespace
                grids = rt.generate_grids(seed, w.width + 10, w.camera.aa)
                r.set_partition(0, 1)
                r.build_world(w)
                r.set_samples(grids)
                g_rgb, g_ids, g_st = r.render(gflags, want_hit_ids=True)
                tw = SimpleNamespace(scene=SimpleNamespace(view=r.download_tree()), camera=w.camera, width=w.width,
                                     height=w.height, spp=w.spp)
                o_rgb, o_ids, o_st = oracle.render(tw, grids, oflags)
            else:
                (g_rgb, g_ids, g_st), (o_rgb, o_ids, o_st) = T.run_both(rt, oracle, r, w, seed, gflags, oflags)
            ids = int((g_ids != o_ids).sum())
            counts = (g_st.rays_primary, g_st.rays_shadow, g_st.rays_reflection, g_st.rays_refraction) == \
                     (o_st.rays_primary, o_st.rays_shadow, o_st.rays_reflection, o_st.rays_refraction)
            fin = np.isfinite(o_rgb).all(axis=1)
            d = np.abs(rt.quantize(g_rgb)[fin].astype(np.int32) - rt.quantize(o_rgb)[fin].astype(np.int32))
            if ids or not counts or d.max() > 1:
                bad += 1
                print(f"seed {seed} flags {gflags}: {ids} ids differ, counts equal {counts}, max 8-bit delta {d.max()}", flush=True)
        if seed % 50 == 49:
            print(f"... seed {seed} done, {bad} failures so far", flush=True)
    r.close()
    print(f"seeds {a}..{b - 1}: {bad} failing (seed, policy) pairs")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
