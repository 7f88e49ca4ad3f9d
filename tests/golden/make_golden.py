#!/usr/bin/env python3
"""Generate the golden fixtures tests/golden/*.npz.

    python tests/golden/make_golden.py

Each fixture holds the INPUTS (scene name, config, the sample table itself) and the
outputs of the CPU oracle (oracle/rt_oracle.c) under the reference's semantics (every
reflection ray traced, un-pruned traversal): per-sample primary hit ids, the float
frame, the 8-bit frame of writePPM and the ray counts.

These vectors come from OUR restatement, not from the Haskell reference (no GHC in the
image; the reference itself ships no vectors): they pin the oracle
against regressions and give the GPU tests a fixed target, they do not prove the oracle
right -- tests/test_oracle_kat.py carries the hand-derived answers for that.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import rt_amd as rt  # noqa: E402
from conftest import make_world  # noqa: E402
from oracle import binding as ob  # noqa: E402

# name, scene, width, height, rays per pixel, lens, seed
CASES = [
    ("bench1_c1", "bench1", 80, 60, 1, None, 101),     # configs[0]: 1 spp
    ("bench1_c2", "bench1", 96, 54, 16, None, 102),    # configs[1] shape (16:9, 16 spp)
    ("bench2", "bench2", 64, 48, 4, None, 103),
    ("bench3", "bench3", 64, 48, 4, None, 104),
    ("bench4_dof", "bench4", 64, 36, 16, 0.05, 105),   # configs[3]: depth of field
    ("bench5_glossy", "bench5", 64, 36, 16, None, 106),  # configs[3]: glossy reflections
    ("mesh", "mesh:3000", 64, 36, 4, None, 107),       # configs[2]: station-style mesh
]


def main():
    for name, scene, w, h, rpp, lens, seed in CASES:
        view = (16, 9, 10) if w * 9 == h * 16 else None
        world = make_world(rt, scene, w, h, rpp, lens=lens, view=view)
        grids = rt.generate_grids(seed, w + 10, world.camera.aa)
        rgb, ids, st = ob.render(world, grids, 0)
        rgb_c, ids_c, st_c = ob.render(world, grids, ob.FLAG_CULL_ZERO_WEIGHT)
        assert np.array_equal(ids, ids_c)
        small = np.uint16 if ids[ids != rt.HIT_NONE].max(initial=0) < 65535 else np.uint32
        none = np.iinfo(small).max
        ids_s = np.where(ids == rt.HIT_NONE, none, ids).astype(small)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(
            path, scene=scene, width=w, height=h, rays_per_pixel=rpp,
            lens=-1.0 if lens is None else lens, view=np.array(view or (0, 0, 0)), seed=seed,
            grids=grids, hit_ids=ids_s, hit_none=none, rgb=rgb, rgb8=rt.quantize(rgb),
            rays=np.array([st.rays_primary, st.rays_shadow, st.rays_reflection, st.rays_refraction],
                          np.uint64),
            rays_culled=np.array([st_c.rays_primary, st_c.rays_shadow, st_c.rays_reflection,
                                  st_c.rays_refraction], np.uint64),
            rgb8_culled=rt.quantize(rgb_c), nonfinite_samples=st.nonfinite_samples)
        print(f"{name}: {os.path.getsize(path)} bytes, rays {st.rays_total}, "
              f"nonfinite samples {st.nonfinite_samples}")


if __name__ == "__main__":
    main()
