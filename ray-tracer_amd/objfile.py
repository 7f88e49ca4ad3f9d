"""Writer for `.obj` / `.mtl` pairs in the dialect src/HaObj.hs reads (the file starts
with `mtllib`; per object `o`, then `v*`, then `usemtl [s] f+` groups; `.mtl` entries are
`newmtl, Ns, Ka, Kd, Ks, [Ni], d, illum` in that order).  The reference's own `station.obj` is not
in its repository, so config 3 (BASELINE.json: ".obj mesh scene via HaObj + BVH") is fed with the
host library's generated station-style mesh written out in that dialect and read back through
the loader (`Scene.from_obj` = parseObj + convertMesh + fromMesh): bench.py --from-obj and the tests.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

# the .mtl entries that build_mesh's materials correspond to (ray-tracer_amd/csrc/host/rt_host.cpp):
# material index -> (name, illum, Kd, Ks, Ns, Ni)
MESH_MTL = {
    0: ("hub", 3, (0.6, 0.6, 0.65), (1, 1, 1), 100, 1),
    1: ("ring", 3, (1, 0.842, 0.1), (1, 0.842, 0.1), 50, 1),
    2: ("pod", 6, (0, 0, 0), (1, 1, 1), 1000, 1.5),
    3: ("pod2", 4, (0.2, 0.3, 0.9), (1, 1, 1), 100, 1.3),
    4: ("spoke", 2, (0.9, 0.15, 0.1), (0.5, 0.5, 0.5), 10, 1),
    5: ("floor_a", 2, (1, 1, 1), (0.5, 0.5, 0.5), 5, 1),
    6: ("floor_b", 2, (0.15, 0.15, 0.2), (0.5, 0.5, 0.5), 5, 1),
}


def fmt(x) -> str:
    """Shortest decimal that reads back as the same float32."""
    return np.format_float_positional(np.float32(x), unique=True, trim="0")


def triangles_array(view) -> np.ndarray:
    """The scene's rt_triangle records as a structured array (no per-record ctypes access)."""
    dt = np.dtype([("a", "<f4", 3), ("b", "<f4", 3), ("c", "<f4", 3), ("n", "<f4", 3), ("material", "<i4"), ("object_id", "<i4")])
    n = view.n_triangles
    if n == 0:
        return np.zeros(0, dt)
    buf = C.string_at(view.triangles, n * dt.itemsize)
    return np.frombuffer(buf, dt)


def dump_obj(scene, path: str) -> None:
    """Write a triangle scene (materials as in MESH_MTL): one object, consecutive faces of one
    material form one `usemtl` group with `s off`.  Nine significant digits: every coordinate
    reads back as the same float32."""
    v = scene.view
    tri = triangles_array(v)
    with open(path[:-4] + ".mtl", "w") as f:
        for name, illum, kd, ks, ns, ni in MESH_MTL.values():
            f.write(f"newmtl {name}\nNs {fmt(ns)}\nKa 0 0 0\nKd {' '.join(map(fmt, kd))}\n"
                    f"Ks {' '.join(map(fmt, ks))}\nNi {fmt(ni)}\nd 1\nillum {illum}\n\n")
    with open(path, "w") as f:
        f.write(f"# generated\nmtllib {os.path.basename(path)[:-4]}.mtl\no station\n")
        if len(tri):
            verts = np.stack([tri["a"], tri["b"], tri["c"]], axis=1).reshape(-1, 3).astype(np.float64)
            np.savetxt(f, verts, fmt="v %.9g %.9g %.9g")
        mats = tri["material"]
        starts = np.flatnonzero(np.r_[True, mats[1:] != mats[:-1]]) if len(tri) else np.zeros(0, np.int64)
        ends = np.r_[starts[1:], len(tri)]
        for s0, s1 in zip(starts, ends):
            f.write(f"usemtl {MESH_MTL[int(mats[s0])][0]}\ns off\n")
            i = np.arange(s0, s1, dtype=np.int64)
            np.savetxt(f, np.stack([3 * i + 1, 3 * i + 2, 3 * i + 3], axis=1), fmt="f %d %d %d")
