"""Build recipes for the native libraries (called by __graft_entry__.build()).

    librt_host.so    g++     host-side scene / BVH / sample-table producer
    librt_mi355x.so  hipcc   the C-ABI device library (gfx950); kernels compiled twice,
                             -ffp-contract=off (strict) and -ffp-contract=fast (fma)
    librt_mi355x_dbg.so      the same with -DRT_DEBUG_KNOBS (test-only knobs in the hot loop)
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib")
INC = os.path.join(ROOT, "include")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"


def _run(cmd):
    print("+", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)


def _stale(out, srcs):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(s) > t for s in srcs)


def build_host(force=False):
    os.makedirs(LIB, exist_ok=True)
    out = os.path.join(LIB, "librt_host.so")
    srcs = [os.path.join(CSRC, "host", "rt_host.cpp"), os.path.join(INC, "rt_host.h"),
            os.path.join(INC, "rt_mi355x.h")]
    if force or _stale(out, srcs):
        _run(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-Wall",
              "-I" + INC, srcs[0], "-o", out])
    return out


def build_device(force=False, defines=(), tag="", cflags=()):
    """tag != "" builds an experiment variant lib/librt_mi355x<tag>.so with extra -D flags
    (select it at run time with RT_MI355X_LIB=<path>)."""
    os.makedirs(LIB, exist_ok=True)
    obj = os.path.join(HERE, "build" + tag)
    os.makedirs(obj, exist_ok=True)
    out = os.path.join(LIB, f"librt_mi355x{tag}.so")
    common = [os.path.join(CSRC, "rt_device.h"), os.path.join(CSRC, "rt_fastdiv.h"), os.path.join(INC, "rt_mi355x.h")]
    kern = os.path.join(CSRC, "rt_kernels.inl")
    units = [
        ("rt_kernels_strict", ["-ffp-contract=off"], [kern]),
        ("rt_kernels_fma", ["-ffp-contract=fast"], [kern]),
        ("rt_api", ["-ffp-contract=off"], [os.path.join(CSRC, "rt_build.inl")]),
        ("rt_multi", ["-ffp-contract=off"], []),
    ]
    # -fno-slp-vectorize: left to itself the SLP vectoriser pairs the kernels' scalar float operations into
    # v_pk_mul_f32 / v_pk_add_f32 (IEEE per element, so the same numbers), whose operands must sit in aligned
    # register pairs: the copies and the pressure cost more than the packed issue saves -- 128 -> 118-125
    # VGPRs, 14 -> 8 spilled in the five-wave kernel, c2 1.60 -> 1.50 ms per frame.
    base = [HIPCC, f"--offload-arch={ARCH}", "-O3", "-fno-slp-vectorize", "-std=c++17", "-fPIC", "-I" + INC, "-I" + CSRC]
    base += ["-D" + d for d in defines] + list(cflags)
    objs = []
    relink = force
    for name, flags, extra in units:
        src = os.path.join(CSRC, name + ".hip")
        o = os.path.join(obj, name + ".o")
        if force or _stale(o, [src] + common + extra):
            _run(base + flags + ["-c", src, "-o", o])
            relink = True
        objs.append(o)
    if relink or _stale(out, objs):
        _run([HIPCC, f"--offload-arch={ARCH}", "-shared", "-fPIC"] + objs + ["-ldl", "-o", out])
    return out


def build_harness(force=False):
    """integration/rt_harness.c: the C ABI driven from plain C (gcc), linked against the two
    libraries only (run by tests/test_gpu_multi.py on the GPU box)."""
    src = os.path.join(ROOT, "integration", "rt_harness.c")
    out = os.path.join(LIB, "rt_harness")
    deps = [src, os.path.join(INC, "rt_host.h"), os.path.join(INC, "rt_mi355x.h"),
            os.path.join(LIB, "librt_host.so"), os.path.join(LIB, "librt_mi355x.so")]
    if force or _stale(out, deps):
        _run(["gcc", "-std=c11", "-O2", "-Wall", "-I" + INC, src, "-L" + LIB, "-lrt_mi355x", "-lrt_host",
              "-Wl,-rpath,$ORIGIN", "-Wl,-rpath-link," + LIB, "-o", out])
    return out


def build_all(force=False):
    h, d = build_host(force), build_device(force)
    build_harness(force)
    # the RT_DEBUG_KNOBS variant (tests only: tests/test_gpu_parity.py::test_exact_verdicts_catch_conservative_errors)
    build_device(force, ["RT_DEBUG_KNOBS"], "_dbg")
    return h, d


if __name__ == "__main__":
    defs = [a[2:] for a in sys.argv[1:] if a.startswith("-D")]
    tags = [a[6:] for a in sys.argv[1:] if a.startswith("--tag=")]
    cflags = [a[8:] for a in sys.argv[1:] if a.startswith("--cflag=")]   # experiments: extra hipcc flags
    if tags:
        build_device(True, defs, tags[0], cflags)
    else:
        build_all("--force" in sys.argv)
