"""The C-ABI libraries load and export every entry point their headers declare.
No compute call is made (no GPU needed: dlopen + dlsym only)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DECL = re.compile(r"^\s*(?:const\s+)?[A-Za-z_][A-Za-z0-9_]*\s*\**\s*\b(rth?_[a-z0-9_]+)\s*\(", re.M)


def declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)          # strip comments
    return sorted(set(DECL.findall(text)))


def test_device_library_exports_every_declared_symbol(rt):
    names = declared("rt_mi355x.h")
    assert "rt_render" in names and "rt_init" in names and len(names) >= 15
    assert os.path.exists(rt.DEVICE_LIB_PATH), "librt_mi355x.so not built (run __graft_entry__.build())"
    # RTLD_LAZY load of a HIP library works without a GPU; nothing is called
    lib = C.CDLL(rt.DEVICE_LIB_PATH)
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    # the Python mirror binds exactly the declared set
    assert sorted(rt.DEVICE_SYMBOLS) == names
    lib.rt_abi_version.restype = C.c_int
    assert lib.rt_abi_version() == 1   # host-only call: returns a constant


def test_host_library_exports_every_declared_symbol(rt):
    names = declared("rt_host.h")
    lib = C.CDLL(rt.HOST_LIB_PATH)
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    assert sorted(rt.HOST_SYMBOLS) == names


def test_pod_sizes_match_the_header(rt):
    """ctypes mirrors of the PODs have the C layout (sizes from the field lists in
    include/rt_mi355x.h)."""
    assert C.sizeof(rt.BvhNode) == 32
    assert C.sizeof(rt.Sphere) == 24
    assert C.sizeof(rt.Triangle) == 56
    assert C.sizeof(rt.Material) == 48
    assert C.sizeof(rt.Light) == 48
    assert C.sizeof(rt.Camera) == 4 * (2 + 3 + 1 + 1 + 6 + 9 + 3 + 1)
    assert C.sizeof(rt.Stats) == 12 * 8 + 3 * 8


def test_render_without_gpu_fails_loudly(rt):
    """No CPU fallback: on a box without a GPU rt_init must fail and the wrapper raise."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(rt.RtError):
        rt.Renderer(0)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "ray-tracer_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".inl", ".cpp")):
                text = open(os.path.join(dirpath, fn), errors="ignore").read()
                assert "rt_oracle" not in text and "from oracle" not in text and \
                    "import oracle" not in text, os.path.join(dirpath, fn)
