"""ray-tracer_amd -- MI355X-native trace/shade path of jrraymond/ray-tracer.

Host-side mirror (Python, ctypes) of the reference interface around ``render``:

    reference (Haskell)                         here
    -----------------------------------------   ---------------------------------
    Config / benchNConfig   (Main.hs:50-62)     Config, default_config(name)
    configToWorld, getCam   (Main.hs:104-126)   config_to_world(config, scene)
    sahBVH / noBVH          (BVH.hs:34,130)     Scene.build_bvh("sah" | "none")
    generateGrids           (RayTracer.hs:462)  generate_grids(seed, num, aa)
    render rng world        (RayTracer.hs:18)   render(grids, world, ...)
    writePPM                (Main.hs:128-138)   write_ppm(path, w, h, pixels)

The product path is librt_mi355x.so (hand-written HIP, C ABI in
include/rt_mi355x.h).  There is NO CPU fallback: ``render`` raises if the HIP
library is missing or no GPU is present.  The package name contains a hyphen, so
import it with ``importlib.import_module("ray-tracer_amd")`` (see rt_amd.py at the
repository root for a one-line shim).
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass, field
from typing import Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(_HERE, "lib")
HOST_LIB_PATH = os.path.join(LIB_DIR, "librt_host.so")
DEVICE_LIB_PATH = os.environ.get("RT_MI355X_LIB") or os.path.join(LIB_DIR, "librt_mi355x.so")

# --------------------------------------------------------------------------
# PODs of include/rt_mi355x.h / include/rt_host.h
# --------------------------------------------------------------------------
f32, i32, u32, u64 = C.c_float, C.c_int32, C.c_uint32, C.c_uint64


class BvhNode(C.Structure):
    _fields_ = [("xmin", f32), ("xmax", f32), ("ymin", f32), ("ymax", f32), ("zmin", f32),
                ("zmax", f32), ("left", i32), ("right", i32)]


class Sphere(C.Structure):
    _fields_ = [("center", f32 * 3), ("radius", f32), ("material", i32), ("object_id", i32)]


class Triangle(C.Structure):
    _fields_ = [("a", f32 * 3), ("b", f32 * 3), ("c", f32 * 3), ("n", f32 * 3), ("material", i32),
                ("object_id", i32)]


class Material(C.Structure):
    _fields_ = [("diffuse", f32 * 3), ("specular", f32 * 3), ("phong", f32), ("refl", f32),
                ("refr_ix", f32), ("log_atten", f32 * 3)]


class Light(C.Structure):
    _fields_ = [("corner", f32 * 3), ("edge_a", f32 * 3), ("edge_b", f32 * 3), ("color", f32 * 3)]


class SceneView(C.Structure):
    _fields_ = [("nodes", C.POINTER(BvhNode)), ("n_nodes", i32),
                ("leaf_refs", C.POINTER(u32)), ("n_refs", i32),
                ("spheres", C.POINTER(Sphere)), ("n_spheres", i32),
                ("triangles", C.POINTER(Triangle)), ("n_triangles", i32),
                ("materials", C.POINTER(Material)), ("n_materials", i32),
                ("lights", C.POINTER(Light)), ("n_lights", i32)]


class Camera(C.Structure):
    _fields_ = [("img_w", i32), ("img_h", i32), ("view_w", f32), ("view_h", f32), ("view_d", f32),
                ("aa", i32), ("lens", f32), ("up", f32 * 3), ("eye", f32 * 3), ("u", f32 * 3),
                ("v", f32 * 3), ("w", f32 * 3), ("ambient", f32 * 3), ("max_depth", i32)]


class Stats(C.Structure):
    _fields_ = [("rays_primary", u64), ("rays_shadow", u64), ("rays_reflection", u64),
                ("rays_refraction", u64), ("box_tests", u64), ("sphere_tests", u64),
                ("triangle_tests", u64), ("primary_hits", u64), ("shaded_hits", u64),
                ("wave_ticks_sum", u64), ("wave_ticks_max", u64), ("n_waves", u64),
                ("ms_kernel", C.c_double), ("ms_launch", C.c_double), ("ms_total", C.c_double)]

    @property
    def rays_total(self) -> int:
        return (self.rays_primary + self.rays_shadow + self.rays_reflection + self.rays_refraction)

    def as_dict(self) -> dict:
        d = {k: getattr(self, k) for k, _ in self._fields_}
        d["rays_total"] = self.rays_total
        return d


class Tiling(C.Structure):
    _fields_ = [("tile_w", i32), ("tile_h", i32), ("tiles_x", i32), ("tiles_y", i32),
                ("n_tiles", i32), ("n_local_tiles", i32)]


class HostConfig(C.Structure):
    _fields_ = [("image_w", i32), ("image_h", i32), ("view_w", i32), ("view_h", i32),
                ("view_d", i32), ("reflection_depth", i32), ("anti_aliasing", i32), ("lens", f32),
                ("up", f32 * 3), ("eye", f32 * 3), ("look_at", f32 * 3)]


REF_TRIANGLE = 0x80000000
HIT_NONE = 0xFFFFFFFF
FLAG_FMA = 0x1
FLAG_TRACE_ZERO_WEIGHT = 0x2
FLAG_NO_PRUNE = 0x4
FLAG_COUNT_TESTS = 0x8

DEVICE_SYMBOLS = [
    "rt_init", "rt_shutdown", "rt_upload_scene", "rt_set_camera", "rt_set_samples",
    "rt_set_partition", "rt_get_tiling", "rt_render", "rt_render_async", "rt_render_tiles_async",
    "rt_assemble_async", "rt_collect", "rt_quantize_async", "rt_last_error", "rt_abi_version",
    "rt_wave_diag", "rt_get_stream", "rt_set_concurrency", "rt_set_timing", "rt_render_static_async",
    "rt_assemble8_async", "rt_multi_init", "rt_multi_shutdown", "rt_multi_upload_scene", "rt_multi_set_camera",
    "rt_multi_set_samples", "rt_multi_render", "rt_multi_submit", "rt_multi_wait", "rt_multi_set_concurrency", "rt_multi_last_error", "rt_refit_async", "rt_multi_refit",
    "rt_build_scene", "rt_download_tree", "rt_multi_build_scene",
]
HOST_SYMBOLS = [
    "rth_scene_builtin", "rth_scene_create", "rth_scene_free", "rth_scene_load_obj",
    "rth_scene_add_light", "rth_scene_add_bench6_lights", "rth_scene_build_bvh",
    "rth_scene_view", "rth_scene_n_objects", "rth_scene_bvh_depth", "rth_default_config",
    "rth_config_to_camera", "rth_generate_grids", "rth_write_ppm", "rth_quantize",
    "rth_write_ppm_binary",
    "rth_material_by_name", "rth_make_triangle", "rth_generate_grids_pure64", "rth_mt19937_64",
]


class RtError(RuntimeError):
    pass


# --------------------------------------------------------------------------
# library loading
# --------------------------------------------------------------------------
_host = None
_dev = None


def host_lib() -> C.CDLL:
    global _host
    if _host is None:
        if not os.path.exists(HOST_LIB_PATH):
            raise RtError(f"{HOST_LIB_PATH} missing: run `python -c 'import __graft_entry__ as g; "
                          "g.build()'` in the repository root")
        L = C.CDLL(HOST_LIB_PATH)
        L.rth_scene_builtin.restype = C.c_void_p
        L.rth_scene_builtin.argtypes = [C.c_char_p]
        L.rth_scene_create.restype = C.c_void_p
        L.rth_scene_create.argtypes = [C.POINTER(Sphere), i32, C.POINTER(Triangle), i32,
                                       C.POINTER(u32), i32, C.POINTER(Material), i32,
                                       C.POINTER(Light), i32]
        L.rth_scene_load_obj.restype = C.c_void_p
        L.rth_scene_load_obj.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
        L.rth_scene_add_light.argtypes = [C.c_void_p, C.POINTER(Light)]
        L.rth_scene_add_bench6_lights.argtypes = [C.c_void_p]
        L.rth_scene_free.argtypes = [C.c_void_p]
        L.rth_scene_free.restype = None
        L.rth_scene_build_bvh.argtypes = [C.c_void_p, C.c_int]
        L.rth_scene_view.argtypes = [C.c_void_p, C.POINTER(SceneView)]
        L.rth_scene_n_objects.argtypes = [C.c_void_p]
        L.rth_scene_bvh_depth.argtypes = [C.c_void_p]
        L.rth_default_config.argtypes = [C.c_char_p, C.POINTER(HostConfig)]
        L.rth_config_to_camera.argtypes = [C.POINTER(HostConfig), C.POINTER(Camera)]
        L.rth_generate_grids.argtypes = [u64, i32, i32, C.POINTER(f32)]
        L.rth_generate_grids_pure64.argtypes = [u64, i32, i32, C.POINTER(f32)]
        L.rth_mt19937_64.argtypes = [u64, i32, C.POINTER(u64)]
        L.rth_write_ppm.argtypes = [C.c_char_p, i32, i32, C.POINTER(f32)]
        L.rth_quantize.argtypes = [C.POINTER(f32), C.c_size_t, C.POINTER(C.c_uint8)]
        L.rth_quantize.restype = None
        L.rth_write_ppm_binary.argtypes = [C.c_char_p, i32, i32, C.POINTER(C.c_uint8)]
        L.rth_material_by_name.argtypes = [C.c_char_p, C.POINTER(Material)]
        L.rth_make_triangle.argtypes = [f32 * 3, f32 * 3, f32 * 3, i32, C.POINTER(Triangle)]
        L.rth_make_triangle.restype = None
        _host = L
    return _host


def _share_hip_runtime_with_torch() -> None:
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own
    libamdhip64.so (soname libamdhip64.so.7) and ask for it as "libamdhip64.so"; if
    librt_mi355x.so pulled in /opt/rocm's copy first, a later `import torch` would load
    a second runtime that sees no GPU.  When a torch wheel is installed (and not yet
    imported) load ITS runtime first, so both sides share it.  Plumbing only."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except Exception:
        spec = None
    if spec is None or not spec.origin:
        return
    p = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(p):
        try:
            C.CDLL(p, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def device_lib() -> C.CDLL:
    """The HIP library.  Fails loudly when it has not been built (no fallback)."""
    global _dev
    if _dev is None:
        if not os.path.exists(DEVICE_LIB_PATH):
            raise RtError(f"{DEVICE_LIB_PATH} missing: the HIP extension is not built; there is "
                          "no CPU fallback for the trace/shade path")
        _share_hip_runtime_with_torch()
        L = C.CDLL(DEVICE_LIB_PATH)
        vp = C.c_void_p
        L.rt_init.argtypes = [C.c_int, C.POINTER(vp)]
        L.rt_shutdown.argtypes = [vp]
        L.rt_shutdown.restype = None
        L.rt_upload_scene.argtypes = [vp, C.POINTER(SceneView)]
        L.rt_set_camera.argtypes = [vp, C.POINTER(Camera)]
        L.rt_refit_async.argtypes = [vp, C.POINTER(Sphere), i32, C.POINTER(Triangle), i32, vp]
        L.rt_set_samples.argtypes = [vp, C.POINTER(f32), i32, i32]
        L.rt_set_partition.argtypes = [vp, i32, i32]
        L.rt_get_tiling.argtypes = [vp, C.POINTER(Tiling)]
        L.rt_render.argtypes = [vp, u32, C.POINTER(f32), C.POINTER(u32), C.POINTER(Stats)]
        L.rt_render_async.argtypes = [vp, u32, vp, vp, vp]
        L.rt_render_tiles_async.argtypes = [vp, u32, vp, vp]
        L.rt_render_static_async.argtypes = [vp, u32, vp, vp, vp]
        L.rt_assemble_async.argtypes = [vp, vp, i32, C.c_size_t, vp, vp]
        L.rt_assemble8_async.argtypes = [vp, vp, i32, C.c_size_t, vp, vp]
        L.rt_multi_init.argtypes = [i32, C.POINTER(i32), C.POINTER(vp)]
        L.rt_multi_shutdown.argtypes = [vp]
        L.rt_multi_shutdown.restype = None
        L.rt_multi_upload_scene.argtypes = [vp, C.POINTER(SceneView)]
        L.rt_multi_set_camera.argtypes = [vp, C.POINTER(Camera)]
        L.rt_multi_set_samples.argtypes = [vp, C.POINTER(f32), i32, i32]
        L.rt_multi_render.argtypes = [vp, u32, C.POINTER(f32), C.POINTER(C.c_uint8), C.POINTER(Stats)]
        L.rt_multi_submit.argtypes = [vp, u32, C.POINTER(f32), C.POINTER(C.c_uint8)]
        L.rt_multi_wait.argtypes = [vp, C.POINTER(Stats)]
        L.rt_multi_set_concurrency.argtypes = [vp, i32]
        L.rt_multi_refit.argtypes = [vp, C.POINTER(Sphere), i32, C.POINTER(Triangle), i32]
        L.rt_multi_build_scene.argtypes = [vp, C.POINTER(Sphere), i32, C.POINTER(Triangle), i32, C.POINTER(Material), i32,
                                           C.POINTER(Light), i32]
        L.rt_multi_last_error.argtypes = [vp]
        L.rt_multi_last_error.restype = C.c_char_p
        L.rt_collect.argtypes = [vp, vp, C.POINTER(Stats)]
        L.rt_quantize_async.argtypes = [vp, vp, vp, C.c_size_t, vp]
        L.rt_wave_diag.argtypes = [vp, C.POINTER(u64), i32]
        L.rt_build_scene.argtypes = [vp, C.POINTER(Sphere), i32, C.POINTER(Triangle), i32, C.POINTER(Material), i32,
                                     C.POINTER(Light), i32]
        L.rt_download_tree.argtypes = [vp, C.POINTER(BvhNode), i32, C.POINTER(u32), i32, C.POINTER(i32), C.POINTER(i32)]
        L.rt_get_stream.argtypes = [vp, C.POINTER(vp)]
        L.rt_set_concurrency.argtypes = [vp, i32]
        L.rt_set_timing.argtypes = [vp, i32]
        L.rt_last_error.argtypes = [vp]
        L.rt_last_error.restype = C.c_char_p
        L.rt_abi_version.restype = C.c_int
        _dev = L
    return _dev


# --------------------------------------------------------------------------
# host-side mirror of the reference API
# --------------------------------------------------------------------------
Vec3 = Tuple[float, float, float]


@dataclass
class Config:
    """``Config`` of src/Main.hs:50-62 (cScene is the `scene` passed separately)."""
    image_width: int = 400
    image_height: int = 300
    view_width: int = 4
    view_height: int = 3
    view_distance: int = 7
    reflection_depth: int = 3
    anti_aliasing: int = 1          # rays per pixel
    lens: float = 0.0
    up: Vec3 = (0.0, 1.0, 0.0)
    eye: Vec3 = (0.0, 0.0, 0.0)
    look_at: Vec3 = (1.0, 0.0, 0.0)

    def _c(self) -> HostConfig:
        return HostConfig(self.image_width, self.image_height, self.view_width, self.view_height,
                          self.view_distance, self.reflection_depth, self.anti_aliasing, self.lens,
                          (f32 * 3)(*self.up), (f32 * 3)(*self.eye), (f32 * 3)(*self.look_at))


def default_config(name: str) -> Config:
    """benchNConfig of src/Main.hs:151-235 ("bench1" .. "bench6", "scene1")."""
    hc = HostConfig()
    if host_lib().rth_default_config(name.encode(), C.byref(hc)) != 0:
        raise RtError(f"unknown config {name!r}")
    return Config(hc.image_w, hc.image_h, hc.view_w, hc.view_h, hc.view_d, hc.reflection_depth,
                  hc.anti_aliasing, hc.lens, tuple(hc.up), tuple(hc.eye), tuple(hc.look_at))


class Scene:
    """[Object] + [Light] (+ the BVH built from them): wObjects / wLights."""

    def __init__(self, handle: int):
        if not handle:
            raise RtError("scene construction failed")
        self._h = C.c_void_p(handle)
        self._view: Optional[SceneView] = None

    @classmethod
    def builtin(cls, name: str, bvh: str = "sah") -> "Scene":
        s = cls(host_lib().rth_scene_builtin(name.encode()))
        s.build_bvh(bvh)
        return s

    @classmethod
    def create(cls, spheres: Sequence[Sphere], triangles: Sequence[Triangle],
               order: Sequence[int], materials: Sequence[Material], lights: Sequence[Light],
               bvh: str = "sah") -> "Scene":
        sa = (Sphere * max(1, len(spheres)))(*spheres)
        ta = (Triangle * max(1, len(triangles)))(*triangles)
        oa = (u32 * max(1, len(order)))(*order)
        ma = (Material * max(1, len(materials)))(*materials)
        la = (Light * max(1, len(lights)))(*lights)
        s = cls(host_lib().rth_scene_create(sa, len(spheres), ta, len(triangles), oa, len(order),
                                            ma, len(materials), la, len(lights)))
        s.build_bvh(bvh)
        return s

    @classmethod
    def from_obj(cls, path: str, lights: Optional[Sequence[Light]] = None, bvh: str = "sah") -> "Scene":
        """parseObj + convertMesh + fromMesh (HaObj.hs:63, Convert.hs:15-39).  `lights`
        None = the eight point lights of bench6World (Main.hs:239-246)."""
        import time
        err = C.create_string_buffer(512)
        t0 = time.perf_counter()
        h = host_lib().rth_scene_load_obj(path.encode(), err, 512)
        if not h:
            raise RtError(f"cannot load {path}: {err.value.decode()}")
        s = cls(h)
        if lights is None:
            host_lib().rth_scene_add_bench6_lights(s._h)
        else:
            for l in lights:
                host_lib().rth_scene_add_light(s._h, C.byref(l))
        t1 = time.perf_counter()
        s.build_bvh(bvh)
        # (what bench.py --from-obj reports: parse + convert, then sahBVH)
        s.load_seconds, s.build_seconds = t1 - t0, time.perf_counter() - t1
        return s

    def build_bvh(self, builder: str = "sah") -> None:
        """sahBVH (BoundingVolumeHierarchy.hs:34-51) or noBVH (:130-131)."""
        code = {"sah": 0, "none": 1}[builder]
        if host_lib().rth_scene_build_bvh(self._h, code) != 0:
            raise RtError("BVH build failed")
        self._view = None

    @property
    def view(self) -> SceneView:
        if self._view is None:
            v = SceneView()
            host_lib().rth_scene_view(self._h, C.byref(v))
            self._view = v
        return self._view

    @property
    def n_objects(self) -> int:
        return host_lib().rth_scene_n_objects(self._h)

    @property
    def bvh_depth(self) -> int:
        return host_lib().rth_scene_bvh_depth(self._h)

    def nodes(self) -> np.ndarray:
        v = self.view
        dt = np.dtype([("box", "<f4", (6,)), ("left", "<i4"), ("right", "<i4")])
        if v.n_nodes == 0:
            return np.zeros(0, dt)
        return np.ctypeslib.as_array(C.cast(v.nodes, C.POINTER(C.c_uint8)),
                                     (v.n_nodes * 32,)).view(dt).copy()

    def leaf_refs(self) -> np.ndarray:
        v = self.view
        if v.n_refs == 0:
            return np.zeros(0, np.uint32)
        return np.ctypeslib.as_array(v.leaf_refs, (v.n_refs,)).copy()

    def __del__(self):
        try:
            if self._h:
                host_lib().rth_scene_free(self._h)
                self._h = None
        except Exception:
            pass


@dataclass
class World:
    """``World`` of src/Types.hs:132-146: camera scalars + scene."""
    camera: Camera
    scene: Scene

    @property
    def width(self) -> int:
        return self.camera.img_w

    @property
    def height(self) -> int:
        return self.camera.img_h

    @property
    def spp(self) -> int:
        return self.camera.aa * self.camera.aa


def config_to_world(config: Config, scene: Scene) -> World:
    """configToWorld (src/Main.hs:104-120); the scene already carries objects+lights."""
    cam = Camera()
    hc = config._c()
    if host_lib().rth_config_to_camera(C.byref(hc), C.byref(cam)) != 0:
        raise RtError("config_to_world failed")
    return World(cam, scene)


def generate_grids(seed: int, num: int, aa: int) -> np.ndarray:
    """generateGrids (src/RayTracer.hs:462-467) before `cycle`: [num][aa*aa][6] float32."""
    out = np.empty((num, aa * aa, 6), np.float32)
    rc = host_lib().rth_generate_grids(seed, num, aa, out.ctypes.data_as(C.POINTER(f32)))
    if rc != 0:
        raise RtError("generate_grids failed")
    return out


def generate_grids_pure64(seed: int, num: int, aa: int) -> np.ndarray:
    """generateGrids (pureMT seed) num aa, seed-level: the table a GHC build of the reference would
    produce from that seed with mersenne-random-pure64 0.2 / random 1.1 / random-shuffle 0.0.4
    (restated from the packages' published algorithms; PARITY UNPINNED, see include/rt_host.h)."""
    out = np.empty((num, aa * aa, 6), np.float32)
    if host_lib().rth_generate_grids_pure64(seed, num, aa, out.ctypes.data_as(C.POINTER(f32))) != 0:
        raise RtError("generate_grids_pure64 failed")
    return out


def mt19937_64(seed: int, n: int) -> np.ndarray:
    out = np.empty(n, np.uint64)
    if host_lib().rth_mt19937_64(seed, n, out.ctypes.data_as(C.POINTER(u64))) != 0:
        raise RtError("mt19937_64 failed")
    return out


def quantize(rgb: np.ndarray) -> np.ndarray:
    """toTxt of writePPM (src/Main.hs:131): truncate(255 * clamp x) as uint8."""
    a = np.ascontiguousarray(rgb, np.float32)
    out = np.empty(a.shape, np.uint8)
    host_lib().rth_quantize(a.ctypes.data_as(C.POINTER(f32)), a.size,
                            out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out


def write_ppm(path: str, w: int, h: int, pixels: np.ndarray) -> None:
    """writePPM (src/Main.hs:128-138): ASCII P3, one line of values."""
    a = np.ascontiguousarray(pixels, np.float32)
    assert a.size == w * h * 3
    if host_lib().rth_write_ppm(path.encode(), w, h, a.ctypes.data_as(C.POINTER(f32))) != 0:
        raise RtError(f"cannot write {path}")


def write_ppm_binary(path: str, w: int, h: int, rgb8: np.ndarray) -> None:
    """Binary P6 of already quantised pixels (what rt_quantize_async produces on the GPU)."""
    a = np.ascontiguousarray(rgb8, np.uint8)
    assert a.size == w * h * 3
    if host_lib().rth_write_ppm_binary(path.encode(), w, h, a.ctypes.data_as(C.POINTER(C.c_uint8))) != 0:
        raise RtError(f"cannot write {path}")


def material(name: str) -> Material:
    m = Material()
    if host_lib().rth_material_by_name(name.encode(), C.byref(m)) != 0:
        raise RtError(f"unknown material {name!r}")
    return m


def make_triangle(a: Vec3, b: Vec3, c: Vec3, material_index: int) -> Triangle:
    t = Triangle()
    host_lib().rth_make_triangle((f32 * 3)(*a), (f32 * 3)(*b), (f32 * 3)(*c), material_index,
                                 C.byref(t))
    return t


# --------------------------------------------------------------------------
# the device path
# --------------------------------------------------------------------------
class Renderer:
    """One rt_ctx (one GPU).  Thin, explicit wrapper over the C ABI."""

    def __init__(self, device: int = 0):
        self.lib = device_lib()
        h = C.c_void_p()
        rc = self.lib.rt_init(device, C.byref(h))
        if rc != 0 or not h:
            raise RtError(f"rt_init(device={device}) failed with {rc}: is a GPU present?")
        self._h = h
        self.world: Optional[World] = None

    def _check(self, rc: int, what: str) -> None:
        if rc != 0:
            msg = self.lib.rt_last_error(self._h)
            raise RtError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")

    def set_world(self, world: World) -> None:
        self._check(self.lib.rt_upload_scene(self._h, C.byref(world.scene.view)), "rt_upload_scene")
        self._check(self.lib.rt_set_camera(self._h, C.byref(world.camera)), "rt_set_camera")
        self.world = world

    def build_world(self, world: World) -> None:
        """rt_build_scene: the world's objects, materials and lights with a tree built ON THE GPU
        (linear BVH) instead of the world's own (host sahBVH) tree; camera as set_world."""
        v = world.scene.view
        self._check(self.lib.rt_build_scene(self._h, v.spheres, v.n_spheres, v.triangles, v.n_triangles, v.materials,
                                            v.n_materials, v.lights, v.n_lights), "rt_build_scene")
        self._check(self.lib.rt_set_camera(self._h, C.byref(world.camera)), "rt_set_camera")
        self.world = world

    def download_tree(self) -> SceneView:
        """The device-built tree as a SceneView over the current world's objects (rt_download_tree):
        what rt_upload_scene -- or the oracle -- takes."""
        nn, nr = i32(0), i32(0)
        self._check(self.lib.rt_download_tree(self._h, None, 0, None, 0, C.byref(nn), C.byref(nr)), "rt_download_tree")
        nodes = (BvhNode * max(1, nn.value))()
        refs = (u32 * max(1, nr.value))()
        self._check(self.lib.rt_download_tree(self._h, nodes, nn.value, refs, nr.value, C.byref(nn), C.byref(nr)),
                    "rt_download_tree")
        v = self.world.scene.view
        out = SceneView(nodes, nn.value, refs, nr.value, v.spheres, v.n_spheres, v.triangles, v.n_triangles,
                        v.materials, v.n_materials, v.lights, v.n_lights)
        out._keep = (nodes, refs, v)
        return out

    def refit(self, scene: "Scene") -> None:
        """rt_refit_async: the uploaded scene's objects moved to `scene`'s positions (same counts
        and order); boxes refitted on the GPU, the tree's topology kept."""
        v = scene.view
        self._check(self.lib.rt_refit_async(self._h, v.spheres, v.n_spheres, v.triangles, v.n_triangles, None),
                    "rt_refit_async")

    def set_samples(self, grids: np.ndarray) -> None:
        g = np.ascontiguousarray(grids, np.float32)
        assert g.ndim == 3 and g.shape[2] == 6
        self._check(self.lib.rt_set_samples(self._h, g.ctypes.data_as(C.POINTER(f32)), g.shape[0],
                                            g.shape[1]), "rt_set_samples")

    def set_partition(self, rank: int, nranks: int) -> None:
        self._check(self.lib.rt_set_partition(self._h, rank, nranks), "rt_set_partition")

    def set_concurrency(self, n_concurrent: int) -> None:
        self._check(self.lib.rt_set_concurrency(self._h, n_concurrent), "rt_set_concurrency")

    def set_timing(self, on: bool) -> None:
        """Record (default) or skip the events behind Stats.ms_launch / ms_kernel."""
        self._check(self.lib.rt_set_timing(self._h, 1 if on else 0), "rt_set_timing")

    def tiling(self) -> Tiling:
        t = Tiling()
        self._check(self.lib.rt_get_tiling(self._h, C.byref(t)), "rt_get_tiling")
        return t

    def render(self, flags: int = 0, want_hit_ids: bool = False):
        """rt_render with host buffers.  Returns (rgb[H*W,3], hit_ids or None, Stats)."""
        w = self.world
        assert w is not None
        rgb = np.zeros((w.height * w.width, 3), np.float32)
        ids = np.zeros((w.height * w.width, w.spp), np.uint32) if want_hit_ids else None
        st = Stats()
        self._check(self.lib.rt_render(self._h, flags, rgb.ctypes.data_as(C.POINTER(f32)),
                                       ids.ctypes.data_as(C.POINTER(u32)) if want_hit_ids else None,
                                       C.byref(st)), "rt_render")
        return rgb, ids, st

    # device-pointer forms (pointers are plain integers, e.g. torch.Tensor.data_ptr())
    def render_async(self, d_rgb: int, d_hit_ids: int = 0, flags: int = 0, stream: int = 0) -> None:
        self._check(self.lib.rt_render_async(self._h, flags, d_rgb, d_hit_ids or None,
                                             stream or None), "rt_render_async")

    def render_static_async(self, d_rgb: int, d_hit_ids: int = 0, flags: int = 0, stream: int = 0) -> None:
        """rt_render_static_async: rt_render_async on the library's static work memory (no
        allocation; at most 262 144 pixels, depth 6; calls on one GPU must not overlap)."""
        self._check(self.lib.rt_render_static_async(self._h, flags, d_rgb, d_hit_ids or None,
                                                    stream or None), "rt_render_static_async")

    def render_tiles_async(self, d_tiles: int, flags: int = 0, stream: int = 0) -> None:
        self._check(self.lib.rt_render_tiles_async(self._h, flags, d_tiles, stream or None),
                    "rt_render_tiles_async")

    def assemble_async(self, d_gathered: int, nranks: int, stride_floats: int, d_rgb: int,
                       stream: int = 0) -> None:
        self._check(self.lib.rt_assemble_async(self._h, d_gathered, nranks, stride_floats, d_rgb,
                                               stream or None), "rt_assemble_async")

    def quantize_async(self, d_rgb: int, d_rgb8: int, n: int, stream: int = 0) -> None:
        self._check(self.lib.rt_quantize_async(self._h, d_rgb, d_rgb8, n, stream or None),
                    "rt_quantize_async")

    def collect(self, stream: int = 0) -> Stats:
        st = Stats()
        self._check(self.lib.rt_collect(self._h, stream or None, C.byref(st)), "rt_collect")
        return st

    @property
    def stream(self) -> int:
        """The context's own hipStream_t as an integer (for torch.cuda.ExternalStream)."""
        p = C.c_void_p()
        self._check(self.lib.rt_get_stream(self._h, C.byref(p)), "rt_get_stream")
        return p.value or 0

    def wave_diag(self, n_waves: int) -> np.ndarray:
        """[n_waves, 4] (lifetime ticks @100 MHz, rays, start tick, dry tick) of the last render."""
        out = np.zeros((n_waves, 4), np.uint64)
        self._check(self.lib.rt_wave_diag(self._h, out.ctypes.data_as(C.POINTER(u64)), n_waves),
                    "rt_wave_diag")
        return out

    def close(self) -> None:
        if self._h:
            self.lib.rt_shutdown(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MultiRenderer:
    """One rt_multi: one host process rendering each frame on N GPUs (tiles t % N, RCCL gather
    on rank 0).  `devices` may repeat ids to rehearse the N-rank data path on fewer GPUs."""

    def __init__(self, devices: Sequence[int]):
        self.lib = device_lib()
        h = C.c_void_p()
        arr = (i32 * len(devices))(*devices)
        rc = self.lib.rt_multi_init(len(devices), arr, C.byref(h))
        self._h = h if h else None
        if rc != 0:
            msg = self.lib.rt_multi_last_error(h).decode() if h else ""
            self.close()
            raise RtError(f"rt_multi_init({list(devices)}) failed with {rc}: {msg}")
        self.world: Optional[World] = None

    def _check(self, rc: int, what: str) -> None:
        if rc != 0:
            msg = self.lib.rt_multi_last_error(self._h)
            raise RtError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")

    def set_world(self, world: World) -> None:
        self._check(self.lib.rt_multi_upload_scene(self._h, C.byref(world.scene.view)), "rt_multi_upload_scene")
        self._check(self.lib.rt_multi_set_camera(self._h, C.byref(world.camera)), "rt_multi_set_camera")
        self.world = world

    def set_samples(self, grids: np.ndarray) -> None:
        g = np.ascontiguousarray(grids, np.float32)
        self._check(self.lib.rt_multi_set_samples(self._h, g.ctypes.data_as(C.POINTER(f32)), g.shape[0],
                                                  g.shape[1]), "rt_multi_set_samples")

    def build_world(self, world: World) -> None:
        """rt_multi_build_scene: the tree built on every GPU instead of uploaded."""
        v = world.scene.view
        self._check(self.lib.rt_multi_build_scene(self._h, v.spheres, v.n_spheres, v.triangles, v.n_triangles, v.materials,
                                                  v.n_materials, v.lights, v.n_lights), "rt_multi_build_scene")
        self._check(self.lib.rt_multi_set_camera(self._h, C.byref(world.camera)), "rt_multi_set_camera")
        self.world = world

    def refit(self, scene: "Scene") -> None:
        v = scene.view
        self._check(self.lib.rt_multi_refit(self._h, v.spheres, v.n_spheres, v.triangles, v.n_triangles), "rt_multi_refit")

    def render(self, flags: int = 0, want_float: bool = True, want_8bit: bool = False):
        """Returns (rgb [H*W,3] f32 or None, rgb8 [H*W,3] u8 or None, Stats)."""
        w = self.world
        assert w is not None and (want_float or want_8bit)
        rgb = np.zeros((w.height * w.width, 3), np.float32) if want_float else None
        rgb8 = np.zeros((w.height * w.width, 3), np.uint8) if want_8bit else None
        st = Stats()
        self._check(self.lib.rt_multi_render(self._h, flags,
                                             rgb.ctypes.data_as(C.POINTER(f32)) if want_float else None,
                                             rgb8.ctypes.data_as(C.POINTER(C.c_uint8)) if want_8bit else None,
                                             C.byref(st)), "rt_multi_render")
        return rgb, rgb8, st

    def submit(self, flags: int = 0, want_float: bool = True, want_8bit: bool = False) -> None:
        """rt_multi_submit: enqueue one frame and return; wait() hands it over.  A host keeps K frames
        in flight with K MultiRenderers on the same devices."""
        w = self.world
        assert w is not None and (want_float or want_8bit)
        rgb = np.zeros((w.height * w.width, 3), np.float32) if want_float else None
        rgb8 = np.zeros((w.height * w.width, 3), np.uint8) if want_8bit else None
        self._check(self.lib.rt_multi_submit(self._h, flags,
                                             rgb.ctypes.data_as(C.POINTER(f32)) if want_float else None,
                                             rgb8.ctypes.data_as(C.POINTER(C.c_uint8)) if want_8bit else None),
                    "rt_multi_submit")
        self._rgb, self._rgb8 = rgb, rgb8   # (kept alive, and only replaced once the frame is accepted)

    def set_concurrency(self, n: int) -> None:
        self._check(self.lib.rt_multi_set_concurrency(self._h, n), "rt_multi_set_concurrency")

    def wait(self):
        """rt_multi_wait: (rgb or None, rgb8 or None, Stats) of the frame submitted last."""
        st = Stats()
        self._check(self.lib.rt_multi_wait(self._h, C.byref(st)), "rt_multi_wait")
        return self._rgb, self._rgb8, st

    def close(self) -> None:
        if self._h:
            self.lib.rt_multi_shutdown(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def render(grids: np.ndarray, world: World, flags: int = 0, device: int = 0) -> np.ndarray:
    """``render rng world`` (src/RayTracer.hs:18-23) with the grids given explicitly.

    Returns the [Color] list as float32 [H*W, 3], top row first.  Runs on the GPU
    through librt_mi355x.so; raises RtError when that is impossible.
    """
    r = Renderer(device)
    try:
        r.set_world(world)
        r.set_samples(grids)
        rgb, _, _ = r.render(flags)
        return rgb
    finally:
        r.close()


def tile_shape(spp: int) -> Tuple[int, int]:
    """Tile decomposition used by the device library (pure index math, mirrored here
    for the CPU-side multi-rank tests): about 1024 samples per tile, power-of-two
    sides, width >= height."""
    px = max(1, 1024 // max(1, spp))
    lg = max(0, px.bit_length() - 1)
    tw = 1 << ((lg + 1) // 2)
    th = 1 << (lg // 2)
    return tw, th


def tiles_of_rank(width: int, height: int, spp: int, rank: int, nranks: int):
    """Global tile ids of `rank` under the static interleaved partition (t % nranks)."""
    tw, th = tile_shape(spp)
    tx = (width + tw - 1) // tw
    ty = (height + th - 1) // th
    return [t for t in range(tx * ty) if t % nranks == rank], (tw, th, tx, ty)
