import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def rt():
    import rt_amd
    return rt_amd


@pytest.fixture(scope="session")
def oracle():
    from oracle import binding
    binding.lib()
    return binding


def make_world(rt, name, width, height, rays_per_pixel, bvh="sah", depth=None, view=None, lens=None):
    cfg = rt.default_config(name if not name.startswith("mesh:") else "mesh")
    cfg.image_width, cfg.image_height = width, height
    cfg.anti_aliasing = rays_per_pixel
    if depth is not None:
        cfg.reflection_depth = depth
    if view is not None:
        cfg.view_width, cfg.view_height, cfg.view_distance = view
    if lens is not None:
        cfg.lens = lens
    scene = rt.Scene.builtin(name, bvh=bvh)
    return rt.config_to_world(cfg, scene)
