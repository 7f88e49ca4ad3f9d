"""rt_render_static_async (include/rt_mi355x.h): the one entry point that allocates nothing.
It is rt_render_async with every per-frame work buffer in the library's static device array;
called through the C ABI, its frame, hit ids and ray counters must be BIT-IDENTICAL to
rt_render_async's (integer pixel sums make the frame independent of the launch's wave count),
on two calls in a row, the second with a smaller input (the array keeps the first call's
contents), and it must neither allocate device memory nor accept sizes beyond its limits."""
import numpy as np
import pytest

from conftest import make_world

pytestmark = pytest.mark.gpu


def both(rt, r, torch, world, seed):
    grids = rt.generate_grids(seed, world.width + 10, world.camera.aa)
    r.set_partition(0, 1)
    r.set_world(world)
    r.set_samples(grids)
    n = world.width * world.height
    out = {}
    for name in ("ref", "static"):
        rgb = torch.full((n * 3,), -1.0, dtype=torch.float32, device="cuda:0")
        ids = torch.full((n * world.spp,), 12345, dtype=torch.int32, device="cuda:0")
        torch.cuda.synchronize()
        free0 = torch.cuda.mem_get_info(0)[0]
        if name == "ref":
            r.render_async(rgb.data_ptr(), ids.data_ptr(), 0)
        else:
            r.render_static_async(rgb.data_ptr(), ids.data_ptr(), 0)
        st = r.collect()
        free1 = torch.cuda.mem_get_info(0)[0]
        out[name] = (rgb.cpu().numpy().view(np.uint32), ids.cpu().numpy(), st, free0 - free1)
    return out


def test_static_entry_point_equals_render_async_twice_and_shrinking(rt):
    import torch
    r = rt.Renderer(0)
    try:
        # first call: a 640x360 frame of the headline scene at 4 rays per pixel (921 600 primaries);
        # second call: a SMALLER frame of a different scene (glass + area light, 16 rays per pixel)
        for k, (name, w, h, rpp, seed) in enumerate([("bench1", 640, 360, 4, 3), ("bench2", 200, 120, 16, 4)]):
            world = make_world(rt, name, w, h, rpp)
            o = both(rt, r, torch, world, seed)
            (a_rgb, a_ids, a_st, _), (b_rgb, b_ids, b_st, used) = o["ref"], o["static"]
            assert np.array_equal(a_rgb, b_rgb), f"call {k}: frames differ"
            assert np.array_equal(a_ids, b_ids), f"call {k}: hit ids differ"
            assert (a_st.rays_primary, a_st.rays_shadow, a_st.rays_reflection, a_st.rays_refraction,
                    a_st.primary_hits, a_st.shaded_hits) == \
                   (b_st.rays_primary, b_st.rays_shadow, b_st.rays_reflection, b_st.rays_refraction,
                    b_st.primary_hits, b_st.shaded_hits)
            assert b_st.rays_primary == w * h * rpp and b_st.rays_reflection > 0
            assert b_st.n_waves <= 512                       # the static launch's wave cap
            assert used == 0, f"call {k}: device memory in use grew by {used} bytes"
        # once more the larger one after the smaller: nothing of the small frame's state leaks
        world = make_world(rt, "bench1", 640, 360, 4)
        o = both(rt, r, torch, world, 3)
        assert np.array_equal(o["ref"][0], o["static"][0]) and np.array_equal(o["ref"][1], o["static"][1])
        # a FRESH context whose first frame is the static one: its own work buffers were never
        # allocated (they are allocated on the first rt_render_async), so the frame can only have
        # come out of the static memory
        r2 = rt.Renderer(0)
        try:
            r2.set_world(world)
            r2.set_samples(rt.generate_grids(3, world.width + 10, world.camera.aa))
            rgb = torch.full((world.width * world.height * 3,), -1.0, dtype=torch.float32, device="cuda:0")
            torch.cuda.synchronize()
            free0 = torch.cuda.mem_get_info(0)[0]
            r2.render_static_async(rgb.data_ptr())
            st = r2.collect()
            assert torch.cuda.mem_get_info(0)[0] == free0
            assert np.array_equal(rgb.cpu().numpy().view(np.uint32), o["ref"][0]) and st.rays_total == o["ref"][2].rays_total
        finally:
            r2.close()
    finally:
        r.close()


def test_static_entry_point_rejects_what_exceeds_its_memory(rt):
    import torch
    r = rt.Renderer(0)
    try:
        buf = torch.zeros(1920 * 1080 * 3, dtype=torch.float32, device="cuda:0")
        world = make_world(rt, "bench1", 1920, 1080, 1)
        r.set_world(world)
        r.set_samples(rt.generate_grids(1, 1930, 1))
        with pytest.raises(rt.RtError, match="exceed the static work memory's limit of 262144"):
            r.render_static_async(buf.data_ptr())
        world = make_world(rt, "bench1", 64, 48, 1, depth=7)
        r.set_world(world)
        r.set_samples(rt.generate_grids(1, 74, 1))
        with pytest.raises(rt.RtError, match="max_depth 7 exceeds"):
            r.render_static_async(buf.data_ptr())
        r.set_partition(1, 2)
        world = make_world(rt, "bench1", 64, 48, 1)
        r.set_world(world)
        with pytest.raises(rt.RtError, match="whole frames"):
            r.render_static_async(buf.data_ptr())
        r.set_partition(0, 1)
        r.render_static_async(buf.data_ptr())             # within the limits: the context still works
        st = r.collect()
        assert st.rays_primary == 64 * 48
    finally:
        r.close()
