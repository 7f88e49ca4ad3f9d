"""The N-GPU form of the C ABI (rt_multi_*, include/rt_mi355x.h): one host process, one rt_ctx
per GPU, tiles t % N, one exchange step per frame (RCCL point-to-point gather on rank 0), assembly.
A one-GPU box can run (i) N = 1 through RCCL itself (library binding, communicator, grouped
send/recv on the render stream) and (ii) N > 1 ranks SHARING the GPU, where the exchange falls back
to device-to-device copies -- same buffers, strides and ordering as the RCCL branch.  More GPUs are
the driver's to run.  Frames must equal the single-context frame bit for bit."""
import os
import subprocess

import numpy as np
import pytest

from conftest import make_world

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def single(rt, world, grids, flags=0):
    r = rt.Renderer(0)
    try:
        r.set_world(world)
        r.set_samples(grids)
        rgb, _, st = r.render(flags)
        return rgb, st
    finally:
        r.close()


@pytest.mark.parametrize("devices", [[0], [0, 0, 0], [0] * 8])
def test_multi_frame_is_bitwise_the_single_context_frame(rt, devices):
    m = rt.MultiRenderer(devices)
    try:
        # two frames in a row on one rt_multi: 200x150 (sides not multiples of the 16x16 tiles), then
        # a larger frame of another scene (buffers grow), 8-bit gather included
        for name, w, h, rpp, seed in (("bench1", 200, 150, 4, 9), ("bench2", 320, 200, 16, 5)):
            world = make_world(rt, name, w, h, rpp)
            grids = rt.generate_grids(seed, w + 10, world.camera.aa)
            ref, ref_st = single(rt, world, grids)
            m.set_world(world)
            m.set_samples(grids)
            rgb, _, st = m.render(0)
            assert np.array_equal(rgb.view(np.uint32), ref.view(np.uint32))
            assert (st.rays_primary, st.rays_shadow, st.rays_reflection, st.rays_refraction, st.primary_hits) == \
                   (ref_st.rays_primary, ref_st.rays_shadow, ref_st.rays_reflection, ref_st.rays_refraction,
                    ref_st.primary_hits)
            _, rgb8, _ = m.render(0, want_float=False, want_8bit=True)       # ranks quantise, bytes are gathered
            assert np.array_equal(rgb8, rt.quantize(ref))
            both_f, both_8, _ = m.render(0, want_float=True, want_8bit=True)  # floats gathered, root quantises
            assert np.array_equal(both_f.view(np.uint32), ref.view(np.uint32)) and np.array_equal(both_8, rgb8)
    finally:
        m.close()


@pytest.mark.parametrize("devices", [[0], [0, 0, 0]])
def test_frames_in_flight_through_submit_and_wait(rt, devices):
    """rt_multi_submit / rt_multi_wait: K rt_multi on the same devices, gone round by one host
    thread, keep K frames in flight.  Every frame is bitwise the single-context frame, whatever
    was in flight beside it; misuse (two submits, a wait without a submit) is an error."""
    K = 3
    worlds = [make_world(rt, n, 160, 120, 4) for n in ("bench1", "bench5", "bench2")]
    grids = [rt.generate_grids(31 + i, 170, w.camera.aa) for i, w in enumerate(worlds)]
    refs = [single(rt, w, g)[0] for w, g in zip(worlds, grids)]
    ms = [rt.MultiRenderer(devices) for _ in range(K)]
    try:
        for i, m in enumerate(ms):
            m.set_world(worlds[i])
            m.set_samples(grids[i])
        for m in ms:
            m.submit(0)
        for rnd in range(4):
            for i, m in enumerate(ms):
                rgb, _, st = m.wait()
                assert np.array_equal(rgb.view(np.uint32), refs[i].view(np.uint32))
                assert st.rays_primary == 160 * 120 * 4
                m.submit(0, want_float=(rnd % 2 == 0), want_8bit=True)
                if rnd % 2 == 1:
                    with pytest.raises(rt.RtError, match="has not been waited for"):
                        m.submit(0)
                    _, rgb8, _ = m.wait()
                    assert np.array_equal(rgb8, rt.quantize(refs[i]))
                    m.submit(0)
        for m in ms:
            m.wait()
        with pytest.raises(rt.RtError, match="no frame has been submitted"):
            ms[0].wait()
    finally:
        for m in ms:
            m.close()


def test_inputs_are_locked_while_a_frame_is_in_flight(rt):
    """rt_multi_wait sizes what it hands over from the frame that was submitted; scene, camera and
    samples cannot be changed under a frame in flight (RT_ERR_STATE), and the frame that was in flight
    arrives whole."""
    big = make_world(rt, "bench1", 160, 120, 4)
    small = make_world(rt, "bench5", 64, 48, 1)
    g_big, g_small = rt.generate_grids(3, 170, big.camera.aa), rt.generate_grids(4, 74, small.camera.aa)
    ref_big, ref_small = single(rt, big, g_big)[0], single(rt, small, g_small)[0]
    m = rt.MultiRenderer([0, 0])
    try:
        m.set_world(big)
        m.set_samples(g_big)
        m.submit(0)
        with pytest.raises(rt.RtError, match="in flight"):
            m.set_world(small)
        with pytest.raises(rt.RtError, match="in flight"):
            m.set_samples(g_small)
        rgb, _, _ = m.wait()
        assert rgb.shape == ref_big.shape and np.array_equal(rgb.view(np.uint32), ref_big.view(np.uint32))
        m.set_world(small)
        m.set_samples(g_small)
        rgb, _, _ = m.render(0)
        assert np.array_equal(rgb.view(np.uint32), ref_small.view(np.uint32))
    finally:
        m.close()


def test_multi_with_device_built_trees(rt):
    """rt_multi_build_scene: every rank builds the tree on its GPU; the assembled frame is bitwise the
    frame of one context that built the same tree."""
    world = make_world(rt, "bench3", 200, 150, 4)
    grids = rt.generate_grids(8, 210, world.camera.aa)
    r = rt.Renderer(0)
    try:
        r.build_world(world)
        r.set_samples(grids)
        ref, _, ref_st = r.render(0)
    finally:
        r.close()
    m = rt.MultiRenderer([0, 0, 0])
    try:
        m.build_world(world)
        m.set_samples(grids)
        rgb, _, st = m.render(0)
        assert np.array_equal(rgb.view(np.uint32), ref.view(np.uint32)) and st.rays_total == ref_st.rays_total
    finally:
        m.close()


def test_multi_reports_errors(rt):
    with pytest.raises(rt.RtError, match="rt_init"):
        rt.MultiRenderer([0, 4711])                    # no such GPU
    m = rt.MultiRenderer([0, 0])
    try:
        with pytest.raises(rt.RtError, match="camera and samples"):
            m.world = make_world(rt, "bench1", 32, 24, 1)
            m.render(0)
    finally:
        m.close()


def test_c_harness_drives_the_abi_from_plain_c(rt, oracle, tmp_path):
    """integration/rt_harness.c (gcc, no Python, no torch): host library -> World, device library
    -> frame, writePPM; plus the rt_multi path with one GPU through RCCL.  The P3 file it writes
    must be the oracle's frame to the parity bar."""
    exe = os.path.join(ROOT, "ray-tracer_amd", "lib", "rt_harness")
    assert os.path.exists(exe), "rt_harness not built (run __graft_entry__.build())"
    out = os.path.join(tmp_path, "img.ppm")
    p = subprocess.run([exe, "bench1", "160", "120", "4", out, "1"], capture_output=True, text=True, timeout=110)
    assert p.returncode == 0, p.stderr[-1000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("scene=")][-1]   # (RCCL may print banners of its own)
    kv = dict(x.split("=") for x in line.split())
    assert kv["multi_frame_identical"] == "1" and kv["multi_8bit_identical"] == "1"
    toks = open(out).read().split()
    assert toks[:4] == ["P3", "160", "120", "255"]
    img = np.array(toks[4:], np.int32).reshape(-1, 3)
    world = make_world(rt, "bench1", 160, 120, 4)
    grids = rt.generate_grids(2024, 170, 2)
    o_rgb, _, o_st = oracle.render(world, grids, oracle.FLAG_CULL_ZERO_WEIGHT, want_hit_ids=False)
    assert int(kv["rays"]) == o_st.rays_total == int(kv["multi_rays"])
    assert np.abs(img - rt.quantize(o_rgb).astype(np.int32)).max() <= 1
