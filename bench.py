#!/usr/bin/env python3
"""bench.py -- Mrays/s and 1080p frame time of the trace/shade path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload c2]

One step = one rendered frame of the workload (one pass of the hot path over one batch of
synthetic input).  Default workload "c2" is BASELINE.json configs[1]: BenchmarkScene
(bench1) at 1920x1080, view plane 16x9x10, 16 rays per pixel (aa 4), area-light soft
shadows, reflection depth 6.  With N > 1 (launched by torch.distributed.run, one rank per
GPU) the tiles of the SAME frame are split statically across the ranks (t % N), each rank
renders its tiles into a compact buffer, one RCCL gather brings them to rank 0, which
assembles the framebuffer: fixed total work => "scaling": "strong".

Inputs (scene, BVH, sample table) are uploaded before the timed region; the frame stays in
HBM.  value = rays traversed by all ranks / max-over-ranks wall time.  Two frames are kept in
flight (two rt_ctx on two streams, --frames-in-flight): frames are independent, and the next
frame's blocks fill the wave slots that the tail of the current one leaves empty.  The
roofline leg times render_kernel alone (one frame in flight) with HIP events.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Every context renders on its own stream and the gather / assembly run on two more: give each its
# own hardware queue (HIP's default of 4 makes a fifth stream share one and serialise with it; up to
# six contexts, two streams of the collective path, torch's own).
# Must be set before the runtime starts.
if int(os.environ.get("GPU_MAX_HW_QUEUES", "0") or 0) < 16:
    os.environ["GPU_MAX_HW_QUEUES"] = os.environ.get("RT_HW_QUEUES", "16")

import numpy as np
import torch
import torch.distributed as dist

import rt_amd as rt
import importlib
rtd = importlib.import_module("ray-tracer_amd.distributed")

PEAK_VALU_TFLOPS = 157.3   # MI355X fp32 vector peak (MI355X_MICROARCH.md, chip-level parameters)
PEAK_HBM_GBPS = 8000.0     # HBM3E spec
PEAK_VALU_ISSUE = 1228.8   # G wave64 instructions/s: 1024 SIMDs x 2.4 GHz / 2 cycles (the issue rate behind PEAK_VALU_TFLOPS)
BYTES_PER_RAY = 128        # 56-B ray record written+read once + 8-B hit record x2
SEED = 0x5EED1234

_ORBIT = None

WORKLOADS = {
    # name: (scene, config name, width, height, (view w,h,d), rays per pixel, lens)
    "c1": ("bench1", "bench1", 800, 600, (8, 6, 7), 1, None),
    "c2": ("bench1", "bench1", 1920, 1080, (16, 9, 10), 16, None),
    "c3": ("mesh:100000", "mesh", 1920, 1080, (16, 9, 10), 4, None),
    "c4-glossy": ("bench5", "bench5", 1920, 1080, (16, 9, 10), 64, None),
    "c4-dof": ("bench4", "bench4", 1920, 1080, (16, 9, 10), 64, 0.05),
    # config 5: animation at 4K -- every step is the NEXT frame of the orbit (objects
    # re-evaluated, sahBVH rebuilt on the host, scene re-uploaded inside the timed region)
    "c5": ("orbit animation", None, 3840, 2160, (16, 9, 8), 4, None),
}


LOADER = {}   # --from-obj: what the loader path took (seconds), reported in config


def scene_through_obj_loader(scene_name: str):
    """Config 3 "via HaObj" (BASELINE.json configs[2]): the generated mesh written as .obj/.mtl in
    the HaObj dialect, read back by parseObj + convertMesh + fromMesh (src/HaObj.hs:63,
    src/Convert.hs:15-39, restated in rth_scene_load_obj) and given to sahBVH -- instead of the
    programmatic scene.  Same lights as the programmatic scene (bench6World's, moved with the mesh)."""
    import tempfile
    objfile = importlib.import_module("ray-tracer_amd.objfile")
    src = rt.Scene.builtin(scene_name)
    d = tempfile.mkdtemp(prefix="rt_obj_")
    path = os.path.join(d, "station.obj")
    t0 = time.perf_counter()
    objfile.dump_obj(src, path)
    t1 = time.perf_counter()
    scene = rt.Scene.from_obj(path, lights=[src.view.lights[i] for i in range(src.view.n_lights)])
    a, b = objfile.triangles_array(src.view), objfile.triangles_array(scene.view)
    same = (len(a) == len(b) and all(np.array_equal(a[f], b[f]) for f in ("a", "b", "c", "n"))
            and np.array_equal(src.nodes()["box"], scene.nodes()["box"]) and np.array_equal(src.leaf_refs(), scene.leaf_refs()))
    LOADER.update({"obj_bytes": os.path.getsize(path), "triangles": int(scene.view.n_triangles),
                   "write_obj_s": round(t1 - t0, 3), "parse_convert_s": round(scene.load_seconds, 3),
                   "sahBVH_s": round(scene.build_seconds, 3), "same_triangles_and_tree_as_programmatic_scene": bool(same)})
    for f in (path, path[:-4] + ".mtl"):
        os.remove(f)
    os.rmdir(d)
    return scene


def make_world(name: str, t: float = 0.0, from_obj: bool = False):
    scene_name, cfg_name, w, h, view, rpp, lens = WORKLOADS[name]
    if cfg_name is None:
        anim = importlib.import_module("ray-tracer_amd.animation")
        global _ORBIT
        if _ORBIT is None:
            _ORBIT = anim.parse_scene_description(anim.ORBIT_DESCRIPTION)
        cfg = rt.Config(image_width=w, image_height=h, anti_aliasing=rpp, reflection_depth=6)
        return _ORBIT.world_at(t, cfg)
    cfg = rt.default_config(cfg_name)
    cfg.image_width, cfg.image_height = w, h
    cfg.view_width, cfg.view_height, cfg.view_distance = view
    cfg.anti_aliasing = rpp
    if lens is not None:
        cfg.lens = lens
    scene = scene_through_obj_loader(scene_name) if from_obj else rt.Scene.builtin(scene_name)
    return rt.config_to_world(cfg, scene)


def algorithmic_flops(st) -> float:
    """N_box*18 + N_sph*37 + N_tri*56 + shade, shade = 140 flops per shaded
    hit (60 direct + 80 bounce set-up) + 40 per shadow ray (light vector, Blinn-Phong)."""
    return (18.0 * st.box_tests + 37.0 * st.sphere_tests + 56.0 * st.triangle_tests
            + 140.0 * st.shaded_hits + 40.0 * st.rays_shadow)


def pmc_traffic(workload: str):
    """HBM bytes per launch of render_kernel from the committed PMC pass (rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE in separate runs, tools/prof_pmc.sh -> profiles/): counters
    cannot be read inside this process.  FETCH_SIZE (KB) is doubled as MI355X_MICROARCH.md
    prescribes for gfx950 (it tallies 128-B requests at 64 B); WRITE_SIZE (KB) is exact."""
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]_{workload}_pmc.txt")))
    if not found:
        return None, None, {}
    path = found[-1]   # the newest round's pass
    vals = {}
    for line in open(path):
        f = line.split()
        if len(f) >= 3 and f[0] == "#" and f[1] == "source_sha256":
            vals["_source_sha256"] = f[2]
        if len(f) >= 3 and f[-1].startswith("avg="):
            vals[f[0]] = float(f[-1][4:])
    if "FETCH_SIZE" not in vals or "WRITE_SIZE" not in vals:
        return None, None, vals
    return (2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0, os.path.relpath(path, ROOT), vals


def pmc_is_stale(pmc) -> bool:
    """True when the committed counter pass was taken with other kernel / host sources than the ones
    in this tree (tools/pmc_summary.py stamps every pass with their SHA-256; an unstamped pass
    counts as stale): roofline.traffic and valu_issue then mix this run's times with old counters."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import pmc_summary
        return pmc.get("_source_sha256") != pmc_summary.source_sha256()
    except Exception:
        return True


def cpu_baseline(world, grids, seconds_target: float = 25.0):
    """Time the CPU restatement (oracle, kind 'port') single-threaded -- the reference's
    render is a sequential map -- on stripes of rows of the SAME frame."""
    from oracle import binding as ob
    H = world.height
    stripes = [(r, min(H, r + 27)) for r in range(0, H, 54)]  # every other 27-row stripe: half the frame
    rays = 0
    t0 = time.perf_counter()
    rows = 0
    for (a, b) in stripes:
        _, _, st = ob.render(world, grids, 0, rows=(a, b), threads=1, want_hit_ids=False)
        rays += st.rays_total
        rows += b - a
        if time.perf_counter() - t0 > seconds_target:
            break
    dt = time.perf_counter() - t0
    out = {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": 1, "kind": "port",
           "sample": f"{rows} of {H} rows of the same frame (27-row stripes every 54 rows), "
                     f"{rays} rays in {dt:.1f} s; reference ray accounting (every reflection ray "
                     "traced); restatement of the reference algorithm, not the Haskell binary"}
    # Beside it: the same restatement with OpenMP over rows on the host cores of this
    # box's share -- what a parallel CPU port would reach; the reference itself is sequential.
    threads = max(1, min(16, ob.lib().rto_max_threads(), os.cpu_count() or 1))
    if threads > 1:
        # whole frames of the same workload until at least 4 s have passed (a sub-second sample was
        # +-15 % from run to run)
        t1 = time.perf_counter()
        rays1, frames = 0, 0
        while time.perf_counter() - t1 < 4.0:
            _, _, st = ob.render(world, grids, 0, threads=threads, want_hit_ids=False)
            rays1 += st.rays_total
            frames += 1
        dt1 = time.perf_counter() - t1
        out["all_cores"] = {"value": rays1 / dt1 / 1e6, "unit": "Mrays/s", "cores": threads,
                            "sample": f"{frames} whole frame(s) of the same workload, {rays1} rays in {dt1:.1f} s"}
    return out


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--flags", type=int, default=0, help="rt_render flags (1 = FMA build)")
    ap.add_argument("--frames-in-flight", type=int, default=6)
    ap.add_argument("--share-gpu", default="auto", choices=["auto", "yes", "no"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-collective", action="store_true",
                    help="self-test: run the gather/assemble path with one rank and check the frame")
    ap.add_argument("--shard", default="tiles", choices=["tiles", "frames"],
                    help="N > 1: 'tiles' = every frame split over the ranks + RCCL gather (default); 'frames' "
                         "(animation, SURVEY 8e / config 5) = rank r renders whole frames r, r+N, ... with no "
                         "data-path collective")
    ap.add_argument("--rebuild-every", type=int, default=1, metavar="K",
                    help="animation (c5): a context rebuilds sahBVH on the host and uploads only for every K-th of "
                         "its frames and refits the boxes on the GPU in between (rt_refit_async); 1 = rebuild "
                         "every frame, as the reference's host would")
    ap.add_argument("--device-build", action="store_true",
                    help="animation (c5): every frame's tree is built on the GPU (rt_build_scene, a linear BVH) instead of "
                         "sahBVH on the host + upload; a tree-changing mode (ties may resolve differently)")
    ap.add_argument("--from-obj", action="store_true",
                    help="mesh workloads (c3): the scene goes dump_obj -> Scene.from_obj (parseObj/convertMesh/fromMesh) -> "
                         "sahBVH instead of the programmatic scene; loader and build seconds are reported in config")
    ap.add_argument("--rehearse-share", default=None, metavar="R/N",
                    help="one-GPU timing rehearsal of rank R's share of an N-rank job (no collective); "
                         "the line then reports that share's rays only")
    args = ap.parse_args()

    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world_size != args.gpus:
        if world_size == 1 and args.gpus > 1:
            raise SystemExit("launch with python -m torch.distributed.run --nproc-per-node N for --gpus N")
        args.gpus = world_size
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the trace/shade path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world_size > 1 or args.force_collective:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group("nccl", rank=rank, world_size=world_size, device_id=dev)

    if args.from_obj and not WORKLOADS[args.workload][0].startswith("mesh:"):
        raise SystemExit("--from-obj: only the mesh workloads have an .obj form (c3)")
    world = make_world(args.workload, from_obj=args.from_obj)
    grids = rt.generate_grids(SEED, world.width + 10, world.camera.aa)

    def make_renderer():
        r = rt.Renderer(local_rank)
        r.set_world(world)
        r.set_samples(grids)
        return r

    # F contexts render alternate frames on their own streams (ray-tracer_amd/distributed.py):
    # the next frame's blocks fill the wave slots the current frame's tail leaves empty
    by_frames = args.shard == "frames" and world_size > 1
    df = rtd.DistributedFrame(make_renderer, 0 if by_frames else rank, 1 if by_frames else world_size, dev,
                              in_flight=args.frames_in_flight,
                              force_collective=args.force_collective, share_gpu=args.share_gpu,
                              rehearse_share=tuple(int(x) for x in args.rehearse_share.split("/"))
                              if args.rehearse_share else None,
                              tune_worlds=(lambda k: make_world(args.workload, float(k)))
                              if WORKLOADS[args.workload][1] is None else None)

    def fence():
        df.finish()
        if world_size > 1:
            dist.barrier()
        torch.cuda.synchronize()

    animated = WORKLOADS[args.workload][1] is None
    frame_no = [0]

    per_context = {}

    def submit():
        if animated:
            frame_no[0] += 1
            # frame index: this rank's own sequence, or every world_size-th frame of the job's
            k = frame_no[0] * world_size + rank if by_frames else frame_no[0]
            slot = df.k % df.depth
            n = per_context.get(slot, 0)
            per_context[slot] = n + 1
            df.submit(args.flags, make_world(args.workload, float(k)), refit=(n % max(1, args.rebuild_every)) != 0,
                      device_build=args.device_build)
        else:
            df.submit(args.flags)

    # The animated workload builds a scene in Python every frame; a generation-2 collection of the
    # cyclic garbage collector (tens of ms with torch's object graph loaded) in the middle of a
    # 40 ms timed region doubled the measured frame time at random.  Collect now, not then.
    import gc
    gc.collect()
    gc.disable()
    for _ in range(args.warmup):
        submit()
    fence()
    st = df.r.collect(df.streams[0].cuda_stream)   # per-frame ray counters of this rank (same every frame)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        submit()
    fence()
    dt = time.perf_counter() - t0
    gc.enable()

    rays = torch.tensor([st.rays_total, st.rays_primary, st.rays_shadow,
                         st.rays_reflection + st.rays_refraction], dtype=torch.float64, device=dev)
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world_size > 1:
        dist.all_reduce(rays, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    rays_frame = float(rays[0].item())
    dt = float(tmax.item())

    out = None
    if rank == 0:
        ms_per_step = dt * 1e3 / args.steps
        value = rays_frame * args.steps / dt / 1e6
        out = {
            "metric": "Mrays/s (primary+secondary); 1080p frame time = ms_per_step",
            "value": round(value, 2), "unit": "Mrays/s", "n_gpus": world_size, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {WORKLOADS[args.workload][0]} "
                                   f"{world.width}x{world.height}, {world.spp} spp, depth "
                                   f"{world.camera.max_depth}, lens {world.camera.lens}",
                       "arithmetic": "fma" if args.flags & 1 else "strict (no contraction, bit-exact hits)",
                       "rays_per_frame": int(rays_frame),
                       "rays_primary": int(rays[1].item()), "rays_shadow": int(rays[2].item()),
                       "rays_bounce": int(rays[3].item()),
                       "frames_in_flight": df.mode[0], "grid_fraction_per_context": f"1/{df.mode[1]}",
                       "parallelism": (f"frames%{world_size} (whole frames per rank, no collective)" if by_frames
                                       else f"tiles%{world_size}" + ("+rccl_gather" if world_size > 1 else "")),
                       "sample_seed": SEED},
        }
        if args.rehearse_share:
            out["config"]["rehearsal_of_rank_share"] = args.rehearse_share
        if args.from_obj:
            out["config"]["scene_via_obj_loader"] = dict(LOADER)
        if animated:
            out["config"]["bvh_per_frame"] = ("linear BVH built on the GPU every frame (rt_build_scene; host sahBVH still runs in make_world, unused)"
                                              if args.device_build else
                                              "sahBVH rebuilt on the host + upload every frame" if args.rebuild_every <= 1 else
                                              f"rebuilt + uploaded every {args.rebuild_every} frames of a context, boxes refitted on the GPU in between")
        if by_frames:
            # a step = one frame on EVERY rank (world_size frames); per-GPU work is fixed as N grows
            out["scaling"] = "weak"
            out["config"]["frames_per_step"] = world_size
    if args.force_collective:
        ref = torch.zeros_like(df.frames[0])
        df.r.set_partition(0, 1)
        df.r.render_async(ref.data_ptr(), 0, args.flags, df.streams[0].cuda_stream)
        df.r.collect(df.streams[0].cuda_stream)
        same = all(torch.equal(ref, f) for f in df.frames)
        out["config"]["collective_path_frame_identical"] = bool(same)
        if not same:
            detail = []
            for i, f in enumerate(df.frames):
                bad = torch.nonzero(ref != f).flatten()
                if bad.numel():
                    lower = int((f[bad] < ref[bad]).sum())
                    zero = int((f[bad] == 0).sum())
                    detail.append(f"slot {i}: {bad.numel()} of {ref.numel()} floats differ ({lower} lower than the direct "
                                  f"frame, {zero} of them zero), first at {int(bad[0])}, last at {int(bad[-1])}")
            raise SystemExit("gathered + assembled frame differs from the directly rendered one (mode "
                             f"{df.mode}, {df.k} frames submitted): " + "; ".join(detail))
    if world_size == 1 and not args.force_collective and not args.rehearse_share:
        # Roofline of the dominant kernel.  Its launch duration is measured with HIP events
        # recorded by the library on the launch stream right around render_kernel
        # (rt_stats.ms_kernel), one frame in flight, averaged over `steps` launches -- the
        # number `rocprofv3 --kernel-trace --stats -- python bench.py --frames-in-flight 1`
        # reports as that kernel's average.  (With 2 frames in flight kernels overlap, so
        # the per-kernel duration is no longer the per-frame cost; `value` is the throughput.)
        r, stream = df.r, df.streams[0].cuda_stream
        r.set_concurrency(1)   # the whole GPU for the one frame in flight
        r.set_timing(True)     # the library's events around render_kernel
        kms, lms = [], []
        for _ in range(args.steps):
            r.render_async(df.frames[0].data_ptr(), 0, args.flags, stream)
            s1 = r.collect(stream)
            kms.append(s1.ms_kernel)
            lms.append(s1.ms_launch)
        kernel_ms = float(np.mean(kms))
        r.render_async(df.frames[0].data_ptr(), 0, args.flags | rt.FLAG_COUNT_TESTS, stream)
        cst = r.collect(stream)     # test counts under the same traversal policy
        flops = algorithmic_flops(cst)
        tf = flops / (kernel_ms * 1e-3) / 1e12
        model_gbps = rays_frame * BYTES_PER_RAY / (kernel_ms * 1e-3) / 1e9
        out["config"]["frame_latency_ms"] = round(float(np.mean(lms)), 4)
        # the reference traces more rays than the default policy launches (every reflection ray,
        # every light's shadow ray): state that count beside the one `value` is based on
        # (with the counting build, so that this untimed launch does not enter the profiler's
        #  average for the timed kernel)
        r.render_async(df.frames[0].data_ptr(), 0, args.flags | rt.FLAG_TRACE_ZERO_WEIGHT | rt.FLAG_COUNT_TESTS, stream)
        out["config"]["rays_per_frame_reference_accounting"] = int(r.collect(stream).rays_total)
        traffic, traffic_src, pmc = pmc_traffic(args.workload)
        pmc_stale = pmc_is_stale(pmc) if pmc else None
        out["roofline"] = {
            "bound": "valu_fp32", "achieved": round(tf, 3), "peak": PEAK_VALU_TFLOPS,
            "unit": "TFLOP/s", "frac": round(tf / PEAK_VALU_TFLOPS, 4), "traffic": traffic,
            "traffic_from_committed_pmc_pass": traffic_src, "pmc_stale": pmc_stale,
            "kernel": "render_kernel", "kernel_ms": round(kernel_ms, 4),
            "kernel_ms_how": "HIP events around render_kernel on its launch stream, 1 frame in flight, "
                             f"mean of {args.steps} launches",
            "algorithmic_flops_per_launch": flops,
            "algorithmic_flops_per_ray": round(flops / rays_frame, 1),
            "tests_per_launch": {"box": cst.box_tests, "sphere": cst.sphere_tests,
                                 "triangle": cst.triangle_tests, "shaded_hits": cst.shaded_hits},
            "frac_at_throughput": round(flops / (ms_per_step * 1e-3) / 1e12 / PEAK_VALU_TFLOPS, 4),
            # HBM: what the counters saw (bytes of the committed PMC pass / this run's kernel time);
            # beside it, NOT a measurement, the SURVEY 8(d) streaming model of 128 B per ray
            "hbm": {"achieved": round(traffic / (kernel_ms * 1e-3) / 1e9, 2) if traffic else None,
                    "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                    "frac": round(traffic / (kernel_ms * 1e-3) / 1e9 / PEAK_HBM_GBPS, 5) if traffic else None,
                    "bytes_per_ray_measured": round(traffic / rays_frame, 1) if traffic else None,
                    "model_bytes_per_ray": BYTES_PER_RAY, "model_gbps_if_streamed": round(model_gbps, 2)},
            "note": "no MFMA-shaped work on this path and the scene is on-chip: the binding bound is "
                    "fp32 VALU issue (PMC: profiles/), not HBM; hbm.achieved is counter traffic over "
                    "kernel time; model_* price the ray-queue streaming model of SURVEY 8(d)",
        }
        if "SQ_INSTS_VALU" in pmc:
            # The bound that binds ("What the measurements say"): VALU ISSUE.  Wave64
            # instructions per launch from the committed counter pass over this run's kernel time,
            # against 1024 SIMDs x 2.4 GHz / 2 cycles -- the rate of full-rate two-operand fp32, which
            # is what the 157.3 TFLOP/s peak assumes.  Three-operand, compare/select and 64-bit
            # encodings take ~4 cycles (tools/micro/mix_rate.hip): this kernel's mix can issue at
            # ~3.1 cycles per instruction at best, i.e. frac ~0.65 is its ceiling.
            wi = pmc["SQ_INSTS_VALU"]
            rate = wi / (kernel_ms * 1e-3) / 1e9
            out["roofline"]["valu_issue"] = {
                "achieved": round(rate, 1), "peak": PEAK_VALU_ISSUE, "unit": "G wave64 instructions/s",
                "frac": round(rate / PEAK_VALU_ISSUE, 4),
                "wave_instructions_per_launch": wi,
                "cycles_per_instruction_per_simd": round(1024 * 2.4e9 * kernel_ms * 1e-3 / wi, 2),
                "lanes_active": round(pmc["SQ_THREAD_CYCLES_VALU"] / (wi * 64), 3)
                if "SQ_THREAD_CYCLES_VALU" in pmc else None,
                "from_committed_pmc_pass": traffic_src}
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(world, grids)
    if rank == 0:
        print(json.dumps(out), flush=True)
    df.close()
    if world_size > 1 or args.force_collective:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
