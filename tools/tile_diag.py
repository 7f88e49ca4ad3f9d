#!/usr/bin/env python3
"""Per-tile / per-wave time diagnostics of one frame (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import rt_amd as rt
NR = int(os.environ.get("NRANKS", "1"))
for wl in sys.argv[1:] or ["c2"]:
    world = bench.make_world(wl)
    grids = rt.generate_grids(bench.SEED, world.width + 10, world.camera.aa)
    r = rt.Renderer(0)
    r.set_world(world); r.set_samples(grids); r.set_partition(0, NR)
    for i in range(3):
        rgb, _, st = r.render(0)
    til = r.tiling()
    print(wl, f"kernel {st.ms_kernel:.3f} ms  tiles {til.n_tiles} ({til.tile_w}x{til.tile_h})  "
          f"waves {st.n_waves}  wave avg {st.wave_ticks_sum/max(1,st.n_waves)/100:.1f} us  "
          f"wave max {st.wave_ticks_max/100:.1f} us "
          f"rays {st.rays_total}")
    wd = r.wave_diag(int(st.n_waves)).astype(float)
    import numpy as np
    life = wd[:, 0] / 100.0
    pc = np.percentile(life, [0, 10, 50, 90, 99, 100])
    top = np.argsort(life)[-3:]
    print("   wave lifetime us percentiles 0/10/50/90/99/100:", " ".join(f"{x:.0f}" for x in pc),
          " rays/wave min/med/max:", int(wd[:, 1].min()), int(np.median(wd[:, 1])), int(wd[:, 1].max()),
          " slowest waves rays:", [int(wd[i, 1]) for i in top])
    r.close()
    t0 = wd[:, 2] - wd[:, 2].min()
    dry = (t0 + wd[:, 3]) / 100.0
    end = (t0 + wd[:, 0]) / 100.0
    print("   start us p50/p100:", f"{np.percentile(t0/100,50):.0f} {t0.max()/100:.0f}",
          " dry us p0/p50/p100:", " ".join(f"{x:.0f}" for x in np.percentile(dry, [0, 50, 100])),
          " end us p0/p50/p99/p100:", " ".join(f"{x:.0f}" for x in np.percentile(end, [0, 50, 99, 100])),
          " drain (end-dry) us p50/p90/p100:", " ".join(f"{x:.0f}" for x in np.percentile(end - dry, [50, 90, 100])))
