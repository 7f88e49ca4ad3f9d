#!/usr/bin/env python3
"""Per-tile / per-wave time diagnostics of one frame (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import numpy as np
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
    wdi = r.wave_diag(int(st.n_waves))
    steps = (wdi[:, 2] >> np.uint64(40)).astype(float)
    wdi[:, 2] &= np.uint64((1 << 40) - 1)
    given = ((wdi[:, 1] >> np.uint64(40)) & np.uint64(0xfff)).astype(int)
    taken = (wdi[:, 1] >> np.uint64(52)).astype(int)
    wdi[:, 1] &= np.uint64((1 << 40) - 1)
    print(f'   work sharing: chunks given {given.sum()} taken {taken.sum()} (waves giving {np.count_nonzero(given)}, taking {np.count_nonzero(taken)})')
    wd = wdi.astype(float)
    life = wd[:, 0] / 100.0
    pc = np.percentile(life, [0, 10, 50, 90, 99, 100])
    top = np.argsort(life)[-3:]
    print("   wave lifetime us percentiles 0/10/50/90/99/100:", " ".join(f"{x:.0f}" for x in pc),
          " rays/wave min/med/max:", int(wd[:, 1].min()), int(np.median(wd[:, 1])), int(wd[:, 1].max()),
          " slowest waves rays:", [int(wd[i, 1]) for i in top])
    t0 = wd[:, 2] - wd[:, 2].min()
    dry = (t0 + wd[:, 3]) / 100.0
    end = (t0 + wd[:, 0]) / 100.0
    print("   start us p50/p100:", f"{np.percentile(t0/100,50):.0f} {t0.max()/100:.0f}",
          " dry us p0/p50/p100:", " ".join(f"{x:.0f}" for x in np.percentile(dry, [0, 50, 100])),
          " end us p0/p50/p99/p100:", " ".join(f"{x:.0f}" for x in np.percentile(end, [0, 50, 99, 100])),
          " drain (end-dry) us p50/p90/p100:", " ".join(f"{x:.0f}" for x in np.percentile(end - dry, [50, 90, 100])))
    slow = np.argsort(end)[-6:]
    print('   slowest waves (rays, steps, dry us, end us):', [(int(wd[i,1]), int(steps[i]), int(dry[i]), int(end[i])) for i in slow],
          ' median wave (rays, steps):', int(np.median(wd[:,1])), int(np.median(steps)))
    ph = r.wave_diag(16384 + 40)[16384:].reshape(-1).astype(float)
    if ph[:6].sum() > 0:
        names = ["pop/other", "closest traversal", "shade (direct)", "shadow traversal", "accumulate", "emit bounces",
                 "level select", "packet decode", "samples+getRay", "fetch next packet", "-", "-"]
        print(f"   first step: wave start -> rays generated {ph[10]/st.n_waves:.0f} cycles avg, -> step done {ph[11]/st.n_waves:.0f} cycles avg")
        ph[10] = ph[11] = 0
        tot = ph[:12].sum()
        print("   phases (share of summed wave cycles): " + "  ".join(f"{n} {100*v/tot:.1f}%" for n, v in zip(names, ph[:12]) if v))
        for nm, i in (("closest node", 0), ("closest leaf", 1), ("shadow node", 2), ("shadow leaf", 3)):
            lane_it, wave_it = ph[16 + i], ph[20 + i] / 2.0**20
            print(f"   {nm} iterations: {wave_it/max(1,ph[12]):.1f} per scheduling step, {lane_it/max(1,wave_it):.1f} lanes in each, "
                  f"{ph[24 + i]/max(1,wave_it):.0f} cycles of wave time each ({100*ph[24 + i]/tot:.1f}% of all wave cycles)")
        wh = ph[120:129]
        if wh[0] > 0:
            print(f"   shadow-ray pool, lanes per loop iteration ({int(wh[0])} iterations): at a node {wh[1]/wh[0]:.1f}, at a leaf {wh[2]/wh[0]:.1f}, "
                  f"idle waiting for a re-arm {wh[3]/wh[0]:.1f}, idle with the pool handed out (tail) {wh[4]/wh[0]:.1f}")
        if wh[5] > 0:
            print(f"   closest-hit traversal (scene in memory), lanes per loop iteration ({int(wh[5])} iterations): at a node {wh[6]/wh[5]:.1f}, "
                  f"at a leaf {wh[7]/wh[5]:.1f}, finished or never active {wh[8]/wh[5]:.1f}")
        if ph[29] > 0:
            print(f"   shadow-ray pool re-arms: {ph[29]/max(1,ph[12]):.1f} per scheduling step, {ph[30]/ph[29]:.1f} rays handed out in each, "
                  f"{ph[28]/ph[29]:.0f} cycles of wave time each ({100*ph[28]/tot:.1f}% of all wave cycles)")
        for nm, off in (("closest-hit traversals", 32), ("shadow rays of the pool / single-light shadow traversals", 72)):
            h = ph[off:off + 40]
            if h.sum() > 0:
                c = np.cumsum(h) / h.sum()
                print(f"   deepest stack per traversal, {nm} ({int(h.sum())}): " +
                      " ".join(f"<={d}:{100*c[d]:.3f}%" for d in (2, 4, 6, 8, 10, 12, 14, 16, 20, 24, 28) if d < 40) +
                      f"  max {int(np.nonzero(h)[0].max())}")
        tk = ph[112:120]
        if tk[7] > 0:
            print("   wide-node visits with record index below: " + " ".join(f"{64 << k}:{100*tk[k]/tk[7]:.1f}%" for k in range(7)) +
                  f"  (of {int(tk[7])} lane visits)")
        print(f"   steps {int(ph[12])}  avg active lanes/step {ph[13]/max(1,ph[12]):.1f}  cycles/step/wave {tot/max(1,ph[12]):.0f}")
    r.close()
