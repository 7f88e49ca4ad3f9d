#!/usr/bin/env python3
"""Time line of single scheduling steps (GPU box, the -DRT_TRACE_STEPS build):
    python ray-tracer_amd/build.py --tag=_trace -DRT_TRACE_STEPS
    RT_MI355X_LIB=.../librt_mi355x_trace.so python tools/step_trace.py [workload] [nranks]
One frame alone on the GPU; for the traced wave with the most steps, every step's phases in
microseconds (wall clock, 10 ns resolution), then the median per phase over its bounce steps."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
import rt_amd as rt

WAVES, STEPS, MARKS = 64, 48, 12
wl = sys.argv[1] if len(sys.argv) > 1 else "c1"
nr = int(sys.argv[2]) if len(sys.argv) > 2 else 1
world = bench.make_world(wl)
grids = rt.generate_grids(bench.SEED, world.width + 10, world.camera.aa)
r = rt.Renderer(0)
r.set_world(world); r.set_samples(grids); r.set_partition(0, nr)
for _ in range(3):
    _, _, st = r.render(0)
raw = r.wave_diag(16384 + (160 + WAVES * STEPS * MARKS) // 4).reshape(-1)
tr = raw[(16384 * 4 + 160):].astype(np.int64).reshape(WAVES, STEPS, MARKS)
print(f"{wl} (share 1/{nr}): render_kernel {st.ms_kernel:.3f} ms, {st.n_waves} waves, rays {st.rays_total}")
# order of the marks inside a step, and what ends at each
order = [(6, "choose level"), (7, "decode packet"), (8, "samples + getRay"), (9, "fetch next packet"), (0, "rays obtained (pop)"),
         (1, "closest-hit traversal"), (2, "normal, light ray"), (3, "shadow traversal"), (10, "lit term"), (4, "pixel sums + atomics"),
         (5, "emit bounces (queue stores)")]
n_steps = (tr[:, :, 5] != 0).sum(axis=1)
w = int(np.argmax(n_steps))
print(f"traced wave {w}: {n_steps[w]} steps (most among the first {WAVES} waves; median {int(np.median(n_steps))})")
rows = []
prev_end = None
for s in range(n_steps[w]):
    t = tr[w, s]
    marks = [(i, name, t[i]) for i, name in order if t[i] != 0]
    start = prev_end if prev_end is not None else marks[0][2]
    durs = {}
    last = start
    for i, name, v in marks:
        durs[name] = (v - last) / 100.0
        last = v
    primary = t[7] != 0
    rows.append((s, primary, (last - start) / 100.0, durs))
    prev_end = last
names = [n for _, n in order]
print("step kind     total us | " + " | ".join(f"{n[:14]:>14s}" for n in names))
for s, primary, tot, d in rows:
    print(f"{s:4d} {'primary' if primary else 'bounce ':7s} {tot:9.2f} | " + " | ".join(f"{d.get(n, float('nan')):14.2f}" for n in names))
b = [d for _, p, _, d in rows if not p]
if b:
    print("median over the bounce steps: " + ", ".join(f"{n} {np.median([d[n] for d in b if n in d]):.2f}" for n in names if any(n in d for d in b))
          + f"; step total {np.median([t for _, p, t, _ in rows if not p]):.2f} us")
r.close()
