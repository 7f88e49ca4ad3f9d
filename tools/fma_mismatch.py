#!/usr/bin/env python3
"""RT_FLAG_FMA (contracted arithmetic) against the oracle (strict, one rounding per operation):
the rate of primary hit-id mismatches BASELINE.json's north_star asks to have reported, and what
they do to the 8-bit frame.  Run on the GPU box; output kept under profiles/."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import rt_amd as rt
from oracle import binding as ob
from conftest import make_world

def psnr8(a, b):
    d = a.astype(np.float64) - b.astype(np.float64)
    mse = float(np.mean(d * d))
    return 99.0 if mse == 0 else 10 * np.log10(255.0 ** 2 / mse)

r = rt.Renderer(0)
print("PARITY UNPINNED: the oracle is a restatement of the reference (no GHC here).")
print(f"{'scene':12s} {'size':9s} {'spp':>3s} | {'strict: id mismatches':>22s} | {'FMA: id mismatches':>28s} {'8-bit max':>9s} {'px > 1':>8s} {'PSNR dB':>8s} {'ray count delta':>15s}")
for name, rpp, lens in (("bench1", 16, None), ("bench2", 4, None), ("bench3", 4, None), ("bench4", 16, 0.05), ("bench5", 16, None), ("mesh:3000", 4, None), ("mesh:100000", 4, None)):
    W, H = 480, 270
    w = make_world(rt, name, W, H, rpp, view=(16, 9, 10), lens=lens)
    g = rt.generate_grids(5, W + 10, w.camera.aa)
    r.set_world(w); r.set_samples(g)
    o_rgb, o_ids, o_st = ob.render(w, g, ob.FLAG_CULL_ZERO_WEIGHT)
    s_rgb, s_ids, s_st = r.render(0, want_hit_ids=True)
    f_rgb, f_ids, f_st = r.render(rt.FLAG_FMA, want_hit_ids=True)
    o8, f8 = rt.quantize(o_rgb), rt.quantize(f_rgb)
    d = np.abs(f8.astype(int) - o8.astype(int))
    ms, mf = int((s_ids != o_ids).sum()), int((f_ids != o_ids).sum())
    print(f"{name:12s} {W}x{H:<5d} {w.spp:3d} | {ms:10d} / {o_ids.size:<9d} | {mf:8d} / {o_ids.size:<9d} = {100.0*mf/o_ids.size:.5f} % {d.max():9d} {int((d.max(axis=1) > 1).sum()):8d} {psnr8(f8, o8):8.1f} {f_st.rays_total - o_st.rays_total:15d}")
r.close()
