#!/usr/bin/env python3
"""Summarise tools/prof_pmc.sh output: per-counter average over the dispatches of the
render kernel (non-counting variant)."""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc"
pat = sys.argv[2] if len(sys.argv) > 2 else "render_kernel"
acc = defaultdict(list)
for f in sorted(glob.glob(root + "/*/**/*counter_collection.csv", recursive=True)):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row.get("Kernel_Name", "")
            args = k.split(pat + "<")[-1].split(">")[0].split(",") if pat + "<" in k else []
            counting = len(args) > 1 and args[1].strip() == "true"   # render_kernel<PRUNE, COUNT, LDS>
            if pat in k and not counting:
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for name in sorted(acc):
    v = acc[name]
    print(f"{name:40s} n={len(v):3d} avg={sum(v)/len(v):.6g}")
