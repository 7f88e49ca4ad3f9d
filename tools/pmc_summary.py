#!/usr/bin/env python3
"""Summarise tools/prof_pmc.sh output: per-counter average over the dispatches of the
render kernel (non-counting variant), stamped with the SHA-256 of the sources the library
was built from."""
import csv
import glob
import hashlib
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_sha256():
    """What the counters belong to: SHA-256 over the kernel and host sources of the library
    (bench.py puts "pmc_stale": true in its line when the tree's sources differ from this stamp)."""
    h = hashlib.sha256()
    # (build.py: the compiler flags belong to what was measured)
    for f in ("ray-tracer_amd/csrc/rt_kernels.inl", "ray-tracer_amd/csrc/rt_api.hip", "ray-tracer_amd/csrc/rt_device.h",
              "ray-tracer_amd/build.py"):
        h.update(open(os.path.join(ROOT, f), "rb").read())
    return h.hexdigest()


def main():
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
    if acc:
        print(f"# source_sha256 {source_sha256()}  (rt_kernels.inl + rt_api.hip + rt_device.h + build.py at the time of the passes)")
    for name in sorted(acc):
        v = acc[name]
        print(f"{name:40s} n={len(v):3d} avg={sum(v)/len(v):.6g}")


if __name__ == "__main__":
    main()
