#!/usr/bin/env python3
"""Quick A/B of device-library variants (run on the GPU box):
   python tools/quick.py "<lib1>,<lib2>" "<workload1>,<workload2>" "<flags1>,<flags2>" [steps]
Each (lib, workload, flags) runs bench.py in a child process; prints ms/frame and Mrays/s."""
import json
import os
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = sys.argv[1].split(",")
wls = sys.argv[2].split(",") if len(sys.argv) > 2 else ["c2"]
flags = sys.argv[3].split(",") if len(sys.argv) > 3 else ["0"]
steps = sys.argv[4] if len(sys.argv) > 4 else "5"
for wl in wls:
    for fl in flags:
        for lib in libs:
            env = dict(os.environ)
            if lib and lib != "default":
                env["RT_MI355X_LIB"] = os.path.join(root, "ray-tracer_amd", "lib", f"librt_mi355x{lib}.so")
            p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", wl, "--flags", fl,
                                "--steps", steps, "--warmup", "2", "--no-cpu-baseline"],
                               env=env, capture_output=True, text=True, timeout=300)
            line = [l for l in p.stdout.splitlines() if l.startswith("{")]
            if not line:
                print(f"{wl:10s} flags={fl} lib={lib:10s} FAILED rc={p.returncode} {p.stderr[-300:]}", flush=True)
                continue
            j = json.loads(line[-1])
            print(f"{wl:10s} flags={fl} lib={lib:10s} {j['ms_per_step']:9.3f} ms  {j['value']:9.1f} Mrays/s  "
                  f"rays={j['config']['rays_per_frame']}", flush=True)
