#!/usr/bin/env python3
"""Static VALU / SALU / LDS / VMEM instruction counts between the phase markers of render_kernel
(build with -DRT_ASM_MARKS -S).  usage: tools/phase_static.py [kernel-substring]
Loops are counted once; the traversal phases are loop bodies (dynamic counts: tile_diag.py)."""
import re, subprocess, sys, collections
kern = sys.argv[1] if len(sys.argv) > 1 else "render_kernelILb1ELb0ELb1ELb0ELi4ELi5E"
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fno-slp-vectorize", "-std=c++17", "-fPIC", "-Iinclude", "-Iray-tracer_amd/csrc",
       "-ffp-contract=off", "-DRT_ASM_MARKS", "-S", "--cuda-device-only", "ray-tracer_amd/csrc/rt_kernels_strict.hip", "-o", "/tmp/marks.s"]
subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
names = {0: "obtain rays (pop / decode)", 1: "closest traversal", 2: "shade (direct)", 3: "shadow traversal", 4: "accumulate",
         5: "emit bounces", 6: "level select", 7: "packet decode", 8: "samples+getRay", 9: "fetch next packet"}
inside = False
cur = "prologue"
cnt = collections.defaultdict(lambda: collections.Counter())
for line in open("/tmp/marks.s"):
    if line.startswith("_ZN") and kern in line and line.rstrip().endswith(":") or (kern in line and line.startswith("_ZN") and ": " in line):
        inside = True
        continue
    if not inside:
        continue
    t = line.strip()
    if t.startswith("s_endpgm"):
        break
    m = re.match(r"; PHASE_MARK (\d+) line (\d+)", t)
    if m:
        cur = "after mark %s at line %s" % (m.group(1), m.group(2))
        continue
    if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
        continue
    op = t.split()[0]
    kind = "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else "vmem"
    cnt[cur][kind] += 1
    if op.startswith(("v_div_", "v_rcp", "v_sqrt", "v_rsq")):
        cnt[cur]["div/trans"] += 1
for k, v in cnt.items():
    print(f"{k:60s} " + "  ".join(f"{a} {b}" for a, b in sorted(v.items())))
