#!/bin/bash
# Quick throughput + single-frame numbers of some workloads (run through gpurun).
#   tools/perf_quick.sh <tag> [workload ...]      (RT_MI355X_LIB selects a variant library)
set -u
TAG=${1:-perf}; shift || true
WL=${@:-"c2 c3"}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
for w in $WL; do
  timeout -k 10 200 python bench.py --workload $w --no-cpu-baseline --steps 30 --warmup 5 > "$OUT/$w.json" 2> "$OUT/$w.err" || { echo "bench $w FAILED"; tail -5 "$OUT/$w.err"; exit 1; }
  python - "$OUT/$w.json" <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=j.get("roofline",{})
print(f'{j["config"]["workload"][:28]:28s} {j["ms_per_step"]:8.3f} ms/frame  {j["value"]/1e3:7.2f} Grays/s  mode {j["config"]["frames_in_flight"]},{j["config"]["grid_fraction_per_context"]}  kernel {r.get("kernel_ms")} ms  frac {r.get("frac")}  boxes {r.get("tests_per_launch",{}).get("box")}')
PY
done
