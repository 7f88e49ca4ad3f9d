#!/bin/bash
# 1/8-share rehearsal of a few ranks for variant libraries:  tools/share_quick.sh "<workloads>" <lib-suffix> ...
set -u
WL=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in "$@"; do
  unset RT_MI355X_LIB
  [ -n "$v" ] && export RT_MI355X_LIB="$GRAFT_REPO_ROOT/ray-tracer_amd/lib/librt_mi355x$v.so"
  echo "== ${v:-shipped}"
  for w in $WL; do
    for r in 0 3; do
      timeout -k 10 120 python bench.py --workload $w --no-cpu-baseline --steps 30 --warmup 5 --rehearse-share $r/8 > /tmp/share.json 2> /tmp/share.err || { echo "share $w $r failed"; tail -3 /tmp/share.err; exit 1; }
      python - "$w" "$r" <<'PY'
import json,sys
j=json.loads(open("/tmp/share.json").read().strip().splitlines()[-1])
c=j["config"]
print(f'{sys.argv[1]:10s} rank {sys.argv[2]}/8  {j["ms_per_step"]:.4f} ms/frame  latency {c.get("frame_latency_ms")}  mode {c["frames_in_flight"]},{c["grid_fraction_per_context"]}')
PY
    done
  done
done
