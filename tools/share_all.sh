#!/bin/bash
# One-GPU rehearsal of every rank's share of an 8-rank job (bench.py --rehearse-share r/8): what
# bounds strong scaling before the gather.  Two columns per rank: frames in flight (the pipelined
# host, bench.py's default) and ONE frame at a time (what a blocking rt_multi_render gives).
#   tools/share_all.sh <out file> [workload ...]
set -u
OUT=${1:-gpurun_out/share8.txt}; shift || true
WL=${@:-"c2 c3 c4-glossy c4-dof"}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
: > "$OUT"
for w in $WL; do
  for r in 0 1 2 3 4 5 6 7; do
    timeout -k 10 120 python bench.py --workload $w --no-cpu-baseline --steps 30 --warmup 5 --rehearse-share $r/8 > /tmp/share.json 2> /tmp/share.err || { echo "share $w $r failed"; tail -3 /tmp/share.err; exit 1; }
    timeout -k 10 120 python bench.py --workload $w --no-cpu-baseline --steps 30 --warmup 5 --rehearse-share $r/8 --frames-in-flight 1 > /tmp/share1.json 2> /tmp/share.err || { echo "share $w $r (1 in flight) failed"; tail -3 /tmp/share.err; exit 1; }
    python - "$w" "$r" >> "$OUT" <<'PY'
import json,sys
j=json.loads(open("/tmp/share.json").read().strip().splitlines()[-1])
k=json.loads(open("/tmp/share1.json").read().strip().splitlines()[-1])
c=j["config"]
print(f'{sys.argv[1]:10s} rank {sys.argv[2]}/8  {j["ms_per_step"]:.4f} ms/frame pipelined (mode {c["frames_in_flight"]},{c["grid_fraction_per_context"]})  {k["ms_per_step"]:.4f} ms/frame one at a time  rays {c["rays_per_frame"]}')
PY
  done
done
cat "$OUT"
