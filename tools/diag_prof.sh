#!/bin/bash
# Per-phase / stack-depth / top-of-tree diagnostics with the profiling build (run through gpurun):
#   tools/diag_prof.sh <out file> [workload ...]      needs ray-tracer_amd/lib/librt_mi355x_prof.so
#   (python ray-tracer_amd/build.py --tag=_prof -DRT_PROFILE_PHASES)
set -u
OUT=${1:-gpurun_out/diag_prof.txt}; shift || true
WL=${@:-"c2 c3"}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export RT_MI355X_LIB="$GRAFT_REPO_ROOT/ray-tracer_amd/lib/librt_mi355x_prof.so"
timeout -k 10 300 python tools/tile_diag.py $WL > "$OUT" 2> "$OUT.err" || { echo "tile_diag failed"; tail -5 "$OUT.err"; exit 1; }
cat "$OUT"
