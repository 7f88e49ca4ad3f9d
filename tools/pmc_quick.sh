#!/bin/bash
# One PMC pass (instruction counts) for a workload, optionally with a variant library:
#   tools/pmc_quick.sh <tag> <workload> [lib-suffix]
set -u
TAG=$1; W=$2; V=${3:-}
[ -n "$V" ] && export RT_MI355X_LIB="$GRAFT_REPO_ROOT/ray-tracer_amd/lib/librt_mi355x$V.so"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG/pmc_$W$V
rm -rf "$OUT"; mkdir -p "$OUT"
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 --output-format csv -d "$OUT/p" -- python3 bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline --frames-in-flight 1 ${PMC_ARGS:-} > "$OUT/log" 2>&1 || { echo "pmc failed"; tail -5 "$OUT/log"; exit 1; }
echo "== $W $V"; python3 tools/pmc_summary.py "$OUT"
