#!/bin/bash
# Extra PMC passes (issue stalls, scalar/branch time, instruction cache) for render_kernel:
#   tools/prof_pmc2.sh <outdir> [bench args...]
set -u
OUT=${1:-gpurun_out/pmc2}; shift || true
ARGS=${@:-"--steps 2 --warmup 1 --no-cpu-baseline --frames-in-flight 1"}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
pass() {
  name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 bench.py $ARGS > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -5 "$OUT/$name.log"; return 1; }
}
pass sq4 SQ_WAVE_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_IFETCH SQ_INSTS_BRANCH &&
pass sq5 SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_INSTS_SALU SQ_BUSY_CU_CYCLES
echo "pmc2 done"
