#!/bin/bash
# PMC passes for the trace/shade kernel (run on the GPU box through gpurun):
#   tools/prof_pmc.sh <outdir> [bench args...]
# One rocprofv3 run per pass (counters do not split over passes; see the environment notes).
set -u
OUT=${1:-gpurun_out/pmc}; shift || true
ARGS=${@:-"--steps 2 --warmup 1 --no-cpu-baseline --frames-in-flight 1"}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
pass() {
  name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 bench.py $ARGS > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -5 "$OUT/$name.log"; return 1; }
}
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY &&
pass sq2 SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM &&
pass sq3 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE &&
pass tcc1 FETCH_SIZE TCC_REQ_sum GRBM_GUI_ACTIVE GRBM_TA_BUSY &&
pass tcc2 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum &&
pass tcp1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum
echo "pmc done"
