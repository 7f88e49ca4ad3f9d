#!/bin/bash
# Two PMC passes (instruction mix; L1 / TA activity) of one workload (run through gpurun):
#   tools/pmc_mem.sh <tag> <workload>     (environment selects library / knobs)
set -u
TAG=$1; W=$2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG/pmc_$W
rm -rf "$OUT"; mkdir -p "$OUT"
ARGS="--workload $W --steps 2 --warmup 1 --no-cpu-baseline --frames-in-flight 1"
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d "$OUT/p1" -- python3 bench.py $ARGS > "$OUT/log1" 2>&1 || { echo "pmc 1 failed"; tail -5 "$OUT/log1"; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum GRBM_GUI_ACTIVE GRBM_TA_BUSY --output-format csv -d "$OUT/p2" -- python3 bench.py $ARGS > "$OUT/log2" 2>&1 || { echo "pmc 2 failed"; tail -5 "$OUT/log2"; exit 1; }
python3 tools/pmc_summary.py "$OUT"
