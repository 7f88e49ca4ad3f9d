#!/bin/bash
# GPU parity suite + one default bench line (run through gpurun).  Output under gpurun_out/<tag>/.
set -u
TAG=${1:-run}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
timeout -k 10 780 python -u -m pytest tests -m gpu -x -v --timeout 170 --timeout-method thread > "$OUT/gpu_tests.log" 2>&1
rc=$?
tail -5 "$OUT/gpu_tests.log"
if [ $rc -ne 0 ]; then echo "gpu tests FAILED rc=$rc"; grep -n "FAILED\|Error\|error" "$OUT/gpu_tests.log" | tail -20; exit 1; fi
timeout -k 10 300 python bench.py > "$OUT/bench_c2.json" 2> "$OUT/bench_c2.err" || { echo "bench FAILED"; tail -20 "$OUT/bench_c2.err"; exit 1; }
cat "$OUT/bench_c2.json"
