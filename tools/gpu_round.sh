#!/bin/bash
# One GPU session: parity tests, bench line, rocprofv3 kernel stats (run through gpurun).
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1 || { echo "gpu tests FAILED"; tail -40 gpurun_out/gpu_tests.log; exit 1; }
tail -3 gpurun_out/gpu_tests.log
timeout -k 10 300 python bench.py > gpurun_out/bench.log 2>&1 || { echo "bench FAILED"; tail -20 gpurun_out/bench.log; exit 1; }
tail -1 gpurun_out/bench.log
rm -rf gpurun_out/prof_stats
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --frames-in-flight 1 > gpurun_out/prof_stats.log 2>&1 || { echo "rocprof FAILED"; tail -20 gpurun_out/prof_stats.log; exit 1; }
find gpurun_out/prof_stats -name '*kernel_stats.csv' -exec cat {} \;
