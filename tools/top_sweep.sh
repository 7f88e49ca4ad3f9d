#!/bin/bash
# c3 (scene in memory) under the TOP builds' knobs (run through gpurun): tools/top_sweep.sh <tag> [workload]
set -u
TAG=${1:-top}; W=${2:-c3}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in "RT_TOP=0" "RT_TOP=16" "RT_TOP=8" "RT_TOP=16 RT_STACK_CAP=8" "RT_TOP=16 RT_STACK_CAP=16" "RT_TOP=16 RT_LDS_TOP=64" "RT_TOP=16 RT_LDS_TOP=256"; do
  echo "== $v"
  env $v bash tools/perf_quick.sh "${TAG}_$(echo $v | tr ' =' '__')" $W || exit 1
done
