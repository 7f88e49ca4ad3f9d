#!/bin/bash
# tools/ab_libs.sh "tag1 tag2 ..." workload...  -- share_time per library variant ("" = the shipped one)
TAGS=$1; shift
for t in $TAGS; do
  echo "== lib ${t}"
  if [ "$t" = "base" ]; then unset RT_MI355X_LIB; else export RT_MI355X_LIB=$GRAFT_REPO_ROOT/ray-tracer_amd/lib/librt_mi355x_${t}.so; fi
  timeout -k 10 200 python -u tools/share_time.py "$@" || exit 1
done
