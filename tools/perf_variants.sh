#!/bin/bash
# Throughput of several variant libraries side by side (run through gpurun).
#   tools/perf_variants.sh <tag> "<workloads>" <lib-suffix|env=VALUE> ...     ("" = the shipped library)
# e.g. tools/perf_variants.sh v1 "c2 c3" "" _sm RT_FIVE_WAVES=0
set -u
TAG=$1; WL=$2; shift 2
for v in "$@"; do
  unset RT_MI355X_LIB RT_FIVE_WAVES
  case "$v" in
    *=*) export "$v"; name="$v" ;;
    "") name="shipped" ;;
    *) export RT_MI355X_LIB="$GRAFT_REPO_ROOT/ray-tracer_amd/lib/librt_mi355x$v.so"; name="$v" ;;
  esac
  echo "== $name"
  bash tools/perf_quick.sh "$TAG$name" $WL || exit 1
done
