#!/bin/bash
# tools/sweep_env.sh VAR "v1 v2 ..." workload...   -- share_time per value of an env knob (GPU box)
VAR=$1; VALS=$2; shift 2
for v in $VALS; do
  echo "== $VAR=$v"
  env $VAR=$v timeout -k 10 200 python -u tools/share_time.py "$@" || exit 1
done
