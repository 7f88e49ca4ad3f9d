#!/bin/bash
# A/B of an environment knob on the GPU box: tools/ab.sh VAR "v1 v2" "workloads" [flags]
cd "$GRAFT_REPO_ROOT"
VAR=$1; VALS=$2; WLS=${3:-c2}; FL=${4:-0}
for wl in $WLS; do for v in $VALS; do
  line=$(env $VAR=$v timeout -k 10 200 python bench.py --workload $wl --flags $FL --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1) || { echo "FAILED $wl $VAR=$v"; exit 1; }
  echo "$wl $VAR=$v $(echo "$line" | python -c 'import sys,json; j=json.loads(sys.stdin.read()); print(j["ms_per_step"], "ms", j["value"], "Mrays/s")')"
done; done
