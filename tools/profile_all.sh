#!/bin/bash
# Everything the roofline numbers are recomputed from, for every workload of bench.py (run through
# gpurun):  tools/profile_all.sh <round tag, e.g. r02> [workload ...]
#   profiles/<tag>_<w>_bench.json         the bench line (default mode, with cpu_baseline for c2)
#   profiles/<tag>_<w>_kernel_stats.csv   rocprofv3 --kernel-trace --stats, one frame in flight
#   profiles/<tag>_<w>_pmc.txt            PMC passes (separate runs), averaged per render_kernel launch
#   profiles/<tag>_c2_trace_inflight.csv  kernel trace of the DEFAULT (frames in flight) mode of c2
set -u
TAG=${1:-r02}; shift || true
WL=${@:-"c2 c1 c3 c4-glossy c4-dof c5"}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
S=gpurun_out/profile_$TAG
mkdir -p "$S" profiles
for w in $WL; do
  echo "== $w"
  extra="--no-cpu-baseline"; [ "$w" = c2 ] && extra=""
  rm -rf "$S/${w}_stats"
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$S/${w}_stats" -- python3 bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline --frames-in-flight 1 > "$S/${w}_stats.log" 2>&1 || { echo "rocprof stats $w failed"; tail -5 "$S/${w}_stats.log"; exit 1; }
  find "$S/${w}_stats" -name '*kernel_stats.csv' -exec cp {} "$S/${w}_kernel_stats.csv" \;
  rm -rf "$S/pmc_$w"
  bash tools/prof_pmc.sh "$S/pmc_$w" --workload $w --steps 2 --warmup 1 --no-cpu-baseline --frames-in-flight 1 > "$S/${w}_pmc.log" 2>&1 || { echo "pmc $w failed"; tail -5 "$S/${w}_pmc.log"; exit 1; }
  bash tools/prof_pmc2.sh "$S/pmc_$w" --workload $w --steps 2 --warmup 1 --no-cpu-baseline --frames-in-flight 1 >> "$S/${w}_pmc.log" 2>&1 || { echo "pmc2 $w failed"; tail -5 "$S/${w}_pmc.log"; exit 1; }
  python3 tools/pmc_summary.py "$S/pmc_$w" > "$S/${w}_pmc.txt"
  # the bench line last: its roofline.traffic is read from THIS run's counter pass
  cp "$S/${w}_pmc.txt" "profiles/${TAG}_${w}_pmc.txt"
  timeout -k 10 300 python bench.py --workload $w $extra > "$S/${w}_bench.json" 2> "$S/${w}_bench.err" || { echo "bench $w failed"; tail -5 "$S/${w}_bench.err"; exit 1; }
  head -c 300 "$S/${w}_bench.json"; echo
done
if echo "$WL" | grep -qw c2; then
  rm -rf "$S/c2_trace"
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d "$S/c2_trace" -- python3 bench.py --workload c2 --steps 10 --warmup 2 --no-cpu-baseline > "$S/c2_trace.log" 2>&1 || { echo "trace failed"; tail -5 "$S/c2_trace.log"; exit 1; }
  find "$S/c2_trace" -name '*kernel_trace.csv' -exec cp {} "$S/c2_trace_inflight.csv" \;
fi
echo "profile_all done"
