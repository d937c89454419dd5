#!/bin/bash
# One bench line + rocprofv3 kernel statistics per BASELINE configuration (the same process gives both), then the
# hardware-counter passes of the headline workload.  Usage: tools/prof_round2.sh <outdir> [workloads...]
set -u
OUT=${1:-gpurun_out/r2}; shift || true
WL=${@:-sponza bunny spheres cornell lucy}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p "$OUT"
for W in $WL; do
  timeout -k 10 420 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_$W" -- python3 bench.py --workload $W --steps 10 --warmup 3 --cpu-budget 9 > "$OUT/bench_$W.json" 2> "$OUT/bench_$W.err"
  rc=$?
  echo "bench $W rc=$rc $(cut -c1-160 "$OUT/bench_$W.json")"
  [ $rc -eq 0 ] || exit 1
  find "$OUT/stats_$W" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats_$W.csv" \;
done
