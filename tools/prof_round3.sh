#!/bin/bash
# One bench line + rocprofv3 kernel statistics per configuration (the same process gives both): the five BASELINE
# configurations, the textured instance (the headline geometry with images) and the light-sampling instance (the
# headline scene with rt_params.light_mode 1); one frame in flight only, so that the kernel statistics are those of
# launches that have the chip to themselves.  Usage: tools/prof_round3.sh <outdir> [names...]
set -u
OUT=${1:-gpurun_out/r3}; shift || true
WL=${@:-sponza bunny spheres cornell lucy sponza_tex sponza_lights}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p "$OUT"
for W in $WL; do
  case $W in
    sponza_lights) ARGS="--workload sponza --light-mode 1" ;;
    *) ARGS="--workload $W" ;;
  esac
  timeout -k 10 420 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_$W" -- python3 bench.py $ARGS --steps 10 --warmup 3 --cpu-budget 9 --pipelined 0 > "$OUT/bench_$W.json" 2> "$OUT/bench_$W.err"
  rc=$?
  echo "bench $W rc=$rc $(cut -c1-160 "$OUT/bench_$W.json")"
  [ $rc -eq 0 ] || exit 1
  find "$OUT/stats_$W" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats_$W.csv" \;
done
