#!/bin/bash
# A/B of kernel builds on the headline workload: tools/ab_bench.sh <steps> lib1.so lib2.so ... (paths relative to the
# repository; "base" = the library in raytrace_amd/lib).  One bench line per build under gpurun_out/ab_<name>.json;
# stops at the first failure.
cd "$GRAFT_REPO_ROOT" || exit 1
steps=$1; shift
mkdir -p gpurun_out
for lib in "$@"; do
  name=$(basename "$lib" .so)
  if [ "$lib" = base ]; then unset RT_HIP_LIB; else export RT_HIP_LIB="$lib"; fi
  timeout -k 10 150 python3 bench.py --steps "$steps" --warmup 3 --no-cpu-baseline $AB_ARGS > "gpurun_out/ab_$name.json" 2> "gpurun_out/ab_$name.err" || { echo "$name FAILED"; tail -3 "gpurun_out/ab_$name.err"; exit 1; }
  python3 - "$name" <<'PY'
import json, sys
name = sys.argv[1]
d = json.loads(open(f"gpurun_out/ab_{name}.json").read().strip().splitlines()[-1])
print(f"{name:12s} {d['value']:9.1f} Mrays/s  {d['ms_per_step']:8.3f} ms  kernel {d['roofline']['kernel_ms_avg']:8.3f} ms  lanes {d['roofline']['active_lane_fraction_per_step']:.3f}")
PY
done
