#!/bin/bash
# Per-ray work of kernel builds on one workload: tools/ab_work.sh <steps> lib1.so lib2.so ... ("base" = raytrace_amd/lib);
# AB_ARGS as for tools/ab_bench.sh.  One line per build.
cd "$GRAFT_REPO_ROOT" || exit 1
steps=$1; shift
mkdir -p gpurun_out
for lib in "$@"; do
  name=$(basename "$lib" .so)
  if [ "$lib" = base ]; then unset RT_HIP_LIB; else export RT_HIP_LIB="$lib"; fi
  timeout -k 10 250 python3 bench.py --steps "$steps" --warmup 3 --no-cpu-baseline $AB_ARGS > "gpurun_out/abw_$name.json" 2> "gpurun_out/abw_$name.err" || { echo "$name FAILED"; tail -3 "gpurun_out/abw_$name.err"; exit 1; }
  python3 - "$name" <<'PY'
import json, sys
name = sys.argv[1]
d = json.loads(open(f"gpurun_out/abw_{name}.json").read().strip().splitlines()[-1])
p = d["roofline"]["per_ray"]
print(f"{name:10s} {d['value']:8.1f} Mrays/s kernel {d['roofline']['kernel_ms_avg']:8.3f} ms lanes {d['roofline']['active_lane_fraction_per_step']:.3f} "
      f"fetch {p['node_fetches']:.2f} cull {p['cull_tests']:.2f} tri {p['tri_tests_done']:.2f} steps*64 {p['sched_steps_x64']:.2f} B/ray {p['alg_bytes']:.0f}")
PY
done
