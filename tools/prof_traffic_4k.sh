#!/bin/bash
# HBM traffic of ONE full-size launch of the headline workload: FETCH_SIZE and WRITE_SIZE in a pass of their own each
# (one frame, no warm-up), summed over the kernel's dispatch.  Usage: tools/prof_traffic_4k.sh <outdir>
set -u
OUT=${1:-gpurun_out/traffic4k}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p "$OUT"
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-parity-check --pipelined 0 > "$OUT/fetch.log" 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-parity-check --pipelined 0 > "$OUT/write.log" 2>&1 || exit 1
python3 tools/pmc_summary.py "$OUT" k_main > "$OUT/summary.txt"
cat "$OUT/summary.txt"
grep -h "^{" "$OUT/fetch.log" | head -1 > "$OUT/bench_line.json"
