#!/bin/bash
# FETCH_SIZE of a known number of requested bytes, per access pattern (tools/gather_probe.hip). Usage: tools/prof_gather_probe.sh <outdir>
set -u
OUT=${1:-gpurun_out/gather_probe}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p "$OUT"
for M in stream gather64 gather128; do
  timeout -s KILL 90 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/$M" -- raytrace_amd/bin/gather_probe $M > "$OUT/$M.json" 2> "$OUT/$M.err" || exit 1
  timeout -s KILL 90 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum --output-format csv -d "$OUT/${M}_ea" -- raytrace_amd/bin/gather_probe $M > /dev/null 2> "$OUT/${M}_ea.err" || echo "(EA counters not available)"
  echo "$M $(cat $OUT/$M.json)"
  grep -h "k_gather\|k_stream" $(find "$OUT/$M" "$OUT/${M}_ea" -name "*counter_collection.csv") | awk -F, '{print $(NF-1), $NF}'
done
