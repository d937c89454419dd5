#!/bin/bash
# Full-size evidence for one build: the default bench line, the rocprofv3 kernel-trace statistics of
# the same command, and the hardware-counter passes at reduced size.  Usage: tools/prof_round.sh <tag>
set -u
TAG=${1:-r1}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
timeout -k 10 400 python3 bench.py --steps 5 --warmup 3 > "$OUT/bench.json" 2> "$OUT/bench.err" || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 bench.py --steps 5 --warmup 3 --no-cpu-baseline > "$OUT/bench_prof.json" 2> "$OUT/prof.err" || exit 1
tools/prof_pmc.sh "$OUT/pmc" || exit 1
python3 tools/pmc_summary.py "$OUT/pmc" k_main > "$OUT/pmc_summary.txt"
