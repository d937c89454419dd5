#!/bin/bash
# A/B of builds over several workloads: tools/ab_multi.sh "<workloads>" lib...
WL=$1; shift
for w in $WL; do
  for lib in "$@"; do
    RT_HIP_LIB=$lib timeout -k 10 400 python3 -u bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$w', '$lib', '%.1f Mrays/s %.2f ms' % (d['value'], d['ms_per_step']), flush=True)
" || exit 1
  done
done
