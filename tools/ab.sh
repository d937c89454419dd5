#!/bin/bash
# A/B timing of alternative builds of the HIP library on one reduced-size bench frame.
# usage: tools/ab.sh "<bench args>" lib1.so lib2.so ...
ARGS=$1; shift
for lib in "$@"; do
  for rep in 1 2; do
    RT_HIP_LIB=$lib timeout -k 10 200 python3 -u bench.py $ARGS --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$lib', 'Mrays/s %.1f' % d['value'], 'ms %.1f' % d['roofline']['kernel_ms_avg'])
" || exit 1
  done
done
