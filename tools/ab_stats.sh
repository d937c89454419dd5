#!/bin/bash
# like ab.sh, but also prints the scheduler statistics of each build
ARGS=$1; shift
for lib in "$@"; do
  RT_HIP_LIB=$lib timeout -k 10 200 python3 -u bench.py $ARGS --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']; print('$lib', 'Mrays/s %.1f' % d['value'], 'steps_x64/ray %.1f' % r['per_ray']['sched_steps_x64'], 'lanes/step %.3f' % r['active_lane_fraction_per_step'])
" || exit 1
done
