#!/bin/bash
# like ab.sh, but prints the executed work per ray of each build
ARGS=$1; shift
for lib in "$@"; do
  RT_HIP_LIB=$lib timeout -k 10 200 python3 -u bench.py $ARGS --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']['per_ray']; print('$lib', 'Mrays/s %.1f' % d['value'], 'visits %.2f cull %.2f tri %.2f steps_x64 %.1f' % (r['node_visits'], r['cull_tests'], r['tri_tests_done'], r['sched_steps_x64']))
" || exit 1
done
