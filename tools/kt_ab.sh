#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for lib in pretex base cold256; do
  RT_BENCH_PG=none RT_HIP_LIB=ab/$lib.so timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_$lib -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/kt_$lib.json 2>/dev/null
  echo $lib; grep -h "k_main\|k_put_cold\|fillBuffer" $(find gpurun_out/kt_$lib -name "*kernel_stats.csv") | cut -d, -f2-8 | head -4
done
