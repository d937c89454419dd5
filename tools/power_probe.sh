#!/bin/bash
# Samples engine clock and power while a multi-frame bench runs (diagnostic only).
cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 200 python3 bench.py --sizex 1080 --sizey 1920 --steps 120 --warmup 1 --no-cpu-baseline > gpurun_out/pp_bench.json 2>gpurun_out/pp_bench.err &
BP=$!
for i in $(seq 1 40); do
  echo "t=$i $(timeout 10 rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|Power \(|power" | tr -s '\t ' ' ' | tr '\n' ';')"
  if ! kill -0 $BP 2>/dev/null; then break; fi
  sleep 0.7
done
wait $BP
grep -o '"value": [0-9.]*' gpurun_out/pp_bench.json | head -1
