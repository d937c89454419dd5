#!/bin/bash
# Hardware-counter passes over one reduced-size bench frame; each pass is its own rocprofv3 run
# (counters are not split over passes automatically). Usage: tools/prof_pmc.sh <outdir> [bench args]
set -u
OUT=${1:-gpurun_out/pmc}; shift || true
ARGS=${@:---sizex 1080 --sizey 1920 --steps 1 --warmup 0 --no-cpu-baseline}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p "$OUT"
run() { # name counters...
  local name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 bench.py $ARGS > "$OUT/$name.log" 2>&1
  local rc=$?
  echo "pass $name rc=$rc"
  return $rc
}
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU &&
run sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 &&
run sq3 SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT &&
run tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE &&
run fetch FETCH_SIZE &&
run write WRITE_SIZE TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
