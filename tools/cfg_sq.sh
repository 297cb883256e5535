#!/bin/bash
# SQ counters of the dominant kernels of BASELINE configs #1, #2 and #4 (two --pmc passes each over a short bench.py --config N run,
# summarised per kernel by tools/pmc_kernel.py):  bash tools/cfg_sq.sh r05   -> gpurun_out/<tag>/cfg_sq_counters.txt
tag=${1:-rXX}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}" || exit 1
: > $out/cfg_sq_counters.txt
for spec in "4 gp_reg_kernel" "2 map_task_kernel" "1 map_persist_kernel"; do
  set -- $spec; c=$1; k=$2
  rm -rf $out/sq$c
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $out/sq$c/p1 -- python3 bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT --output-format csv -d $out/sq$c/p2 -- python3 bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  echo "== cfg #$c: $k" >> $out/cfg_sq_counters.txt
  python tools/pmc_kernel.py $out/sq$c $k >> $out/cfg_sq_counters.txt 2>&1
  rm -rf $out/sq$c
done
cat $out/cfg_sq_counters.txt
