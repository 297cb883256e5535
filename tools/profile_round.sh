#!/bin/bash
# All the measurements a round's profiles/ directory is built from, in one gpurun call:
#   bash tools/profile_round.sh r02        (writes gpurun_out/<tag>/..., copy what is to be judged into profiles/)
tag=${1:-rXX}
out=gpurun_out/$tag
mkdir -p $out
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp
cd "$root" || exit 1
python bench.py > $out/bench.json 2> $out/bench.err
python bench.py --scaling strong --no-cpu-baseline > $out/bench_strong1.json 2>> $out/bench.err
for c in 2 4 5; do python bench.py --config $c --no-cpu-baseline > $out/bench_cfg$c.json 2>> $out/bench.err; done
tools/mfma_peak > $out/mfma_peak.txt 2>&1
# per-kernel times of the default bench command (graph replays + the eager instrumented pass)
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --no-cpu-baseline --no-other-configs > $out/bench_under_rocprof.json 2>/dev/null
cp $(ls $out/stats/*/*kernel_stats.csv | head -1) $out/bench_kernel_stats.csv
# ... and of BASELINE configs #2 (the task-fused MAP kernel) and #4 (the n = 128 GP kernel)
for c in 2 4; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats$c -- python3 bench.py --config $c --no-cpu-baseline > /dev/null 2>&1
  cp $(ls $out/stats$c/*/*kernel_stats.csv | head -1) $out/cfg${c}_kernel_stats.csv; rm -rf $out/stats$c
done
# ... of the reference launchers' own SVGD / VI shape (task-fused likelihood launch) and of cfg #3's 1/8 strong-scaling shard (round 6)
for c in ref_svgd ref_vi ref_map shard128; do
  python bench.py --config $c --no-cpu-baseline > $out/bench_$c.json 2>> $out/bench.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$c -- python3 bench.py --config $c --no-cpu-baseline > /dev/null 2>&1
  cp $(ls $out/stats_$c/*/*kernel_stats.csv | head -1) $out/${c}_kernel_stats.csv; rm -rf $out/stats_$c
done
PACOH_SVGD_TASK_FUSED=0 python bench.py --config ref_svgd --no-cpu-baseline > $out/bench_ref_svgd_general.json 2>> $out/bench.err
PACOH_SVGD_TASK_FUSED=0 python bench.py --config ref_vi --no-cpu-baseline > $out/bench_ref_vi_general.json 2>> $out/bench.err
PACOH_MAP_TASK_FUSED=0 python bench.py --config ref_map --no-cpu-baseline > $out/bench_ref_map_general.json 2>> $out/bench.err
# SQ counters of the task-fused kernel at the launcher shape
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $out/sqt/p1 -- python3 bench.py --config ref_svgd --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT --output-format csv -d $out/sqt/p2 -- python3 bench.py --config ref_svgd --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
for k in map_task_kernel fused_reduce_slab svgd_update_kernel; do echo "== $k (ref_svgd)"; python tools/pmc_kernel.py $out/sqt $k; done > $out/ref_svgd_sq_counters.txt 2>&1
rm -rf $out/sqt
# HBM traffic: separate FETCH_SIZE / WRITE_SIZE passes
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc/fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc/write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs > /dev/null 2>&1
python tools/pmc_summary.py $out/pmc $out/pmc_hbm_traffic.json > $out/pmc_hbm_traffic.txt
# SQ counters of the step's kernels
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $out/sq/p1 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT --output-format csv -d $out/sq/p2 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs > /dev/null 2>&1
for k in gp_reg_kernel mlp_fused_bwd_kernel mlp_fused_fwd_kernel; do echo "== $k"; python tools/pmc_kernel.py $out/sq $k; done > $out/pmc_sq_counters.txt 2>&1
# large-context path (cfg 5), fp64 and fp32
rocprofv3 --kernel-trace --stats --output-format csv -d $out/dense64 -- python3 tools/dense_profile.py > /dev/null 2>&1
cp $(ls $out/dense64/*/*kernel_stats.csv | head -1) $out/dense_kernel_stats_fp64.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $out/dense32 -- python3 tools/dense_profile.py f32 > /dev/null 2>&1
cp $(ls $out/dense32/*/*kernel_stats.csv | head -1) $out/dense_kernel_stats_fp32.csv
python tools/host_issue_time.py 1024 > $out/host_issue.txt 2>/dev/null
python tools/host_issue_time.py 128 >> $out/host_issue.txt 2>/dev/null
rm -rf $out/stats $out/pmc $out/sq $out/dense64 $out/dense32
ls -la $out
head -c 1500 $out/bench.json; echo; cat $out/mfma_peak.txt; head -12 $out/bench_kernel_stats.csv; head -8 $out/dense_kernel_stats_fp64.csv; cat $out/host_issue.txt
# HBM traffic of the large-context path's kernels (what bench.py's other_configs.cfg5.hbm block reads)
bash tools/dense_pmc.sh $tag
# SQ counters of the large-context path's kernels
bash tools/dense_sq.sh $tag > /dev/null 2>&1
python tools/imq_time.py 1024 200 2> /dev/null | grep "ms per step" > $out/imq_step.txt
