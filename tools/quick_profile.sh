out=gpurun_out/$1
mkdir -p $out
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp
cd "$root" || exit 1
python bench.py --no-cpu-baseline --no-other-configs > $out/bench.json 2> $out/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --no-cpu-baseline --no-other-configs > $out/bench_under_rocprof.json 2>/dev/null
cp $(ls $out/stats/*/*kernel_stats.csv | head -1) $out/bench_kernel_stats.csv
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $out/sq/p1 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT --output-format csv -d $out/sq/p2 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs > /dev/null 2>&1
for k in gp_reg_kernel mlp_fused_bwd_kernel mlp_fused_fwd_kernel; do echo "== $k"; python tools/pmc_kernel.py $out/sq $k; done > $out/pmc_sq_counters.txt 2>&1
rm -rf $out/stats $out/sq
python - <<PY
import json,csv
d=json.load(open('$out/bench.json')); print(d['value'], d['ms_per_step'], d['host_ms_per_step'], d['kernel_rooflines'])
for r in csv.DictReader(open('$out/bench_kernel_stats.csv')):
    if int(r['Calls'])>=100: print('%-60s %5s %9.1f us'%(r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3))
PY
head -30 $out/pmc_sq_counters.txt
