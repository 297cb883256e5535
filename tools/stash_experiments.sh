# the stash-traffic experiments on the cfg #3 step, judged on the whole replayed step.  Needs the variant library
#   python -m meta_learning_pacoh_amd._build --variant sn1 -DPACOH_EXP_STASH_NETS=1     (-> profiles/r04_stash_experiments.txt)
export TMPDIR=/tmp
out=gpurun_out/stash; mkdir -p $out
echo "== (a) one network stashed, the other recomputed (variant sn1) vs product" > $out/log.txt
for i in 1 2; do
python bench.py --steps 200 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('product   ms/step', d['ms_per_step'], d['kernel_ms_per_step'])" >> $out/log.txt
PACOH_LIB=$PWD/meta_learning_pacoh_amd/lib/libpacoh_gp_sn1.so python bench.py --steps 200 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('one-net   ms/step', d['ms_per_step'], d['kernel_ms_per_step'])" >> $out/log.txt
done
echo "== (b) chunked pass" >> $out/log.txt
for c in 1 2 4 8; do python tools/stash_probe.py $c 100 2>/dev/null | grep "ms per pass" >> $out/log.txt; done
for c in 1 4 8; do
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/p$c/fetch -- python3 tools/stash_probe.py $c 3 > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/p$c/write -- python3 tools/stash_probe.py $c 3 > /dev/null 2>&1
  echo "-- chunks=$c: HBM bytes per launch (FETCHx2 + WRITE)" >> $out/log.txt
  python tools/pmc_summary.py $out/p$c $out/pmc_$c.json | grep -E "mlp_fused|gp_reg" >> $out/log.txt
  rm -rf $out/p$c
done
cat $out/log.txt
