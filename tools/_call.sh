mkdir -p gpurun_out/r06e; export TMPDIR=/tmp
PACOH_LIB=$PWD/meta_learning_pacoh_amd/lib/libpacoh_gp_mpst.so PACOH_NO_GRAPH=1 python tools/svgd_task_stamps.py 2>&1 | grep "mt stamp" | head -20 | tr '\n' ' '; echo
for c in ref_svgd ref_vi 2; do
  python bench.py --config $c --no-cpu-baseline > gpurun_out/r06e/bench_$c.json 2> gpurun_out/r06e/bench_$c.err
done
python - <<'PY'
import json
for c in ('ref_svgd', 'ref_vi', '2'):
    try:
        d = json.load(open('gpurun_out/r06e/bench_%s.json' % c))
        print(c, d['ms_per_step'], d['steady']['ms_per_step'], d.get('gpu_ms_per_step_noise_resident'), d['kernel_ms_per_step'], d['config'].get('finite'))
    except Exception as e:
        print(c, 'ERR', e)
PY
timeout 900 python -m pytest tests/test_gpu_svgd_task.py tests/test_gpu_map_persist.py -q 2>&1 | tail -4
