export TMPDIR=/tmp; mkdir -p gpurun_out/r06i
timeout 1500 python -m pytest tests/test_gpu_map_persist.py -q -k "wide" 2>&1 | tail -15
python bench.py --config ref_map --no-cpu-baseline 2> gpurun_out/r06i/ref_map.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('ref_map', d['ms_per_step'], d['steady']['ms_per_step'], d['kernel_ms_per_step'], d['config'].get('finite'))"
tail -3 gpurun_out/r06i/ref_map.err
