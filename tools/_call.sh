export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_map_persist.py -q -k "wide" 2>&1 | tail -4
python bench.py --config ref_map --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('ref_map', d['ms_per_step'], d['steady']['ms_per_step'], d['kernel_ms_per_step'])"
