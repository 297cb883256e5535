export TMPDIR=/tmp
PACOH_SVGD_TASK_FUSED=0 python bench.py --config ref_svgd --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('ref_svgd general', d['ms_per_step'], d['steady']['ms_per_step'], d['kernel_ms_per_step'])"
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -5
