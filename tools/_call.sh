export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/gputests_r06.log 2>&1; grep -E "passed|failed" gpurun_out/gputests_r06.log | tail -2; grep -E "^FAILED|^ERROR" gpurun_out/gputests_r06.log | head
PACOH_MAP_TASK_FUSED=0 python bench.py --config ref_map --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('ref_map general', d['ms_per_step'], d['steady']['ms_per_step'], d['kernel_ms_per_step'])"
