export TMPDIR=/tmp
python -m pytest tests/test_gpu_learners.py -x -q -k "vi" > gpurun_out/t_vi.log 2>&1; grep -E "passed|failed|Error" gpurun_out/t_vi.log | tail -5
python bench.py --config ref_vi --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('ref_vi', d['ms_per_step'], d.get('gpu_ms_per_step_noise_resident'), d.get('ms_per_step_device_noise'))"
