export TMPDIR=/tmp
python -m pytest tests/test_gpu_learners.py tests/test_gpu_kernels.py tests/test_gpu_svgd_task.py -x -q -k "vi or step_begin or VI or feed" > gpurun_out/t_vi.log 2>&1; grep -E "passed|failed|Error" gpurun_out/t_vi.log | tail -5
python bench.py --config ref_vi --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('ref_vi', d['ms_per_step'], d.get('gpu_ms_per_step_noise_resident'), d.get('ms_per_step_device_noise'), d['kernel_ms_per_step'])"
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/vi_prof -- python3 $GRAFT_REPO_ROOT/bench.py --config ref_vi --no-cpu-baseline > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; f=$(ls gpurun_out/vi_prof/*/*kernel_stats.csv | head -1); head -6 $f | cut -c1-200
