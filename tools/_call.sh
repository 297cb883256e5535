export TMPDIR=/tmp
python -m pytest tests/test_gpu_svgd_task.py tests/test_gpu_map_persist.py tests/test_gpu_learners.py tests/test_gpu_multiproc.py -x -q > gpurun_out/t_task.log 2>&1; grep -E "passed|failed|Error" gpurun_out/t_task.log | tail -5
