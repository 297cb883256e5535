export TMPDIR=/tmp; mkdir -p gpurun_out/r06h
(python tools/task_fused_crossover.py 20 10 4; python tools/task_fused_crossover.py 32 10 2; python tools/task_fused_crossover.py 8 10 2) 2>&1 | tee gpurun_out/r06h/crossover.txt
timeout 900 python -m pytest tests/test_gpu_svgd_task.py tests/test_gpu_map_persist.py tests/test_gpu_multiproc.py -q 2>&1 | grep -E "passed|failed"
