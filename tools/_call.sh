export TMPDIR=/tmp; mkdir -p gpurun_out/r06h
python tools/map_task_crossover.py 2>&1 | tee gpurun_out/r06h/map_crossover.txt
timeout 900 python -m pytest tests/test_gpu_map_persist.py tests/test_gpu_learners.py tests/test_gpu_fullsize.py -q 2>&1 | grep -E "passed|failed"
