export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/gputests_r06.log 2>&1; grep -E "passed|failed" gpurun_out/gputests_r06.log | tail -2
