export TMPDIR=/tmp
bash tools/profile_round.sh r06 > gpurun_out/profile_round_r06.log 2>&1
tail -40 gpurun_out/profile_round_r06.log
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r06/gputests.log 2>&1; grep -E "passed|failed" gpurun_out/r06/gputests.log | tail -2
