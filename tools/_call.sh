export TMPDIR=/tmp
bash tools/profile_round.sh r06 > gpurun_out/profile_round_r06.log 2>&1
tail -5 gpurun_out/profile_round_r06.log
