export TMPDIR=/tmp
bash tools/profile_round.sh r06 > gpurun_out/profile_round_r06.log 2>&1
tail -3 gpurun_out/profile_round_r06.log
python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('headline', d['ms_per_step'], d['steady']['ms_per_step'], d['roofline']['traffic_profile'])"
