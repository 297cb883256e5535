export TMPDIR=/tmp
for v in base w5 base w5; do
  if [ $v = base ]; then unset PACOH_LIB; else export PACOH_LIB=$PWD/meta_learning_pacoh_amd/lib/libpacoh_gp_$v.so; fi
  python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['kernel_ms_per_step'])"
done
