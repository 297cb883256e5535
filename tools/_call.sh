export TMPDIR=/tmp
for v in 1; do echo "== gp8=$v"; PACOH_GP8=$v PACOH_LIB=$PWD/meta_learning_pacoh_amd/lib/libpacoh_gp_mpst.so python tools/map_persist_stamps.py 2>&1 | grep "mp stamp" | head -10 | tr '\n' ';'; echo; done
timeout 900 python -m pytest tests/test_gpu_svgd_task.py tests/test_gpu_map_persist.py -q -x 2>&1 | tail -3
for v in 1 0; do PACOH_GP8=$v python bench.py --config 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('cfg1 gp8=$v', d['ms_per_step'], d['steady']['ms_per_step'], d['kernel_ms_per_step'])"; done
