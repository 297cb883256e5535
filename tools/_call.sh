export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_map_persist.py -q -x -k "wide" 2>&1 | tail -4
timeout 300 python bench.py --config ref_map --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('ref_map', d['ms_per_step'], d['steady']['ms_per_step'], d['kernel_ms_per_step'])"
PACOH_LIB=$PWD/meta_learning_pacoh_amd/lib/libpacoh_gp_mpst.so PACOH_NO_GRAPH=1 timeout 300 python tools/svgd_task_stamps.py ref_map 2>&1 | grep "mw stamp" | head -40 | tr '\n' ';'; echo
