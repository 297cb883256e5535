mkdir -p gpurun_out/r06d; export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -8
python bench.py > gpurun_out/r06d/bench.json 2> gpurun_out/r06d/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r06d/bench.json'))
print(d['value'], d['ms_per_step'], d['steady'])
for k,v in d['other_configs'].items():
    print(k, v.get('ms_per_step'), v.get('error'), v.get('kernel_ms_per_step'), (v.get('cpu_baseline') or {}).get('value'))
PY
