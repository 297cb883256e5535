mkdir -p gpurun_out/r06g; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_svgd_task.py tests/test_gpu_map_persist.py -q -x 2>&1 | tail -12
timeout 1500 python -m pytest tests/test_gpu_dense_path.py tests/test_gpu_chol_ll.py tests/test_gpu_fullsize.py tests/test_gpu_kernels.py tests/test_gpu_fuzz.py -q -x 2>&1 | tail -6
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r06g/raw -- python3 tools/raw_abi_dense.py 509 f32 > gpurun_out/r06g/raw_abi.txt 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/r06g/raw/*/*kernel_stats.csv')[0]
rows = list(csv.reader(open(f)))
with open('gpurun_out/r06g/raw_abi_n509_kernels.txt', 'w') as fh:
    fh.write('# rocprofv3 --kernel-trace --stats -- python3 tools/raw_abi_dense.py 509 f32  (pacoh_gp_lml_dense through raw ctypes, 5 calls of 16 problems)\n')
    for r in rows[:14]:
        fh.write('%-110s calls %s avg_ns %s\n' % (r[0][:110], r[1], r[3]))
print(open('gpurun_out/r06g/raw_abi_n509_kernels.txt').read())
PY
tail -2 gpurun_out/r06g/raw_abi.txt; rm -rf gpurun_out/r06g/raw
python tools/dense_pad_ab.py 2>&1 | tail -5
for v in 1 0; do PACOH_GP8=$v python bench.py --config 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('cfg1 gp8=$v', d['ms_per_step'], d['steady']['ms_per_step'], d['kernel_ms_per_step'])"; done
