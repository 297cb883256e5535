mkdir -p gpurun_out/r06b; export TMPDIR=/tmp
python tools/dense_fp32_errors.py > gpurun_out/r06b/dense_fp32_errors.txt 2> gpurun_out/r06b/dense_fp32_errors.err
tail -5 gpurun_out/r06b/dense_fp32_errors.txt; tail -3 gpurun_out/r06b/dense_fp32_errors.err
python tools/spread_probe.py 128 > gpurun_out/r06b/spread_probe_128.txt 2>&1; cat gpurun_out/r06b/spread_probe_128.txt
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15
