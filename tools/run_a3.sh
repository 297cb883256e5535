mkdir -p gpurun_out/a3
timeout 1500 python -m pytest tests/test_gpu_multiproc.py tests/test_gpu_fullsize.py -m gpu -x -q > gpurun_out/a3/tests.log 2>&1; tail -5 gpurun_out/a3/tests.log
python bench.py > gpurun_out/a3/bench_cfg3.json 2> gpurun_out/a3/bench_cfg3.err; tail -c 2500 gpurun_out/a3/bench_cfg3.json; tail -3 gpurun_out/a3/bench_cfg3.err
for c in 2 4 5; do python bench.py --config $c --no-cpu-baseline > gpurun_out/a3/bench_cfg$c.json 2> gpurun_out/a3/bench_cfg$c.err; python -c "
import json;d=json.load(open('gpurun_out/a3/bench_cfg$c.json'));print($c, d['value'], d['ms_per_step'], d['host_ms_per_step'], d['kernel_ms_per_step'], d['roofline'] and (d['roofline']['kernel'], d['roofline']['algorithmic_frac']))" ; tail -2 gpurun_out/a3/bench_cfg$c.err; done
python tools/host_issue_time.py 1024 2>/dev/null; python tools/host_issue_time.py 128 2>/dev/null
for sc in weak strong; do PACOH_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 50 --warmup 5 --scaling $sc 2>gpurun_out/a3/gloo_$sc.err | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('gloo x2', d['scaling'], d['value'], d['ms_per_step'], d['host_ms_per_step'], d['backend'], d['world_size_seen'])"; done
