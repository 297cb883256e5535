"""bench.py's flow around a SHORT timed region (the driver times --steps 20), repeated: warm-up chunks of 64 steps until stable,
then ten 20-step regions bracketed by synchronisations, each with the wall time and the GPU-side (event) time of its steps.
    python tools/short_region_probe.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import meta_learning_pacoh_amd as M  # noqa: E402

model = M.GPRegressionMetaLearnedSVGD(bench.make_tasks(1024, 64, 4), num_particles=20, covar_module='NN', mean_module='NN',
                                      task_batch_size=-1, lr=1e-3, random_seed=0)
t_start = time.perf_counter()
model._train_steps(40)
torch.cuda.synchronize()
prev, t_warm, chunks = None, time.perf_counter(), 0
for _ in range(40):
    t_c = time.perf_counter()
    model._train_steps(64)
    torch.cuda.synchronize()
    cur = time.perf_counter() - t_c
    chunks += 1
    stable = prev is not None and abs(cur - prev) <= 0.03 * prev and time.perf_counter() - t_warm >= 1.0
    flag = torch.tensor([1.0 if (stable or time.perf_counter() - t_warm > 5.0) else 0.0], device='cuda')
    if float(flag.item()) > 0:
        break
    prev = cur
print('warm-up: %d chunks, %.2f s; last chunk %.4f ms/step; mode %s' % (chunks, time.perf_counter() - t_warm, cur / 64 * 1e3,
                                                                       model._step_mode.timings))
for rep in range(10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    model._train_steps(20)
    e1.record()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('region %d: wall %.4f ms/step (host issue %.3f ms total), GPU events %.4f ms/step'
          % (rep, (t2 - t0) / 20 * 1e3, (t1 - t0) * 1e3, e0.elapsed_time(e1) / 20))
    if rep == 4:
        time.sleep(0.05)
