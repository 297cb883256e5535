"""Is the host side being throttled by the container's CPU quota?  Prints the cgroup quota, the cores torch sizes its intra-op pool
by, then runs 24 chunks of 64 training steps and reports per chunk the wall time per step, the host's issue time and the GPU-event
time per step, and finally the cgroup's cpu.stat delta (nr_throttled, throttled_usec).
    python tools/cpu_quota_probe.py [torch threads]        (no argument: torch's default pool)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
def stat():
    for p in ('/sys/fs/cgroup/cpu.stat', '/sys/fs/cgroup/cpu/cpu.stat'):
        if os.path.exists(p):
            return {l.split()[0]: int(l.split()[1]) for l in open(p)}
    return {}
for p in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us', '/sys/fs/cgroup/cpu/cpu.cfs_period_us'):
    if os.path.exists(p):
        print(p, open(p).read().strip())
print('os.cpu_count', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)), 'torch threads', torch.get_num_threads(), 'interop', torch.get_num_interop_threads())
if len(sys.argv) > 1:
    torch.set_num_threads(int(sys.argv[1]))
    print('set torch threads', torch.get_num_threads())
import bench
import meta_learning_pacoh_amd as M
model = M.GPRegressionMetaLearnedSVGD(bench.make_tasks(1024, 64, 4), num_particles=20, covar_module='NN', mean_module='NN',
                                      task_batch_size=-1, lr=1e-3, random_seed=0)
model._train_steps(40)
torch.cuda.synchronize()
s0 = stat()
tw = time.perf_counter()
for rep in range(24):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    model._train_steps(64)
    e1.record()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('chunk %d: wall %.4f ms/step, host issue %.3f ms total, GPU events %.4f ms/step' % (rep, (t2 - t0) / 64 * 1e3, (t1 - t0) * 1e3, e0.elapsed_time(e1) / 64))
    flag = torch.tensor([0.0], device='cuda'); float(flag.item())
s1 = stat()
print('elapsed %.2f s; cpu.stat delta' % (time.perf_counter() - tw), {k: s1[k] - s0.get(k, 0) for k in s1})
