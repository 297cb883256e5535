"""time per PACOH-SVGD step with the IMQ particle kernel at the cfg #3 shape: graph replay vs the same launches issued eagerly
usage: python tools/imq_time.py [n_tasks=1024] [steps=200]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from meta_learning_pacoh_amd import GPRegressionMetaLearnedSVGD, util  # noqa: E402

torch.set_num_threads(util.host_cpu_budget())
T = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
rs = np.random.RandomState(0)
tasks = []
for _ in range(T):
    x = rs.uniform(-3, 3, size=(64, 4))
    tasks.append((x, np.sin(x[:, :1]) + 0.1 * rs.randn(64, 1)))
for kernel in ('IMQ', 'RBF'):
    for no_graph in ('1', '0'):
        os.environ['PACOH_NO_GRAPH'] = no_graph
        m = GPRegressionMetaLearnedSVGD(tasks, num_particles=20, task_batch_size=T, kernel=kernel, random_seed=1)
        m.meta_fit(verbose=False, n_iter=64, log_period=10000)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m.meta_fit(verbose=False, n_iter=steps, log_period=10000)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        print('%s particle kernel, %d tasks x 20 particles, %s: %.4f ms per step' % (kernel, T, 'eager launches' if no_graph == '1' else 'graph replay', dt * 1e3))
