"""Host time per SVGD step of the training loop (meta_fit's path): how long Python needs to issue a step, graph replay vs
eager launches (PACOH_NO_GRAPH=1), against the time the step takes on the GPU.
    python tools/host_issue_time.py [tasks]        (tasks per step: 1024 = cfg #3, 128 = its 1/8 shard)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import meta_learning_pacoh_amd as M  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
model = M.GPRegressionMetaLearnedSVGD(bench.make_tasks(T, 64, 4), num_particles=20, covar_module='NN', mean_module='NN',
                                      task_batch_size=-1, lr=1e-3, random_seed=0)
for mode in ('eager', 'graph', 'eager', 'graph'):          # (alternating: the first measurement also pays for a cold host)
    os.environ['PACOH_GRAPH'] = '1' if mode == 'graph' else '0'
    if mode == 'eager':
        os.environ['PACOH_NO_GRAPH'] = '1'
    else:
        os.environ.pop('PACOH_NO_GRAPH', None)
    model._step_mode.use_graph, model._step_mode.forced = mode == 'graph', True
    model._train_steps(20)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    model._train_steps(400)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('%d tasks x 20 particles, %s: host issue time per step %.4f ms; step %.4f ms'
          % (T, mode, (t1 - t0) / 400 * 1e3, (t2 - t0) / 400 * 1e3))
