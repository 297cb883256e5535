"""host cost of a graph replay per step when one graph holds K consecutive SVGD steps (cfg #3): is hipGraphLaunch's host time per
launch or per kernel node?    python tools/graph_steps_probe.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import meta_learning_pacoh_amd as M  # noqa: E402
from meta_learning_pacoh_amd.engine import capture_graph  # noqa: E402

model = M.GPRegressionMetaLearnedSVGD(bench.make_tasks(1024, 64, 4), num_particles=20, covar_module='NN', mean_module='NN',
                                      task_batch_size=-1, lr=1e-3, random_seed=0)
model._setup_step(model._local_batch_size())
N = 480


def upload():
    idx_rows, sc_rows = model._draw_steps(N, model.lr_scheduler, model.opt_step + 1)
    model._feed.upload(idx_rows, sc_rows)


upload()
graphs = {}
for K in (1, 2, 4, 8):
    def body(K=K):
        for _ in range(K):
            model._body_likelihood()
            model._body_update()
    graphs[K] = capture_graph(body)
for rep in range(2):
    for K in (1, 2, 4, 8):
        upload()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(N // K):
            graphs[K].replay()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print('%d step(s) per graph: host %.4f ms per step, %.4f ms per step in all' % (K, (t1 - t0) / N * 1e3, (t2 - t0) / N * 1e3))
print(open('/proc/loadavg').read().strip())
