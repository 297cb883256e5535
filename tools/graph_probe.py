"""What a hipGraph replay of a whole step costs against issuing its launches eagerly, for short steps (the strong-scaling shard of
cfg #3 and the PACOH-MAP configurations).   python tools/graph_probe.py svgd128|map256|map5"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import meta_learning_pacoh_amd as M  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else 'svgd128'
if which == 'svgd128':
    model = M.GPRegressionMetaLearnedSVGD(bench.make_tasks(128, 64, 4), num_particles=20, task_batch_size=-1, random_seed=0)
elif which == 'map256':
    rs = np.random.RandomState(27)
    model = M.GPRegressionMetaLearned([(rs.uniform(-5, 5, (32, 1)), rs.normal(size=(32, 1))) for _ in range(256)], covar_module='SE',
                                      mean_module='NN', task_batch_size=256, random_seed=1)
else:
    rs = np.random.RandomState(27)
    model = M.GPRegressionMetaLearned([(rs.uniform(-5, 5, (5, 1)), rs.normal(size=(5, 1))) for _ in range(20)], random_seed=1)
model._train_steps(30)
torch.cuda.synchronize()
t0 = time.perf_counter()
model._train_steps(500)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
tag = ' '.join('%s=%s' % (k, os.environ[k]) for k in ('PACOH_NO_GRAPH', 'PACOH_MLP_PATH', 'DEBUG_CLR_GRAPH_PACKET_CAPTURE', 'HIP_LAUNCH_BLOCKING') if k in os.environ)
print('%-8s %-40s host %.4f ms/step, step %.4f ms' % (which, tag or '(default)', (t1 - t0) / 500 * 1e3, (t2 - t0) / 500 * 1e3))
