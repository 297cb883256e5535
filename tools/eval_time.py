"""wall time of eval_datasets (the validation pass meta_fit runs every log_period) on the demo test set, batched pass vs the
per-task loop over eval()."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import meta_learning_pacoh_amd as M                      # noqa: E402
from bench import make_tasks                             # noqa: E402  (synthetic sinusoid tasks)


def timed(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def split(tasks, n_ctx):
    return [(x[:n_ctx], y[:n_ctx], x[n_ctx:], y[n_ctx:]) for x, y in tasks]


train = make_tasks(20, 5, 1)
test = split(make_tasks(20, 55, 1, seed0=5000), 5)
big = split(make_tasks(100, 220, 1, seed0=7000), 20)
for name, model in (('MAP', M.GPRegressionMetaLearned(train, num_iter_fit=20, random_seed=1)),
                    ('SVGD P=10', M.GPRegressionMetaLearnedSVGD(train, num_iter_fit=5, num_particles=10, random_seed=1))):
    model.meta_fit(verbose=False)
    for label, tuples in (('20 tasks x (5 ctx, 50 test)', test), ('100 tasks x (20 ctx, 200 test)', big)):
        loop = timed(lambda: np.array([model.eval(*t) for t in tuples]).mean(0), reps=5)
        batched = timed(lambda: model.eval_datasets(tuples), reps=5)
        print('%-10s %-32s per-task loop %8.2f ms   batched eval_datasets %7.2f ms' % (name, label, loop, batched))
