"""long-run sanity: 3000 bench-sized SVGD steps (finite particles, flat device memory) and a 3000-iteration meta_fit of each learner
on the demo-sized problem with validation logging every 500 iterations"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import meta_learning_pacoh_amd as M                      # noqa: E402
from bench import make_tasks                             # noqa: E402

tasks = make_tasks(1024, 64, 4)
model = M.GPRegressionMetaLearnedSVGD(tasks, num_particles=20, feature_dim=2, random_seed=0)
model.meta_fit(verbose=False, n_iter=10)
torch.cuda.synchronize()
m0 = torch.cuda.memory_allocated()
t0 = time.perf_counter()
model.meta_fit(verbose=False, n_iter=3000)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print('3000 SVGD steps at the bench shape: %.3f ms/step, particles finite: %s, device memory %+d bytes'
      % (dt / 3000 * 1e3, bool(torch.isfinite(model.particles).all()), torch.cuda.memory_allocated() - m0))


def split(ts, n_ctx):
    return [(x[:n_ctx], y[:n_ctx], x[n_ctx:], y[n_ctx:]) for x, y in ts]


train, valid = make_tasks(20, 5, 1), split(make_tasks(20, 55, 1, seed0=5000), 5)
for name, mk in (('MAP', lambda: M.GPRegressionMetaLearned(train, num_iter_fit=3000, random_seed=1)),
                 ('SVGD', lambda: M.GPRegressionMetaLearnedSVGD(train, num_iter_fit=3000, num_particles=10, random_seed=1)),
                 ('VI', lambda: M.GPRegressionMetaLearnedVI(train, num_iter_fit=3000, svi_batch_size=10, random_seed=1))):
    model = mk()
    t0 = time.perf_counter()
    model.meta_fit(valid_tuples=valid, verbose=False, log_period=500)
    torch.cuda.synchronize()
    ll, rmse, calib = model.eval_datasets(valid)
    print('%-5s 3000 iterations with validation every 500: %.2f s; final valid ll %.3f rmse %.3f calib %.3f'
          % (name, time.perf_counter() - t0, ll, rmse, calib))
    assert np.isfinite([ll, rmse, calib]).all()
