"""pacoh_map_persist (K PACOH-MAP iterations per launch) against the four-launch iteration on the same draws, and its time per
iteration at BASELINE configs #1 / #2:   python tools/map_persist_check.py [--time-only]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import meta_learning_pacoh_amd as M                                     # noqa: E402
import bench                                                           # noqa: E402


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-300))


def compare():
    rs = np.random.RandomState(13)
    ragged, even = [], []
    for t in range(7):
        n = 8 + 2 * (t % 3)
        x = rs.uniform(-3, 3, size=(n, 2))
        ragged.append((x, np.sin(x[:, :1]) + 0.3 * x[:, 1:] + 0.05 * rs.randn(n, 1)))
        x = rs.uniform(-3, 3, size=(12, 2))
        even.append((x, np.sin(x[:, :1]) + 0.3 * x[:, 1:] + 0.05 * rs.randn(12, 1)))
    cfgs = [dict(), dict(covar_module='SE', mean_module='NN'), dict(covar_module='NN', mean_module='constant', feature_dim=3),
            dict(covar_module='SE', mean_module='constant'), dict(mean_nn_layers=(16, 16), kernel_nn_layers=(16, 16)),
            dict(learning_mode='learn_mean', covar_module='SE'), dict(learning_mode='learn_kernel', mean_module='constant'),
            dict(mean_nn_layers=(32, 32, 32, 32), kernel_nn_layers=(32, 32, 32, 32), weight_decay=0.0),
            dict(mean_nn_layers=(32, 32), kernel_nn_layers=(16,)), dict(covar_module='SE', mean_module='zero', learning_mode='learn_kernel')]
    worst = 0.0
    for tasks, name in ((ragged, 'ragged'), (even, 'even')):
        for cfg in cfgs:
            kw = dict(task_batch_size=4, lr_params=1e-2, weight_decay=0.05, lr_decay=0.9, random_seed=3)
            kw.update(cfg)
            out = []
            os.environ['PACOH_MAP_TASK_FUSED'] = '0'
            for persist in ('0', '1'):
                os.environ['PACOH_MAP_PERSIST'] = persist
                m = M.GPRegressionMetaLearned(tasks, **kw)
                loss = m.meta_fit(verbose=False, n_iter=14, log_period=4)
                if persist == '1' and m._persist is None:
                    print('   (not taken by the persistent kernel)', cfg)
                out.append((m.theta.clone(), m.exp_avg.clone(), m.exp_avg_sq.clone(), float(loss), float(m._g_cum)))
            # (the kernel network's output bias has an exactly-zero derivative -- a stationary kernel sees differences only --, so its
            #  gradient is rounding noise and AdamW turns the noise's sign into +-lr steps: excluded)
            keep = torch.ones_like(out[0][0], dtype=torch.bool)
            sl = m.layout.slices.get('kernel_nn.out.bias')
            if sl is not None:
                keep[0, sl[0]:sl[1]] = False
            errs = [rel(out[1][k][keep], out[0][k][keep]) for k in range(3)]
            dl = abs(out[1][3] - out[0][3]) / (abs(out[0][3]) + 1e-12)
            worst = max(worst, *errs, dl)
            print('%-6s %-90s theta %.1e m %.1e v %.1e loss %.1e cum %.6f / %.6f' % (name, cfg, errs[0], errs[1], errs[2], dl, out[1][4], out[0][4]), flush=True)
    print('worst relative difference', worst)
    os.environ.pop('PACOH_MAP_PERSIST')
    return worst


def timing():
    from meta_learning_pacoh_amd import _lib as L
    for cfg in (1, 2):
        for persist in ('0', '1'):
            os.environ['PACOH_MAP_PERSIST'] = os.environ['PACOH_MAP_TASK_FUSED'] = persist
            wl = bench.WORKLOADS[cfg](1, 'weak', M, L)
            for k in (64, 128, 64, 1024):
                wl['run'](k)
                torch.cuda.synchronize()
            t0 = time.perf_counter()
            wl['run'](4096)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 4096 * 1e3
            print('cfg #%d PACOH_MAP_PERSIST=PACOH_MAP_TASK_FUSED=%s: %.5f ms per iteration (finite %s)' % (cfg, persist, ms, wl['finite']()), flush=True)
    os.environ.pop('PACOH_MAP_PERSIST')
    os.environ.pop('PACOH_MAP_TASK_FUSED')


if __name__ == '__main__':
    if '--time-only' not in sys.argv:
        compare()
    timing()
