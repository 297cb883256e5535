"""Pin the oracle (A1-A6, A8, A11, A12 + eval metrics) against the ONLY recorded output of the
real reference: the PACOH-MAP demo trajectory stored in demo.ipynb:115-127,164-166
(transcribed into tests/golden/demo_log.json).  ~1 minute single-threaded."""
import json
import os

import numpy as np
import torch

from oracle import pacoh_oracle as O


def test_map_oracle_reproduces_recorded_reference_trajectory(golden_dir):
    with open(os.path.join(golden_dir, 'demo_log.json')) as f:
        gold = json.load(f)
    threads = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        env = O.SinusoidDataset(np.random.RandomState(26))
        train = env.generate_meta_train_data(20, 5)
        test = env.generate_meta_test_data(20, 5, 50)
        model = O.MapOracle(train, weight_decay=0.2, num_iter_fit=12000, random_seed=30)
        log = model.meta_fit(test, log_period=1000)
        final = model.eval_datasets(test)
    finally:
        torch.set_num_threads(threads)
    assert len(log) == len(gold['log'])
    for got, ref in zip(log, gold['log']):
        assert got[0] == ref[0]
        tol = 5e-6 if ref[0] == 1 else 5e-5            # iter-1 loss is exact to the printed digit
        assert abs(got[1] - ref[1]) < tol, (got, ref)
        for k in (2, 3, 4):                            # printed with 3 decimals
            assert abs(got[k] - ref[k]) < 1.5e-3, (got, ref)
    assert abs(final[0] - gold['final_test']['ll']) < 2e-3
    assert abs(final[1] - gold['final_test']['rmse']) < 2e-3
    assert abs(final[2] - gold['final_test']['calib']) < 2e-3


def test_single_task_oracle_reproduces_recorded_reference_log(golden_dir):
    """GPRegressionLearned cell of demo.ipynb: loss / valid-LL / RMSE / calibration printed with 3 decimals"""
    with open(os.path.join(golden_dir, 'demo_log.json')) as f:
        gold = json.load(f)['single_task']
    env = O.SinusoidDataset(np.random.RandomState(26))
    env.generate_meta_train_data(20, 5)
    xc, yc, xt, yt = env.generate_meta_test_data(20, 5, 50)[0]
    model = O.SingleTaskOracle(xc, yc, learning_mode='learn_mean', covar_module='SE', mean_module='constant', random_seed=30)
    log = model.fit(xt, yt)
    assert len(log) == len(gold['log'])
    for got, ref in zip(log, gold['log']):
        assert got[0] == ref[0]
        for k in (1, 2, 3, 4):
            assert abs(got[k] - ref[k]) < 6e-4, (got, ref)
