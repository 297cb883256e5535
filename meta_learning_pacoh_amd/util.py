"""Host-side helpers with the interface of meta_learn/util.py: input shape handling (util.py:44-58), the 'gp-priors' logger
(util.py:60-92) and the learning-rate schedule.  (The quantile bisection of util.py:9-42 is the HIP kernel behind
pacoh_mixture_icdf.)"""
import logging
import os

import numpy as np

LOG_FORMAT = '[%%(asctime)s -%s%%(levelname)s]  %%(message)s'       # the reference's line format, experiment name spliced in


def _as_columns(a, what):
    """1-D array -> one column; anything but a 1-D or 2-D array is refused"""
    a = np.asarray(a)
    if a.ndim not in (1, 2):
        raise AssertionError('%s must be a 1-D or 2-D array, got %d dimensions' % (what, a.ndim))
    return a.reshape(-1, 1) if a.ndim == 1 else a


def _handle_input_dimensionality(x, y=None):
    """(x[, y]) as 2-D arrays [n, d] (/ [n, d_y]) with matching numbers of rows -- the contract of util.py:44-58"""
    x = _as_columns(x, 'x')
    if y is None:
        return x
    y = _as_columns(y, 'y')
    if len(x) != len(y):
        raise AssertionError('x and y hold different numbers of points: %d vs %d' % (len(x), len(y)))
    return x, y


def get_logger(log_dir=None, log_file='output.log', expname=''):
    """The process-wide 'gp-priors' logger, configured on first use: INFO to the console and, given log_dir, to
    log_dir/log_file; later calls return it unchanged.  `logger.log_dir` records where the file handler writes."""
    logger = logging.getLogger('gp-priors')
    if logger.handlers:
        return logger
    logger.setLevel(logging.INFO)
    logger.propagate = False
    formatter = logging.Formatter(LOG_FORMAT % (' %s - ' % expname if expname else ''))
    sinks = [logging.StreamHandler()]
    logger.log_dir = log_dir if log_dir else None
    if logger.log_dir:
        sinks.append(logging.FileHandler(os.path.join(log_dir, log_file)))
    for sink in sinks:
        sink.setLevel(logging.INFO)
        sink.setFormatter(formatter)
        logger.addHandler(sink)
    return logger


class StepLR:
    """torch.optim.lr_scheduler.StepLR(optimizer, 1000, gamma) as used at GPR_meta_mll.py:261-264;
    gamma >= 1 reproduces DummyLRScheduler (util.py:94-100)."""

    def __init__(self, base_lr, step_size=1000, gamma=1.0):
        self.base_lr, self.step_size, self.gamma = base_lr, step_size, gamma
        self.epoch = 0

    def step(self):
        self.epoch += 1

    @property
    def lr(self):
        if self.gamma >= 1.0:
            return self.base_lr
        return self.base_lr * self.gamma ** (self.epoch // self.step_size)

    def lrs(self, k):
        """learning rates of the next k steps (the scheduler itself is not advanced)"""
        if self.gamma >= 1.0:
            return np.full(k, self.base_lr, dtype=np.float64)
        return self.base_lr * self.gamma ** ((self.epoch + np.arange(k)) // self.step_size).astype(np.float64)
