"""Host-side helpers mirroring meta_learn/util.py (logger, shape handling, LR schedule; the quantile bisection of util.py:9-42 is the HIP kernel behind pacoh_mixture_icdf)."""
import logging
import os

import numpy as np


def _handle_input_dimensionality(x, y=None):
    """meta_learn/util.py:44-58"""
    if x.ndim == 1:
        x = np.expand_dims(x, -1)
    assert x.ndim == 2
    if y is not None:
        if y.ndim == 1:
            y = np.expand_dims(y, -1)
        assert x.shape[0] == y.shape[0]
        assert y.ndim == 2
        return x, y
    return x


def get_logger(log_dir=None, log_file='output.log', expname=''):
    """meta_learn/util.py:60-92 (without the absl flag lookup): 'gp-priors' logger, same format."""
    logger = logging.getLogger('gp-priors')
    logger.setLevel(logging.INFO)
    if len(logger.handlers) == 0:
        if len(expname) > 0:
            expname = ' %s - ' % expname
        formatter = logging.Formatter('[%(asctime)s -' + '%s' % expname + '%(levelname)s]  %(message)s')
        sh = logging.StreamHandler()
        sh.setFormatter(formatter)
        sh.setLevel(logging.INFO)
        logger.addHandler(sh)
        logger.propagate = False
        if log_dir is not None and len(log_dir) > 0:
            fh = logging.FileHandler(os.path.join(log_dir, log_file))
            fh.setFormatter(formatter)
            fh.setLevel(logging.INFO)
            logger.addHandler(fh)
            logger.log_dir = log_dir
        else:
            logger.log_dir = None
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
