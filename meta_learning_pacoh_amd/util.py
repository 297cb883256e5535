"""Host-side helpers with the interface of meta_learn/util.py: input shape handling (util.py:44-58), the 'gp-priors' logger
(util.py:60-92) and the learning-rate schedule.  (The quantile bisection of util.py:9-42 is the HIP kernel behind
pacoh_mixture_icdf.)"""
import contextlib
import gc
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


def host_cpu_budget():
    """CPUs this process may really use: the smaller of its affinity mask and its cgroup CPU quota (containers commonly show every
    core of the host -- 256 on the MI355X boxes -- behind a quota of 16).  torch sizes its intra-op pool by the visible cores; its
    workers spin between parallel regions, so a few medium-sized host-side tensor copies per chunk of steps were enough to burn the
    quota, and the kernel then froze the whole process for tens of milliseconds at a time (tools/short_region_probe.py: 11 of 11
    scheduler periods throttled, the GPU starved)"""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    quota = None
    try:
        with open('/sys/fs/cgroup/cpu.max') as fh:                      # cgroup v2: "<quota|max> <period>"
            q, per = fh.read().split()[:2]
            if q != 'max':
                quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            with open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us') as fq, open('/sys/fs/cgroup/cpu/cpu.cfs_period_us') as fp:
                q, per = float(fq.read()), float(fp.read())
                if q > 0 and per > 0:
                    quota = q / per
        except (OSError, ValueError):
            quota = None
    if quota is not None:
        n = min(n, max(1, int(quota)))
    return max(1, n)


@contextlib.contextmanager
def gc_paused():
    """No garbage collection inside a stream capture: a dead reference cycle that holds graphs, events or pinned memory of its own
    (a discarded learner, say) would be finalised by whatever allocation happens to trigger the collector, and destroying those
    inside a capture aborts the process.  torch.cuda.graph() no longer collects before capturing
    (torch.compiler.config.force_cudagraph_gc), so: collect on entry, keep the collector off until the block has ended."""
    was_on = gc.isenabled()
    gc.collect()
    gc.disable()
    try:
        yield
    finally:
        if was_on:
            gc.enable()

