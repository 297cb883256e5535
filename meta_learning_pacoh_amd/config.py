"""Device selection (the reference hard-codes torch.device('cpu') in config.py:3-4; here the hot
path exists only on a HIP device: one process per GPU, LOCAL_RANK picks it)."""
import os

import torch


def get_device():
    if not torch.cuda.is_available():
        raise RuntimeError('meta_learning_pacoh_amd needs a HIP device (MI355X); there is no CPU fallback. '
                           'For CPU use the reference implementation.')
    return torch.device('cuda', int(os.environ.get('LOCAL_RANK', '0')))
