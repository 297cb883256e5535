"""Device selection (the reference hard-codes torch.device('cpu') in config.py:3-4; here the hot
path exists only on a HIP device: one process per GPU, LOCAL_RANK picks it)."""
import os

import torch


def get_device():
    if not torch.cuda.is_available():
        raise RuntimeError('meta_learning_pacoh_amd needs a HIP device (MI355X); there is no CPU fallback. '
                           'For CPU use the reference implementation.')
    local_rank = int(os.environ.get('LOCAL_RANK', '-1'))
    if 0 <= local_rank < torch.cuda.device_count():
        return torch.device('cuda', local_rank)
    return torch.device('cuda', torch.cuda.current_device())     # single-process use, or ranks sharing a device in tests
