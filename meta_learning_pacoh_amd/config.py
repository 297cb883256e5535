"""Device selection (the reference hard-codes torch.device('cpu') in config.py:3-4; here the hot
path exists only on a HIP device: one process per GPU, LOCAL_RANK picks it)."""
import os

import torch


def get_device():
    if not torch.cuda.is_available():
        raise RuntimeError('meta_learning_pacoh_amd needs a HIP device (MI355X); there is no CPU fallback. '
                           'For CPU use the reference implementation.')
    local_rank = int(os.environ.get('LOCAL_RANK', '-1'))
    if 0 <= local_rank < torch.cuda.device_count() and os.environ.get('PACOH_SHARE_DEVICE', '0') != '1':
        # the kernels are launched on the CURRENT device's stream with raw pointers: make the rank's device current, so that
        # a caller who forgot torch.cuda.set_device(LOCAL_RANK) cannot end up launching on device 0 with device-N pointers
        torch.cuda.set_device(local_rank)
        return torch.device('cuda', local_rank)
    return torch.device('cuda', torch.cuda.current_device())     # single-process use, or ranks sharing a device in tests
