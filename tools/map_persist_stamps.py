"""phase stamps of the persistent PACOH-MAP kernel (csrc/map_persist.hip) at BASELINE config #1: build the diagnostic library with
    python -m meta_learning_pacoh_amd._build --variant mpst -DPACOH_MP_STAMPS=1
and run   PACOH_LIB=$PWD/meta_learning_pacoh_amd/lib/libpacoh_gp_mpst.so python tools/map_persist_stamps.py
-> shader cycles between the phase boundaries of the last iteration of a launch, for waves 0 and 15 (device printf)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import meta_learning_pacoh_amd as M                                     # noqa: E402
from meta_learning_pacoh_amd import _lib as L                           # noqa: E402
import bench                                                           # noqa: E402

wl = bench.WORKLOADS[1](1, 'weak', M, L)
for _ in range(2):
    wl['run'](64)
    torch.cuda.synchronize()
