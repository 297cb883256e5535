"""BASELINE config #2 for a profiler: a few hundred PACOH-MAP iterations (256 tasks x 32 points, SE kernel + NN mean)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import meta_learning_pacoh_amd as M                                     # noqa: E402
from meta_learning_pacoh_amd import _lib as L                           # noqa: E402
import bench                                                           # noqa: E402

wl = bench.WORKLOADS[int(sys.argv[1]) if len(sys.argv) > 1 else 2](1, 'weak', M, L)
for k in (64, 128, 64, 512):
    wl['run'](k)
    torch.cuda.synchronize()
