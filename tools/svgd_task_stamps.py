"""phase stamps of the task-fused likelihood kernel (csrc/map_task.hip, map_task_kernel<.., MULTI>) at the reference's SVGD launcher
shape (bench.py: ref_svgd): build the diagnostic library with
    python -m meta_learning_pacoh_amd._build --variant mpst -DPACOH_MP_STAMPS=1
and run   PACOH_LIB=$PWD/meta_learning_pacoh_amd/lib/libpacoh_gp_mpst.so PACOH_NO_GRAPH=1 python tools/svgd_task_stamps.py
-> shader cycles between the phase boundaries of workgroup 7, wave 0 (device printf), in the fourth step of a chunk."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import meta_learning_pacoh_amd as M                                     # noqa: E402
from meta_learning_pacoh_amd import _lib as L                           # noqa: E402
import bench                                                           # noqa: E402

wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else 'ref_svgd'](1, 'weak', M, L)
for _ in range(2):
    wl['run'](8)
    torch.cuda.synchronize()
