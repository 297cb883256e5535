"""VERDICT r5 #5, second half: do the left-looking Cholesky / inverse share a CU between two matrices where LDS allows it?  One
HBM-resident LML+gradient workload at (dtype, n, B) from the command line, 20 calls, for `rocprofv3 --kernel-trace --stats`:
    for B in 256 512; do rocprofv3 --kernel-trace --stats --output-format csv -d out/$B -- python3 tools/dense_two_per_cu.py f32 256 $B; done
A kernel that keeps ONE workgroup per CU takes twice as long at B = 512 as at B = 256 (256 CUs); one whose LDS plan lets two workgroups
share a CU takes less than that (-> profiles/r06_dense_two_per_cu.txt)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from meta_learning_pacoh_amd import _lib as L           # noqa: E402

dt = torch.float32 if sys.argv[1] == 'f32' else torch.float64
n, B = int(sys.argv[2]), int(sys.argv[3])
g = torch.Generator().manual_seed(7)
X = torch.randn(B, n, 8, generator=g, dtype=torch.float64).to(dt).cuda()
Y = torch.randn(B, n, generator=g, dtype=torch.float64).to(dt).cuda()
ls = torch.full((1, 8), 0.6931, dtype=dt, device='cuda')
nz = torch.tensor([0.313], dtype=dt, device='cuda')
os1 = torch.ones(1, dtype=dt, device='cuda')
for _ in range(20):
    out = L.gp_lml_fwdbwd(X, 1, None, L.MEAN_ZERO, Y, 1, ls, os1, nz, B, 1)
torch.cuda.synchronize()
assert int(out[-1].max()) == 0 and bool(torch.isfinite(out[0]).all())
