"""cfg#5 (256 problems, n=512, d=8, fp64) LML+grad through the HBM-resident path, 60 passes (argv[2]) for rocprofv3 --kernel-trace --stats; argv[1] = f32 | f64"""
import sys
import torch
sys.path.insert(0, '.')
from meta_learning_pacoh_amd import _lib as L

dt = torch.float64 if (len(sys.argv) < 2 or sys.argv[1] != 'f32') else torch.float32
X = torch.randn(256, 512, 8, dtype=dt, device='cuda'); Y = torch.randn(256, 512, dtype=dt, device='cuda')
ls = torch.full((1, 8), 0.6931, dtype=dt, device='cuda'); nz = torch.tensor([0.313], dtype=dt, device='cuda')
os1 = torch.ones(1, dtype=dt, device='cuda')
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 60          # >= 50: warm clocks, so that the kernel sum is comparable with bench.py's ms per step
for _ in range(passes):
    out = L.gp_lml_fwdbwd(X, 1, None, L.MEAN_ZERO, Y, 1, ls, os1, nz, 256, 1)
torch.cuda.synchronize()
print(float(out[0].mean()), int(out[-1].max()))
