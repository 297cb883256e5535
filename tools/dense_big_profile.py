"""LML + gradients through the HBM-resident path at a context size ABOVE the left-looking kernels' plan (n = 784 fp32 = the reference's
MNIST context, experiments/data_sim.py:563), for rocprofv3 --kernel-trace --stats:  python tools/dense_big_profile.py [n] [f32|f64] [B] [passes]"""
import sys
import torch
sys.path.insert(0, '.')
from meta_learning_pacoh_amd import _lib as L

n = int(sys.argv[1]) if len(sys.argv) > 1 else 784
dt = torch.float64 if (len(sys.argv) > 2 and sys.argv[2] == 'f64') else torch.float32
B = int(sys.argv[3]) if len(sys.argv) > 3 else 128
passes = int(sys.argv[4]) if len(sys.argv) > 4 else 20
d = 8
# seeded inputs: the `lml mean` column of two builds is comparable only on the same draws
gen = torch.Generator().manual_seed(1000 * n + B)
X = torch.randn(B, n, d, dtype=dt, generator=gen).cuda(); Y = torch.randn(B, n, dtype=dt, generator=gen).cuda()
ls = torch.full((1, d), 0.6931, dtype=dt, device='cuda'); nz = torch.tensor([0.313], dtype=dt, device='cuda')
os1 = torch.ones(1, dtype=dt, device='cuda')
for _ in range(3):
    out = L.gp_lml_fwdbwd(X, 1, None, L.MEAN_ZERO, Y, 1, ls, os1, nz, B, 1)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(passes):
    out = L.gp_lml_fwdbwd(X, 1, None, L.MEAN_ZERO, Y, 1, ls, os1, nz, B, 1)
e.record()
torch.cuda.synchronize()
print('n=%d %s B=%d: %.3f ms per call, lml mean %.6f, info max %d' % (n, str(dt)[6:], B, s.elapsed_time(e) / passes, float(out[0].mean()), int(out[-1].max())))
