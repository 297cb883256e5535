"""posterior predictive through the HBM-resident path (pacoh_gp_predict_dense) for rocprofv3 --kernel-trace --stats:
    python tools/predict_profile.py [n] [m] [f32|f64] [B] [cov]     (defaults 512 128 f32 128, marginal variances only)"""
import sys
import torch
sys.path.insert(0, '.')
from meta_learning_pacoh_amd import _lib as L

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
m = int(sys.argv[2]) if len(sys.argv) > 2 else 128
dt = torch.float64 if (len(sys.argv) > 3 and sys.argv[3] == 'f64') else torch.float32
B = int(sys.argv[4]) if len(sys.argv) > 4 else 128
cov = len(sys.argv) > 5 and sys.argv[5] == 'cov'
d = 8
X = torch.randn(B, n, d, dtype=dt, device='cuda'); Y = torch.randn(B, n, dtype=dt, device='cuda')
Xs = torch.randn(B, m, d, dtype=dt, device='cuda')
ls = torch.full((1, d), 0.6931, dtype=dt, device='cuda'); nz = torch.tensor([0.313], dtype=dt, device='cuda')
os1 = torch.ones(1, dtype=dt, device='cuda')
run = lambda: L.gp_predict(X, 1, None, L.MEAN_ZERO, Y, 1, Xs, 1, None, ls, os1, nz, B, 1, want_cov=cov)
for _ in range(3):
    out = run()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10):
    out = run()
e.record()
torch.cuda.synchronize()
print('predict n=%d m=%d %s B=%d cov=%s: %.3f ms per call, mu mean %.6f, info max %d' % (n, m, str(dt)[6:], B, cov, s.elapsed_time(e) / 10, float(out[0].mean()), int(out[-1].max())))
