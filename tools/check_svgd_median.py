import sys, numpy as np, torch
sys.path.insert(0, '.')
from meta_learning_pacoh_amd import _lib as L
bad = 0
for P in [1, 2, 3, 5, 10, 20, 23, 24, 32, 33, 45, 46, 64]:
    for dt in (torch.float32, torch.float64):
        g = torch.Generator().manual_seed(P)
        X = torch.randn(P, 37, generator=g, dtype=dt)
        if P >= 5: X[3] = X[1]          # duplicate particle -> ties / extra zeros
        s = torch.randn(P, 37, generator=g, dtype=dt)
        phi, bw, _ = L.svgd_phi(X.cuda(), s.cuda(), None)
        d2 = ((X.double().unsqueeze(0) - X.double().unsqueeze(1)) ** 2).sum(-1).numpy()
        ref = np.sqrt(np.median(d2) / (2 * np.log(P + 1)))
        err = abs(float(bw) - ref) / max(ref, 1e-30) if ref > 0 else abs(float(bw))
        if err > (1e-5 if dt == torch.float32 else 1e-12): bad += 1; print('MISMATCH', P, dt, float(bw), ref)
print('median check: %d mismatches' % bad)
