import sys, torch
sys.path.insert(0, '.')
from meta_learning_pacoh_amd import _lib as L
P, D = 20, 2534
g = torch.Generator().manual_seed(0)
X = torch.randn(P, D, generator=g).cuda(); score = torch.randn(P, D, generator=g).cuda()
mu = torch.zeros(D).cuda(); sd = torch.ones(D).cuda(); m = torch.zeros(P, D).cuda(); v = torch.zeros(P, D).cuda()
sc = torch.tensor(L.step_scalars(1.0, 1e-3, 1), dtype=torch.float32, device='cuda')
ws = None
for bw in (None, 0.7):
    for _ in range(5):
        _, ws = L.svgd_update_dev(X, score, mu, sd, 0.1, bw, 'Adam', sc, m, v, workspace=ws)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(200):
        _, ws = L.svgd_update_dev(X, score, mu, sd, 0.1, bw, 'Adam', sc, m, v, workspace=ws)
    e.record(); torch.cuda.synchronize()
    print('bandwidth', bw, ': %.2f us per call (dist + update launches)' % (s.elapsed_time(e) / 200 * 1e3))
