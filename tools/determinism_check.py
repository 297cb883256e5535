"""bitwise repeatability of the fused kernels over many launches (a missing hardware wait state shows up as an occasional
different bit long before it shows up in a tolerance): register-resident GP kernel, LDS-resident GP kernel, large-context path
(fp32 and fp64), fused MLP forward/backward"""
import sys
import torch
sys.path.insert(0, '.')
from meta_learning_pacoh_amd import _lib as L

def rep(name, fn, n=30):
    ref = [t.clone() for t in fn() if torch.is_tensor(t)]
    bad = 0
    for _ in range(n):
        out = [t for t in fn() if torch.is_tensor(t)]
        bad += int(any(not torch.equal(a, b) and not (torch.isnan(a) & torch.isnan(b)).all() for a, b in zip(ref, out)))
    print('%-40s %d of %d repeats differ' % (name, bad, n))
    return bad

g = torch.Generator().manual_seed(3)
total = 0
for n, T, P, dt in ((64, 256, 20, torch.float32), (48, 256, 20, torch.float32), (128, 64, 10, torch.float32), (96, 64, 10, torch.float32), (512, 32, 1, torch.float64), (300, 32, 2, torch.float32), (784, 8, 2, torch.float32)):
    f = 2
    B = T * P
    z = torch.randn(B, n, f, generator=g, dtype=dt).cuda(); mean = (0.3 * torch.randn(B, n, generator=g, dtype=dt)).cuda()
    y = torch.randn(T, n, generator=g, dtype=dt).cuda(); ls = (torch.rand(P, f, generator=g, dtype=dt) + 0.5).cuda()
    noise = (torch.rand(P, generator=g, dtype=dt) * 0.3 + 0.1).cuda()
    total += rep('gp_lml_fwdbwd n=%d %s' % (n, str(dt)[6:]), lambda: L.gp_lml_fwdbwd(z, 1, mean, L.MEAN_VECTOR, y, P, ls, None, noise, B, P))
T, P, n, d = 256, 20, 64, 4
x = torch.randn(T, n, d, generator=g).cuda()
theta = (0.3 * torch.randn(P, 2534, generator=g)).cuda()
hidden = [32, 32]
B = T * P
fwd = lambda: L.mlp2_fwd(x, P, theta, P, d, hidden, 0, 1, 1249, 2, B, n)
total += rep('mlp2_fwd (both networks)', fwd)
ga, gb = torch.randn(B, n, 1, generator=g).cuda(), torch.randn(B, n, 2, generator=g).cuda()
grad = torch.zeros(P, 2534).cuda()
def bwd():
    grad.zero_()
    L.mlp2_bwd(x, P, theta, P, d, hidden, 0, 1, ga, 1249, 2, gb, grad, False, B, n)
    return (grad,)
total += rep('mlp2_bwd (both networks)', bwd)
sys.exit(1 if total else 0)
