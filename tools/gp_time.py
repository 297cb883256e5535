"""time of the fused GP LML+gradient kernel alone at a given context size: python tools/gp_time.py [n ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from meta_learning_pacoh_amd import _lib as L           # noqa: E402

T, P, f = 1024, 20, 2
for n in [int(a) for a in sys.argv[1:]] or [32, 48, 64]:
    B = T * P
    g = torch.Generator().manual_seed(0)
    z = torch.randn(B, n, f, generator=g).cuda()
    mean = (0.3 * torch.randn(B, n, generator=g)).cuda()
    y = torch.randn(T, n, generator=g).cuda()
    ls = (torch.rand(P, f, generator=g) + 0.5).cuda()
    noise = (torch.rand(P, generator=g) * 0.3 + 0.1).cuda()
    run = lambda: L.gp_lml_fwdbwd(z, 1, mean, L.MEAN_VECTOR, y, P, ls, None, noise, B, P)
    for _ in range(3):
        run()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        run()
    e.record()
    torch.cuda.synchronize()
    print('n=%3d  %.4f ms per launch (%d problems)' % (n, s.elapsed_time(e) / 20, B))
