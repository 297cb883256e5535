"""odd context sizes on the large-context path: padded to the left-looking kernels' alignment (default) vs the right-looking generation
(PACOH_DENSE_PAD=0):   python tools/dense_pad_ab.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from meta_learning_pacoh_amd import _lib as L  # noqa: E402

for dtype, n in ((torch.float64, 255), (torch.float64, 401), (torch.float32, 255), (torch.float32, 401), (torch.float32, 509)):
    B, d = 128, 4
    g = torch.Generator().manual_seed(n)
    X = torch.randn(B, n, d, dtype=dtype, generator=g).cuda()
    Y = torch.randn(B, n, dtype=dtype, generator=g).cuda()
    ls = torch.full((1, d), 0.7, dtype=dtype, device='cuda')
    nz = torch.tensor([0.3], dtype=dtype, device='cuda')
    os1 = torch.ones(1, dtype=dtype, device='cuda')
    res = {}
    for pad in ('0', '1'):
        os.environ['PACOH_DENSE_PAD'] = pad
        L.reload_env()
        for _ in range(3):
            out = L.gp_lml_fwdbwd(X, 1, None, L.MEAN_ZERO, Y, 1, ls, os1, nz, B, 1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            out = L.gp_lml_fwdbwd(X, 1, None, L.MEAN_ZERO, Y, 1, ls, os1, nz, B, 1)
        torch.cuda.synchronize()
        res[pad] = ((time.perf_counter() - t0) / 20 * 1e3, out)
    a, b = res['0'][1], res['1'][1]
    err = max(float((x - y).abs().max() / (y.abs().max() + 1e-30)) for x, y in zip(a[:-1], b[:-1]) if x is not None)
    print('%s n = %d, %d problems: right-looking %.3f ms, padded to the left-looking kernels %.3f ms; max rel difference of the outputs %.1e'
          % ('fp64' if dtype == torch.float64 else 'fp32', n, B, res['0'][0], res['1'][0], err))
