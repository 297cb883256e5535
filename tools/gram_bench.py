"""Gram-build micro-benchmark at the headline shape (20480 Grams of 64x64, f=2, fp32) and cfg#5 (n=512, d=8, fp64)."""
import sys
import torch
sys.path.insert(0, '.')
from meta_learning_pacoh_amd import _lib as L

dev = torch.device('cuda:0')


def run(B, P, n, f, dt, iters=30):
    z = torch.randn(B, n, f, dtype=dt, device=dev)
    ls = torch.rand(P, f, dtype=dt, device=dev) + 0.5
    lib = L.load_library()
    K = torch.empty(B, n, n, dtype=dt, device=dev)
    code = L.dtype_code(z)
    st = torch.cuda.current_stream().cuda_stream
    def go():
        lib.pacoh_gram_rbf_ard(z.data_ptr(), 1, z.data_ptr(), 1, ls.data_ptr(), None, None, 0, K.data_ptr(), B, P, n, n, f, code, st)
    for _ in range(5): go()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): go()
    b.record(); torch.cuda.synchronize()
    t = a.elapsed_time(b) / iters * 1e-3
    es = z.element_size()
    byt = B * (n * f * es + n * n * es)
    print('B=%d n=%d f=%d %s, %d launches: %.1f us  %.1f GB/s' % (B, n, f, dt, iters, t * 1e6, byt / t / 1e9))


for it in (10, 20, 50, 100, 400, 20):
    run(20480, 20, 64, 2, torch.float32, iters=it)
run(20480, 20, 64, 4, torch.float32)
run(2560, 10, 128, 2, torch.float32)
run(256, 1, 512, 8, torch.float64)
