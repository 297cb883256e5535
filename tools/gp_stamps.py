"""Phase breakdown of the fused GP kernel from the diagnostic build (python -m meta_learning_pacoh_amd._build --variant stamps
-DPACOH_GP_STAMPS=1; run with PACOH_LIB=.../libpacoh_gp_stamps.so): cycles per problem and phase, summed over the waves."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from meta_learning_pacoh_amd import _lib as L           # noqa: E402

NAMES = ['loads', 'Gram build', 'diagonal blocks (+ wait)', 'panel + trailing', 'Z = L^-1', 'u / quad / logdet', 'W = Z^T Z + mirror',
         'alpha', 'gradient loop', 'reductions + stores']
T, P, f, n = 1024, 20, 2, int(sys.argv[1]) if len(sys.argv) > 1 else 64
B = T * P
g = torch.Generator().manual_seed(0)
z = torch.randn(B, n, f, generator=g).cuda()
mean = (0.3 * torch.randn(B, n, generator=g)).cuda()
y = torch.randn(T, n, generator=g).cuda()
ls = (torch.rand(P, f, generator=g) + 0.5).cuda()
noise = (torch.rand(P, generator=g) * 0.3 + 0.1).cuda()
lib = L.load_library()
fn = lib.pacoh_debug_gp_stamps
fn.argtypes, fn.restype = [ctypes.c_void_p, ctypes.c_int], ctypes.c_int
run = lambda: L.gp_lml_fwdbwd(z, 1, mean, L.MEAN_VECTOR, y, P, ls, None, noise, B, P)
for _ in range(3):
    run()
torch.cuda.synchronize()
fn(None, 1)
reps = 10
for _ in range(reps):
    run()
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 16)()
fn(buf, 0)
tot = sum(buf[:10])
print('n=%d: %.0f cycles per problem (s_memtime domain)' % (n, tot / (B * reps)))
for k, name in enumerate(NAMES):
    print('   %-28s %8.0f cycles  %5.1f %%' % (name, buf[k] / (B * reps), 100.0 * buf[k] / tot))
