"""pacoh_gp_lml_dense called through raw ctypes (no helper of the Python binding) at a context size with misaligned rows, for
    rocprofv3 --kernel-trace --stats -- python3 tools/raw_abi_dense.py [n] [f32|f64]
-> the kernel list must name chol_ll_kernel / trtri_ll_kernel (the left-looking generation): the padding happens inside the entry point"""
import ctypes
import sys

import torch

sys.path.insert(0, '.')
from meta_learning_pacoh_amd import _lib as L

n = int(sys.argv[1]) if len(sys.argv) > 1 else 509
dt = torch.float64 if (len(sys.argv) > 2 and sys.argv[2] == 'f64') else torch.float32
code = L.F64 if dt == torch.float64 else L.F32
lib = L.load_library()
B, f = 16, 4
g = torch.Generator().manual_seed(n)
z = torch.randn(B, n, f, dtype=dt, generator=g).cuda(); y = torch.randn(B, n, dtype=dt, generator=g).cuda()
ls = torch.full((1, f), 0.7, dtype=dt, device='cuda'); nz = torch.tensor([0.3], dtype=dt, device='cuda')
lml = torch.empty(B, dtype=dt, device='cuda'); d_z = torch.empty(B, n, f, dtype=dt, device='cuda'); d_ls = torch.empty(B, f, dtype=dt, device='cuda')
d_nz = torch.empty(B, dtype=dt, device='cuda'); info = torch.empty(B, dtype=torch.int32, device='cuda')
need = lib.pacoh_gp_lml_dense_workspace_bytes(B, n, f, code, 1)
ws = torch.empty(need, dtype=torch.uint8, device='cuda')
p = lambda t: ctypes.c_void_p(t.data_ptr())
for _ in range(5):
    rc = lib.pacoh_gp_lml_dense(p(z), 1, None, L.MEAN_ZERO, p(y), 1, p(ls), None, p(nz), None, None, p(lml), p(d_z), None, p(d_ls), None, p(d_nz),
                                p(info), p(ws), need, B, 1, n, f, code, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
torch.cuda.synchronize()
print('rc', rc, 'n', n, str(dt), 'lml mean %.6f' % float(lml.mean()), 'info max', int(info.max()))
