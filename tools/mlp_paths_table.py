"""the per-particle MLP implementations (fused / mfma / layers -- and, until the table below was read, valu --, csrc/mlp*.hip) timed on the 18 network
shapes of tests/test_gpu_kernels.py (MLP_CASES) -- at the test's own tiny batch and at a production-size one (256 tasks) -- forward +
backward, fp32 and fp64.  A path that is never the fastest on a shape it alone supports has no reason to stay.
    python tools/mlp_paths_table.py > profiles/r05_mlp_paths.txt"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from meta_learning_pacoh_amd import _lib as L  # noqa: E402
from tests.test_gpu_kernels import MLP_CASES     # noqa: E402


def applicable(path, dtype, d_in, hidden, d_out):
    f32 = dtype == torch.float32
    nh = len(hidden)
    if path == 'fused':
        return f32 and 1 <= nh <= 4 and d_in <= 4 and d_out <= 2 and all(h <= 32 for h in hidden)
    if path == 'mfma':
        return f32 and 1 <= nh <= 2 and d_in <= 16 and d_out <= 8 and all(h <= 32 for h in hidden)
    if path == 'valu':
        return nh <= 3 and d_in <= 16 and d_out <= 8 and all(h <= 64 for h in hidden)
    return True


def timeit(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


def main():
    paths = [p for p in ('fused', 'mfma', 'valu', 'layers') if p != 'valu' or os.environ.get('PACOH_HAVE_VALU') == '1']
    print('%-44s %-5s %6s | %s | fastest | the only path' % ('shape (P, T, n, d_in, hidden, d_out)', 'dtype', 'tasks', ' '.join('%9s' % p for p in paths)))
    wins = {p: 0 for p in paths}
    sole = {p: 0 for p in paths}
    for case in MLP_CASES:
        P, T0, n, d_in, hidden, d_out = case
        for dtype in (torch.float32, torch.float64):
            for T in (T0, 256):
                B = T * P
                Dn = sum(h * (p + 1) for h, p in zip(list(hidden) + [d_out], [d_in] + list(hidden)))
                torch.manual_seed(0)
                theta = (0.5 * torch.randn(P, Dn, dtype=dtype)).cuda()
                x = torch.randn(T, n, d_in, dtype=dtype).cuda()
                g = torch.randn(B, n, d_out, dtype=dtype).cuda()
                grad = torch.zeros(P, Dn, dtype=dtype).cuda()
                res = {}
                for path in paths:
                    if not applicable(path, dtype, d_in, hidden, d_out):
                        continue
                    os.environ['PACOH_MLP_PATH'] = path
                    L.reload_env()
                    ws = {}

                    def run():
                        L.mlp_fwd(x, P, theta, Dn, P, d_in, list(hidden), d_out, B, n, ws_holder=ws)
                        ws['b'] = L.mlp_bwd(x, P, theta, Dn, P, d_in, list(hidden), d_out, g, grad, Dn, False, B, n, ws.get('b'))
                    res[path] = timeit(run)
                os.environ.pop('PACOH_MLP_PATH', None)
                L.reload_env()
                best = min(res, key=res.get)
                wins[best] += 1
                only = [p for p in res]
                tag = only[0] if len(only) == 1 else ''
                if tag:
                    sole[tag] += 1
                print('%-44s %-5s %6d | %s | %-7s | %s' % (str(case), 'f32' if dtype == torch.float32 else 'f64', T,
                                                         ' '.join(('%9.1f' % res[p]) if p in res else '%9s' % '-' for p in paths), best, tag))
    print('fastest (fwd + bwd, us) counts:', wins)
    print('shapes only this path takes:', sole)


if __name__ == '__main__':
    main()
