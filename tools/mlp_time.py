#!/usr/bin/env python
"""A/B timing of the per-particle MLP kernels at the cfg #3 shape (1024 tasks x 20 particles, n = 64, d = 4):
mean + kernel-feature network, forward and backward, per implementation (PACOH_MLP_PATH) and tile shape.
    python tools/mlp_time.py [--layers 32,32] [--reps 50]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from meta_learning_pacoh_amd import _lib as L  # noqa: E402


def timeit(fn, reps):
    for _ in range(5):
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
    ap = argparse.ArgumentParser()
    ap.add_argument('--layers', default='32,32')
    ap.add_argument('--reps', type=int, default=50)
    ap.add_argument('--tasks', type=int, default=1024)
    ap.add_argument('--particles', type=int, default=20)
    ap.add_argument('--n', type=int, default=64)
    ap.add_argument('--d', type=int, default=4)
    ap.add_argument('--quick', action='store_true', help='default tile shapes only')
    args = ap.parse_args()
    hidden = [int(v) for v in args.layers.split(',')]
    T, P, n, d = args.tasks, args.particles, args.n, args.d
    B = T * P

    def dnet(o):
        prev, c = d, 0
        for h in hidden:
            c += h * (prev + 1)
            prev = h
        return c + o * (prev + 1)
    Dm, Dk = dnet(1), dnet(2)
    D = Dm + Dk + 3
    torch.manual_seed(0)
    theta = (0.5 * torch.randn(P, D)).cuda()
    x = torch.randn(T, n, d).cuda()
    g_m, g_k = torch.randn(B, n, 1).cuda(), torch.randn(B, n, 2).cuda()
    grad = torch.zeros(P, D).cuda()
    ws = {}

    def pair_fwd():
        L.mlp2_fwd(x, P, theta, P, d, hidden, 0, 1, Dm, 2, B, n)

    def pair_bwd():
        ws['p'] = L.mlp2_bwd(x, P, theta, P, d, hidden, 0, 1, g_m, Dm, 2, g_k, grad, False, B, n, ws.get('p'))

    def stash_pair(fwd_only=False, bwd_only=False):
        st = ws['stash'] = L.mlp2_stash(x, P, d, hidden, 1, 2, B, n, ws.get('stash'))
        if not bwd_only:
            L.mlp2_fwd(x, P, theta, P, d, hidden, 0, 1, Dm, 2, B, n, stash=st)
        if not fwd_only:
            ws['p'] = L.mlp2_bwd(x, P, theta, P, d, hidden, 0, 1, g_m, Dm, 2, g_k, grad, False, B, n, ws.get('p'), stash=st)

    def two_fwd():
        L.mlp_fwd(x, P, theta, D, P, d, hidden, 1, B, n)
        L.mlp_fwd(x, P, theta[:, Dm:], D, P, d, hidden, 2, B, n)

    def two_bwd():
        ws['a'] = L.mlp_bwd(x, P, theta, D, P, d, hidden, 1, g_m, grad, D, False, B, n, ws.get('a'))
        ws['b'] = L.mlp_bwd(x, P, theta[:, Dm:], D, P, d, hidden, 2, g_k, grad[:, Dm:], D, False, B, n, ws.get('b'))

    def setenv(**kw):
        for k, v in kw.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = str(v)
        L.reload_env()

    rows = []
    for path in (None, 'mfma'):
        if path == 'mfma' and (len(hidden) > 2 or args.quick):
            continue
        setenv(PACOH_MLP_PATH=path)
        if path is None and args.quick:
            rows.append(('fused fwd pair', timeit(pair_fwd, args.reps)))
            rows.append(('fused bwd pair', timeit(pair_bwd, args.reps)))
            for ns in range(0, len(hidden) + 1):
                setenv(PACOH_MLP_STASH=ns)
                ws.clear()
                stash_pair()
                rows.append(('stash %d layer(s): fwd' % ns, timeit(lambda: stash_pair(fwd_only=True), args.reps)))
                rows.append(('stash %d layer(s): bwd' % ns, timeit(lambda: stash_pair(bwd_only=True), args.reps)))
                rows.append(('stash %d layer(s): fwd + bwd' % ns, timeit(stash_pair, args.reps)))
            setenv(PACOH_MLP_STASH=None)
            ws.clear()
        elif path is None:
            for pb in (4, 2):
                setenv(PACOH_FUSED_FWD_PB=pb)
                for tpw in (4, 8, 16, 32):
                    setenv(PACOH_FUSED_FWD_TPW=tpw)
                    rows.append(('fused fwd pair pb=%d tpw=%d' % (pb, tpw), timeit(pair_fwd, args.reps)))
            setenv(PACOH_FUSED_FWD_PB=None, PACOH_FUSED_FWD_TPW=None)
            for pb in (4, 2):
                setenv(PACOH_FUSED_BWD_PB=pb)
                ws.clear()
                rows.append(('fused bwd pair pb=%d' % pb, timeit(pair_bwd, args.reps)))
            setenv(PACOH_FUSED_BWD_PB=None)
            ws.clear()
            rows.append(('fused fwd two calls', timeit(two_fwd, args.reps)))
            rows.append(('fused bwd two calls', timeit(two_bwd, args.reps)))
        else:
            ws.clear()
            rows.append(('%s fwd two calls' % path, timeit(two_fwd, args.reps)))
            rows.append(('%s bwd two calls' % path, timeit(two_bwd, args.reps)))
    setenv(PACOH_MLP_PATH=None)
    for name, us in rows:
        print('%-36s %9.1f us' % (name, us))


if __name__ == '__main__':
    main()
