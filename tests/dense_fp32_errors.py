"""the measured norm-wise error of every fp32 gradient the HBM-resident path returns at the context sizes
tests/test_gpu_dense_path.py asserts (n = 129 ... 1024), beside the error of a plain torch fp32 CPU evaluation (autograd through
the same oracle expression in fp32) of the same problem -- both against the fp64 oracle (SURVEY section 7's method).
    python tests/dense_fp32_errors.py > profiles/r06_dense_fp32_errors.txt      (a checker script, not a collected test: it lives
under tests/ because only tests/ may import the oracle)
The problems are the tests' own (same make_problem seeds): `big` = test_dense_lml_fwdbwd_at_odd_and_large_contexts,
`edge` = test_two_level_path_edges."""
import sys
import torch
sys.path.insert(0, '.')
from meta_learning_pacoh_amd import _lib as L
from tests.test_gpu_kernels import make_problem, maxrel, oracle_mll, relerr

L.FORCE_DENSE = True
DEV = 'cuda:0'
NAMES = ['d_z', 'd_mean', 'd_ls', 'd_os', 'd_noise']


def one(tag, T, P, n, f, seed, with_gl):
    dt = torch.float32
    z, mean, y, ls, os_, noise = make_problem(T, P, n, f, dt, seed=seed, per_eval_z=True, noise_lo=0.05)
    gl = (torch.rand(T * P, dtype=dt) + 0.5) if with_gl else torch.ones(T * P, dtype=dt)
    l64 = [t.double().clone().requires_grad_(True) for t in (z, mean, ls, os_, noise)]
    ref = oracle_mll(l64[0], l64[1], y.double(), l64[2], l64[3], l64[4], T, P, True)
    (ref * gl.double()).sum().backward()
    l32 = [t.clone().requires_grad_(True) for t in (z, mean, ls, os_, noise)]
    cpu = oracle_mll(l32[0], l32[1], y, l32[2], l32[3], l32[4], T, P, True)
    (cpu * gl).sum().backward()
    out = L.gp_lml_fwdbwd(z.to(DEV), 1, mean.to(DEV), L.MEAN_VECTOR, y.to(DEV), P, ls.to(DEV), os_.to(DEV), noise.to(DEV), T * P, P,
                          g_lml=gl.to(DEV), want_dz=True)
    lml, d_z, d_mean, d_ls, d_os, d_noise, info = out
    hip = [d_z, d_mean, d_ls.reshape(T, P, f).sum(0), d_os.reshape(T, P).sum(0), d_noise.reshape(T, P).sum(0)]
    row = '%-5s n=%4d B=%d  lml: hip %.1e torch32 %.1e |' % (tag, n, T * P, maxrel(lml, ref), maxrel(cpu, ref))
    worst = 0.0
    for nm, h, a, b in zip(NAMES, hip, l32, l64):
        eh, ec = relerr(h, b.grad), relerr(a.grad, b.grad)
        worst = max(worst, eh)
        row += ' %s %.1e / %.1e |' % (nm, eh, ec)
    print(row + ' info %d  worst hip %.1e' % (int(info.abs().max()), worst), flush=True)


print('norm-wise relative error vs the fp64 oracle:  HIP fp32 / torch-CPU fp32 (same expression, autograd)')
for n in (129, 255, 513, 640, 784, 1000):
    one('big', 1, 2, n, 3, 3 * n + 1, True)
for n, B in ((1024, 1), (516, 9)):
    one('edge', B, 1, n, 2, n + B, False)


def ladder(n, ragged):
    """the healthy problem of test_dense_two_level_path_ladder_and_healthy_neighbours (fp32)"""
    from oracle import pacoh_oracle as O
    f, dt = 3, torch.float32
    gen = torch.Generator().manual_seed(n)
    z = torch.randn(n, f, generator=gen, dtype=dt)
    y = torch.randn(1, n, generator=gen, dtype=dt)
    ls, noise = torch.ones(1, f, dtype=dt), torch.tensor([0.3], dtype=dt)
    nv = n - 77 if ragged else n
    n_valid = torch.tensor([nv], dtype=torch.int32, device=DEV) if ragged else None
    out = L.gp_lml_fwdbwd(z[None].to(DEV), 1, None, L.MEAN_ZERO, y.to(DEV), 1, ls.to(DEV), None, noise.to(DEV), 1, 1, n_valid=n_valid, want_dz=True)
    row = 'ladder n=%4d nv=%4d |' % (n, nv)
    res = {}
    for tag, dd in (('torch32', dt), ('ref', torch.float64)):
        lv = [z[:nv].to(dd).clone().requires_grad_(True), ls[0].to(dd).clone().requires_grad_(True), noise[0].to(dd).clone().requires_grad_(True)]
        v = O.gp_mll(lv[0], torch.zeros(nv, dtype=dd), y[0, :nv].to(dd), lv[1], torch.tensor(1.0, dtype=dd), lv[2])
        v.backward()
        res[tag] = (v.detach(), [t.grad for t in lv])
    hip = [out[1].cpu()[0, :nv], out[3].cpu()[0], out[5].cpu()[0]]
    row += ' lml: hip %.1e torch32 %.1e |' % (abs(float(out[0][0]) - float(res['ref'][0])) / abs(float(res['ref'][0])),
                                               abs(float(res['torch32'][0]) - float(res['ref'][0])) / abs(float(res['ref'][0])))
    for nm, h, a, b in zip(('d_z', 'd_ls', 'd_noise'), hip, res['torch32'][1], res['ref'][1]):
        row += ' %s %.1e / %.1e |' % (nm, relerr(h, b), relerr(a, b))
    print(row, flush=True)


for n in (640, 1000):
    for ragged in (False, True):
        ladder(n, ragged)
