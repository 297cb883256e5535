"""Parity of every HIP kernel (called through the C ABI) against the CPU oracle on seeded inputs.
Tolerances are the north-star bars: <= 1e-4 rel in fp64 (we assert much tighter), <= 1e-2 rel in fp32;
gradients are judged norm-wise over the whole gradient array."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import pacoh_oracle as O


@pytest.fixture(scope='module')
def L():
    if not torch.cuda.is_available():
        pytest.skip('needs a HIP device')
    from meta_learning_pacoh_amd import _lib
    _lib.load_library()
    return _lib


DEV = 'cuda'
TOL = {torch.float32: 2e-3, torch.float64: 1e-9}       # asserted (bars: 1e-2 / 1e-4)


def relerr(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-300))


def maxrel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float(((a - b).abs() / (b.abs() + 1e-6)).max())


def make_problem(T, P, n, f, dtype, seed=0, per_eval_z=True, noise_lo=-1.0):
    g = torch.Generator().manual_seed(seed)
    B = T * P
    z = torch.randn(B if per_eval_z else T, n, f, generator=g, dtype=torch.float64)
    mean = 0.3 * torch.randn(B, n, generator=g, dtype=torch.float64)
    y = torch.randn(T, n, generator=g, dtype=torch.float64)
    ls = torch.nn.functional.softplus(torch.randn(P, f, generator=g, dtype=torch.float64))
    os_ = torch.nn.functional.softplus(torch.randn(P, generator=g, dtype=torch.float64))
    noise = torch.nn.functional.softplus(torch.randn(P, generator=g, dtype=torch.float64) + noise_lo)
    return [t.to(dtype) for t in (z, mean, y, ls, os_, noise)]


def oracle_mll(z, mean, y, ls, os_, noise, T, P, per_eval_z=True):
    """oracle on the same layout: b = t*P + p"""
    B = T * P
    n = z.shape[-2]
    zz = z if per_eval_z else z.unsqueeze(1).expand(T, P, n, z.shape[-1]).reshape(B, n, -1)
    yy = y.unsqueeze(1).expand(T, P, n).reshape(B, n)
    lsb = ls.unsqueeze(0).expand(T, P, -1).reshape(B, 1, -1)
    osb = os_.unsqueeze(0).expand(T, P).reshape(B)
    nb = noise.unsqueeze(0).expand(T, P).reshape(B)
    return O.gp_mll(zz, mean, yy, lsb, osb, nb)


# ------------------------------------------------------------------------------------------ gram
@pytest.mark.parametrize('dtype', [torch.float32, torch.float64])
@pytest.mark.parametrize('shape', [(3, 2, 64, 64, 2), (2, 3, 37, 50, 3), (1, 1, 5, 7, 1), (2, 2, 130, 128, 8)])
def test_gram(L, dtype, shape):
    T, P, n, m, f = shape
    B = T * P
    g = torch.Generator().manual_seed(1)
    z1 = torch.randn(B, n, f, generator=g, dtype=dtype)
    z2 = torch.randn(T, m, f, generator=g, dtype=dtype)
    ls = torch.rand(P, f, generator=g, dtype=dtype) + 0.5
    os_ = torch.rand(P, generator=g, dtype=dtype) + 0.5
    K = L.gram_rbf_ard(z1.to(DEV), 1, z2.to(DEV), P, ls.to(DEV), os_.to(DEV), None, False, B, P)
    z2b = z2.unsqueeze(1).expand(T, P, m, f).reshape(B, m, f)
    lsb = ls.unsqueeze(0).expand(T, P, f).reshape(B, 1, f)
    ref = os_.unsqueeze(0).expand(T, P).reshape(B, 1, 1) * O.gram_rbf_ard(z1, z2b, lsb)
    assert maxrel(K, ref) < (1e-5 if dtype == torch.float32 else 1e-12)


def test_gram_square_with_noise(L):
    B, n, f = 4, 48, 2
    z = torch.randn(B, n, f, dtype=torch.float64)
    ls, noise = torch.ones(1, f, dtype=torch.float64), torch.tensor([0.25], dtype=torch.float64)
    K = L.gram_rbf_ard(z.to(DEV), 1, z.to(DEV), 1, ls.to(DEV), None, noise.to(DEV), True, B, 1)
    ref = O.gram_rbf_ard(z, z, ls) + 0.25 * torch.eye(n, dtype=torch.float64)
    assert maxrel(K, ref) < 1e-12


@pytest.mark.parametrize('case', [(2, 3, 64, 8), (1, 2, 130, 5), (2, 1, 512, 8), (1, 1, 200, 1)])
def test_gram_one_point_set_fp64_symmetric_path(L, case):
    """pacoh_gram_rbf_ard with z1 and z2 the SAME buffer (fp64, f <= 8, n >= 64): lower tiles with the distances on the matrix cores,
    mirrored through LDS -- against the direct-difference oracle, symmetric bit for bit, with and without the noise diagonal"""
    T, P, n, f = case
    B = T * P
    g = torch.Generator().manual_seed(n)
    z = (3.0 * torch.randn(B, n, f, generator=g, dtype=torch.float64)).to(DEV)
    ls = (torch.rand(P, f, generator=g, dtype=torch.float64) + 0.5).to(DEV)
    os_ = (torch.rand(P, generator=g, dtype=torch.float64) + 0.5).to(DEV)
    noise = (torch.rand(P, generator=g, dtype=torch.float64) * 0.1 + 0.01).to(DEV)
    lsb = ls.cpu().unsqueeze(0).expand(T, P, f).reshape(B, 1, f)
    ref = os_.cpu().unsqueeze(0).expand(T, P).reshape(B, 1, 1) * O.gram_rbf_ard(z.cpu(), z.cpu(), lsb)
    K = L.gram_rbf_ard(z, 1, z, 1, ls, os_, None, False, B, P)
    assert maxrel(K, ref) < 1e-12 and torch.equal(K, K.transpose(-1, -2))
    Kn = L.gram_rbf_ard(z, 1, z, 1, ls, os_, noise, True, B, P)
    refn = ref + torch.diag_embed(noise.cpu().unsqueeze(0).expand(T, P).reshape(B, 1).expand(B, n))
    assert maxrel(Kn, refn) < 1e-12 and torch.equal(Kn, Kn.transpose(-1, -2))


# ------------------------------------------------------------------------------------------ lml fwd
CASES = [  # T, P, n, f, per_eval_z
    (4, 3, 5, 2, True),        # cfg #1 shape (demo)
    (6, 1, 32, 1, False),      # cfg #2: MAP, SE on inputs, d=1
    (3, 4, 64, 2, True),       # cfg #3 NN features
    (3, 4, 64, 4, False),      # cfg #3 SE on d=4 inputs
    (2, 2, 128, 2, True),      # cfg #4
    (2, 2, 80, 2, False),      # n > 64 (round 5: two waves per SIMD, W blocks consumed where they are produced): 6 blocks, f <= 2
    (2, 2, 96, 4, True),       # ... 6 blocks, f <= 4
    (1, 3, 112, 3, True),      # ... 8 blocks (7 used), f <= 4 (one wave per SIMD)
    (2, 2, 128, 4, True),      # ... 8 blocks, f <= 4
    (2, 2, 47, 3, True),       # odd n
    (1, 2, 9, 16, True),       # max feature dim
]


@pytest.mark.parametrize('dtype', [torch.float32, torch.float64])
@pytest.mark.parametrize('case', CASES)
def test_lml_fwd(L, dtype, case):
    T, P, n, f, pez = case
    small = n <= L.gp_small_max_n(dtype, False)          # beyond it (fp64, n = 128) the same call runs the HBM-resident path: LML only
    z, mean, y, ls, os_, noise = make_problem(T, P, n, f, dtype, seed=n + f, per_eval_z=pez)
    ref = oracle_mll(z.double(), mean.double(), y.double(), ls.double(), os_.double(), noise.double(), T, P, pez)
    lml, alpha, Lf, info = L.gp_lml_fwd(z.to(DEV), 1 if pez else P, mean.to(DEV), L.MEAN_VECTOR, y.to(DEV), P,
                                        ls.to(DEV), os_.to(DEV), noise.to(DEV), T * P, P, want_alpha=small, want_L=small)
    assert int(info.abs().max()) == 0
    assert maxrel(lml, ref) < TOL[dtype]
    if not small:
        assert alpha is None and Lf is None
        return
    # factor and alpha of problem 0
    zz = (z if pez else z.unsqueeze(1).expand(T, P, n, f).reshape(T * P, n, f)).double()
    K0 = os_[0].double() * O.gram_rbf_ard(zz[0], zz[0], ls[0].double()) + noise[0].double() * torch.eye(n, dtype=torch.float64)
    L0 = torch.linalg.cholesky(K0)
    assert relerr(Lf[0], L0) < TOL[dtype]
    a0 = torch.cholesky_solve((y[0] - mean[0]).double().unsqueeze(-1), L0).squeeze(-1)
    assert relerr(alpha[0], a0) < (5e-2 if dtype == torch.float32 else 1e-8)


def test_lml_mean_modes_and_unit_outputscale(L):
    T, P, n, f = 3, 2, 20, 2
    z, mean, y, ls, os_, noise = make_problem(T, P, n, f, torch.float64, seed=3)
    c = torch.tensor([0.3, -0.7], dtype=torch.float64)
    one = torch.ones(P, dtype=torch.float64)
    ref_c = oracle_mll(z, c.unsqueeze(0).expand(T, P).reshape(-1, 1).expand(-1, n), y, ls, one, noise, T, P)
    lml, *_ = L.gp_lml_fwd(z.to(DEV), 1, c.to(DEV), L.MEAN_CONST, y.to(DEV), P, ls.to(DEV), None, noise.to(DEV), T * P, P)
    assert maxrel(lml, ref_c) < 1e-10
    ref_0 = oracle_mll(z, torch.zeros(T * P, n, dtype=torch.float64), y, ls, one, noise, T, P)
    lml, *_ = L.gp_lml_fwd(z.to(DEV), 1, None, L.MEAN_ZERO, y.to(DEV), P, ls.to(DEV), None, noise.to(DEV), T * P, P)
    assert maxrel(lml, ref_0) < 1e-10


# ------------------------------------------------------------------------------------------ lml bwd
@pytest.mark.parametrize('dtype', [torch.float32, torch.float64])
@pytest.mark.parametrize('case', CASES)
def test_lml_fwdbwd(L, dtype, case):
    T, P, n, f, pez = case                                 # (fp64, n = 128: the dispatcher serves it through the HBM-resident path)
    z, mean, y, ls, os_, noise = make_problem(T, P, n, f, dtype, seed=7 * n + f, per_eval_z=pez)
    gl = torch.rand(T * P, dtype=dtype) + 0.5
    leaves = [t.double().clone().requires_grad_(True) for t in (z, mean, ls, os_, noise)]
    ref = oracle_mll(leaves[0], leaves[1], y.double(), leaves[2], leaves[3], leaves[4], T, P, pez)
    (ref * gl.double()).sum().backward()
    out = L.gp_lml_fwdbwd(z.to(DEV), 1 if pez else P, mean.to(DEV), L.MEAN_VECTOR, y.to(DEV), P, ls.to(DEV),
                          os_.to(DEV), noise.to(DEV), T * P, P, g_lml=gl.to(DEV), want_dz=pez)
    lml, d_z, d_mean, d_ls, d_os, d_noise, info = out
    assert int(info.abs().max()) == 0
    assert maxrel(lml, ref) < TOL[dtype]
    gtol = 1e-2 if dtype == torch.float32 else 1e-8
    if pez:
        assert relerr(d_z, leaves[0].grad) < gtol
    assert relerr(d_mean, leaves[1].grad) < gtol
    # per-problem hyper gradients summed over tasks == autograd of the shared hyper-parameters
    assert relerr(d_ls.reshape(T, P, f).sum(0), leaves[2].grad) < gtol
    assert relerr(d_os.reshape(T, P).sum(0), leaves[3].grad) < gtol
    assert relerr(d_noise.reshape(T, P).sum(0), leaves[4].grad) < gtol


def test_lml_const_mean_grad(L):
    T, P, n, f = 3, 2, 24, 2
    z, _, y, ls, os_, noise = make_problem(T, P, n, f, torch.float64, seed=11)
    c = torch.tensor([0.3, -0.7], dtype=torch.float64, requires_grad=True)
    ref = oracle_mll(z, c.unsqueeze(0).expand(T, P).reshape(-1, 1).expand(-1, n), y, ls, os_, noise, T, P)
    ref.sum().backward()
    out = L.gp_lml_fwdbwd(z.to(DEV), 1, c.detach().to(DEV), L.MEAN_CONST, y.to(DEV), P, ls.to(DEV), os_.to(DEV),
                          noise.to(DEV), T * P, P)
    assert relerr(out[2].reshape(T, P).sum(0), c.grad) < 1e-9


def test_lml_ragged_n_valid(L):
    """tasks of different size padded to a common n (reference: tasks may differ in n, random_gp.py:209-212)"""
    T, P, n, f = 4, 2, 24, 2
    sizes = [24, 5, 17, 1]
    z, mean, y, ls, os_, noise = make_problem(T, P, n, f, torch.float64, seed=5)
    nv = torch.tensor(sizes, dtype=torch.int32)
    out = L.gp_lml_fwdbwd(z.to(DEV), 1, mean.to(DEV), L.MEAN_VECTOR, y.to(DEV), P, ls.to(DEV), os_.to(DEV),
                          noise.to(DEV), T * P, P, n_valid=nv.to(DEV))
    lml, d_z, d_mean, d_ls, d_os, d_noise, info = [o.cpu() for o in out]
    for t in range(T):
        s = sizes[t]
        for p in range(P):
            b = t * P + p
            zz = z[b, :s].clone().requires_grad_(True)
            mm = mean[b, :s].clone().requires_grad_(True)
            hy = [h.clone().requires_grad_(True) for h in (ls[p], os_[p], noise[p])]
            ref = O.gp_mll(zz, mm, y[t, :s], hy[0], hy[1], hy[2])
            ref.backward()
            assert abs(float(lml[b] - ref)) < 1e-10 * max(1, abs(float(ref)))
            assert relerr(d_z[b, :s], zz.grad) < 1e-8 and float(d_z[b, s:].abs().sum()) == 0
            assert relerr(d_mean[b, :s], mm.grad) < 1e-8 and float(d_mean[b, s:].abs().sum()) == 0
            assert relerr(d_ls[b], hy[0].grad) < 1e-8
            assert relerr(d_os[b], hy[1].grad) < 1e-8 and relerr(d_noise[b], hy[2].grad) < 1e-8


def test_lml_jitter_ladder_on_rank_deficient_gram(L):
    """fault injection: identical points + ~zero noise -> plain Cholesky fails in fp32; the kernel
    must climb gpytorch's psd_safe_cholesky jitter ladder and report it in info[]"""
    n, f = 32, 2
    gen = torch.Generator().manual_seed(20)                   # fixed draw: the fp32 error of this ill-conditioned case depends on y
    z = torch.zeros(2, n, f, dtype=torch.float32)
    z[1] = torch.randn(n, f, generator=gen)                   # problem 1 is healthy
    y = torch.randn(1, n, dtype=torch.float32, generator=gen)
    ls = torch.ones(1, f, dtype=torch.float32)
    noise = torch.tensor([1e-12], dtype=torch.float32)
    # B=2 problems of ONE task with P=... use P=1, T=2 and the same y
    yy = y.expand(2, n).contiguous()
    lml, _, _, info = L.gp_lml_fwd(z.to(DEV), 1, None, L.MEAN_ZERO, yy.to(DEV), 1, ls.to(DEV), None, noise.to(DEV), 2, 1)
    info = info.cpu()
    assert int(info[0]) >= 1 and bool(torch.isfinite(lml[0]))
    jit = 1e-6 * 10 ** (int(info[0]) - 1)
    K = torch.ones(n, n, dtype=torch.float64) + (1e-12 + jit) * torch.eye(n, dtype=torch.float64)
    ref = torch.distributions.MultivariateNormal(torch.zeros(n, dtype=torch.float64), K).log_prob(y[0].double()) / n
    assert abs(float(lml[0]) - float(ref)) < 0.15 * abs(float(ref))    # fp32 at condition ~1e7: loose by nature


# ------------------------------------------------------------------------------------------ predict
@pytest.mark.parametrize('dtype', [torch.float32, torch.float64])
@pytest.mark.parametrize('case', [(2, 3, 5, 50, 2), (2, 2, 64, 130, 4), (1, 2, 128, 20, 2)])
def test_predict(L, dtype, case):
    T, P, n, m, f = case
    B = T * P
    z, mean, y, ls, os_, noise = make_problem(T, P, n, f, dtype, seed=m)
    g = torch.Generator().manual_seed(99)
    zt = torch.randn(B, m, f, generator=g, dtype=dtype)
    mt = 0.2 * torch.randn(B, m, generator=g, dtype=dtype)
    mu, var, cov, info = L.gp_predict(z.to(DEV), 1, mean.to(DEV), L.MEAN_VECTOR, y.to(DEV), P, zt.to(DEV), 1, mt.to(DEV),
                                      ls.to(DEV), os_.to(DEV), noise.to(DEV), B, P, want_cov=True)
    yy = y.unsqueeze(1).expand(T, P, n).reshape(B, n).double()
    lsb = ls.unsqueeze(0).expand(T, P, f).reshape(B, 1, f).double()
    osb = os_.unsqueeze(0).expand(T, P).reshape(B).double()
    nb = noise.unsqueeze(0).expand(T, P).reshape(B).double()
    rm, rc = O.gp_predict(z.double(), mean.double(), yy, zt.double(), mt.double(), lsb, osb, nb)
    tol = 5e-3 if dtype == torch.float32 else 1e-9
    assert relerr(mu, rm) < tol
    assert relerr(var, torch.diagonal(rc, dim1=-2, dim2=-1)) < tol
    assert relerr(cov, rc) < tol


@pytest.mark.parametrize('case', [  # T, P, n, m, f, ragged, shared test set, mean mode
    (3, 2, 5, 50, 2, False, False, 'vector'),      # the demo's shape (one block, 5 of 16 rows)
    (2, 3, 20, 37, 1, True, False, 'vector'),      # two blocks, ragged tasks, m not a multiple of 16
    (2, 2, 47, 16, 3, False, True, 'const'),       # three blocks, f <= 4, one test set per task shared by its particles
    (3, 4, 64, 130, 2, True, False, 'zero'),       # cfg #3's context size, four blocks
    (2, 2, 64, 64, 4, False, False, 'vector'),     # ... with four feature dimensions
    (2, 2, 80, 40, 2, False, False, 'vector'),     # n > 64: V = L^-1 K_xs from the registers (six-block kernel, five used)
    (1, 3, 128, 130, 2, True, True, 'const'),      # cfg #4's context size, ragged, shared test set
    (2, 2, 100, 33, 4, False, False, 'zero'),      # eight-block kernel, seven used, f <= 4
    (1, 1, 1, 1, 1, False, False, 'zero'),         # one context point, one test point
    (3, 1, 17, 1, 2, True, True, 'vector'),        # one test point per task, the second block holds one row
    (1, 5, 128, 16, 4, False, False, 'const'),     # eight blocks, f <= 4 (one wave per SIMD), five parameter rows
])
def test_predict_marginal_register_resident(L, case):
    """the marginal posterior predictive (no covariance) of an fp32 RBF GP at n <= 128, f <= 4 runs gp_reg_predict_kernel (round 5: the
    body of the LML kernel with the predictive in place of the gradients): mean and variance against the oracle over block counts,
    ragged tasks, mean modes and shared test sets; with the covariance requested the same kernel hands V = L^-1 K_xs to the covariance kernel"""
    T, P, n, m, f, ragged, shared, mm = case
    B = T * P
    dtype = torch.float32
    z, mean, y, ls, os_, noise = make_problem(T, P, n, f, dtype, seed=3 * n + m)
    g = torch.Generator().manual_seed(7)
    zt = torch.randn(T if shared else B, m, f, generator=g, dtype=dtype)
    sizes = [max(1, n - 7 * t) for t in range(T)] if ragged else [n] * T
    nv = torch.tensor(sizes, dtype=torch.int32, device=DEV) if ragged else None
    if mm == 'vector':
        mean_ctx, mode, mt = mean, L.MEAN_VECTOR, 0.2 * torch.randn(B, m, generator=g, dtype=dtype)
    elif mm == 'const':
        c = torch.tensor([0.3, -0.7, 0.1, 0.5, -0.2][:P], dtype=dtype)
        mean_ctx, mode, mt = c, L.MEAN_CONST, c
    else:
        mean_ctx, mode, mt = None, L.MEAN_ZERO, None
    dev = lambda t: None if t is None else t.to(DEV)
    mu, var, cov, info = L.gp_predict(dev(z), 1, dev(mean_ctx), mode, dev(y), P, dev(zt), P if shared else 1, dev(mt), dev(ls), dev(os_), dev(noise),
                                      B, P, n_valid=nv)
    assert cov is None and int(info.abs().max()) == 0
    mu2, var2, cov2, _ = L.gp_predict(dev(z), 1, dev(mean_ctx), mode, dev(y), P, dev(zt), P if shared else 1, dev(mt), dev(ls), dev(os_), dev(noise),
                                      B, P, n_valid=nv, want_cov=True)                # (the same kernel handing V = L^-1 K_xs to the covariance kernel)
    assert torch.equal(mu, mu2) and torch.equal(var, var2)
    assert relerr(torch.diagonal(cov2, dim1=-2, dim2=-1), var) < 2e-4
    for b in range(B):
        t, p = b // P, b % P
        k = sizes[t]
        mc = mean[b, :k] if mm == 'vector' else (mt[p].expand(k) if mm == 'const' else torch.zeros(k))
        ms = mt[b] if mm == 'vector' else (mt[p].expand(m) if mm == 'const' else torch.zeros(m))
        ztb = zt[t] if shared else zt[b]
        rm, rc = O.gp_predict(z[b:b + 1, :k].double(), mc.unsqueeze(0).double(), y[t:t + 1, :k].double(), ztb.unsqueeze(0).double(),
                              ms.unsqueeze(0).double(), ls[p].double().reshape(1, 1, f), os_[p].double().reshape(1), noise[p].double().reshape(1))
        assert relerr(mu[b], rm[0]) < 5e-3, (b, relerr(mu[b], rm[0]))
        assert relerr(var[b], torch.diagonal(rc[0])) < 5e-3, (b,)
        assert relerr(cov2[b], rc[0]) < 5e-3, (b,)


def test_predict_marginal_register_resident_failure_is_nan(L):
    """identical points with a noise term no rung of the ladder repairs: info = -1 and NaN mean / variance for that problem only"""
    n, m, f = 32, 20, 2
    z = torch.zeros(2, n, f)
    z[1] = torch.randn(n, f, generator=torch.Generator().manual_seed(1))
    y = torch.randn(1, n)
    zt = torch.randn(2, m, f)
    ls = torch.ones(2, f)
    noise = torch.tensor([-1e-2, 0.3])
    mu, var, _, info = L.gp_predict(z.to(DEV), 1, None, L.MEAN_ZERO, y.to(DEV), 2, zt.to(DEV), 1, None, ls.to(DEV), None, noise.to(DEV), 2, 2)
    assert info.cpu().tolist() == [-1, 0]
    assert bool(torch.isnan(mu[0]).all()) and bool(torch.isnan(var[0]).all())
    assert bool(torch.isfinite(mu[1]).all()) and bool((var[1] > 0).all())


# ------------------------------------------------------------------------------------------ dense
@pytest.mark.parametrize('dtype,n,B', [(torch.float32, 50, 5), (torch.float64, 50, 3), (torch.float64, 512, 4),
                                       (torch.float64, 77, 2), (torch.float32, 200, 2)])
def test_mvn_logprob_dense(L, dtype, n, B):
    g = torch.Generator().manual_seed(n)
    z = torch.randn(B, n, 8, generator=g, dtype=torch.float64)
    ls = torch.full((1, 8), 0.6931 * 3, dtype=torch.float64)
    A = O.gram_rbf_ard(z, z, ls) + 0.313 * torch.eye(n, dtype=torch.float64)
    r = torch.randn(B, n, generator=g, dtype=torch.float64)
    ref = torch.distributions.MultivariateNormal(torch.zeros(n, dtype=torch.float64), A).log_prob(r) / n
    ralpha = torch.cholesky_solve(r.unsqueeze(-1), torch.linalg.cholesky(A)).squeeze(-1)   # (torch.linalg.solve's batched LU is flaky on many-core hosts)
    logp, alpha, info = L.mvn_logprob_dense(A.to(dtype).to(DEV).contiguous(), r.to(dtype).to(DEV), 1.0 / n, want_alpha=True)
    assert int(info.abs().max()) == 0
    assert maxrel(logp, ref) < (1e-4 if dtype == torch.float32 else 1e-10)
    assert relerr(alpha, ralpha) < (1e-3 if dtype == torch.float32 else 1e-9)


def test_large_context_lml_fp64_gram_plus_dense(L):
    """config #5 shape at reduced task count: n=512, d=8, fp64, SE kernel; 1e-4 rel bar"""
    T, n, d = 3, 512, 8
    tasks = O.sinusoid_tasks_nd(T, n, d, seed0=1000)
    stats = O.compute_normalization_stats(tasks)
    xy = [O.prepare_task(x, y, stats, torch.float64) for x, y in tasks]
    X, Y = torch.stack([a for a, _ in xy]), torch.stack([b for _, b in xy])
    ls = torch.nn.functional.softplus(torch.zeros(1, d, dtype=torch.float64))
    noise = torch.nn.functional.softplus(torch.tensor([-1.0], dtype=torch.float64))
    ref = O.gp_mll(X, torch.zeros(T, n, dtype=torch.float64), Y, ls.unsqueeze(0), 1.0, noise)
    K = L.gram_rbf_ard(X.to(DEV), 1, X.to(DEV), 1, ls.to(DEV), None, noise.to(DEV), True, T, 1)
    logp, _, info = L.mvn_logprob_dense(K, Y.to(DEV), 1.0 / n)
    assert int(info.abs().max()) == 0
    assert maxrel(logp, ref) < 1e-9


# ------------------------------------------------------------------------------------------ MLP
MLP_CASES = [  # P, T, n, d_in, hidden, d_out
    (3, 4, 64, 4, (32, 32), 2),
    (3, 4, 64, 4, (32, 32), 1),
    (1, 5, 5, 1, (32, 32), 2),          # shared weights (MAP)
    (2, 3, 9, 2, (8, 12), 2),
    (2, 2, 33, 3, (64, 64), 1),
    (2, 2, 7, 2, (), 5),
    (2, 3, 300, 16, (16,), 8),
    (2, 2, 10, 2, (16, 16, 16), 2),
    (3, 5, 64, 4, (32, 32, 32, 32), 2),      # the SVGD / VI launchers' networks (experiments/meta_GPR_SVGD_base_exp.py:29-30)
    (3, 5, 64, 4, (32, 32, 32, 32), 1),
    (2, 3, 20, 1, (32, 32, 32), 2),
    (2, 7, 33, 2, (24, 32, 9, 31), 2),       # ragged widths <= 32, tiles crossing task boundaries
    (1, 2, 5, 1, (128, 128, 128, 128), 2),   # PACOH-MAP launcher (experiments/meta_GPR_mll_base_exp.py:29-30): shared 4 x 128 net
    (2, 3, 70, 4, (128, 128, 128, 128), 1),
    (2, 2, 21, 3, (40, 17, 128, 9, 64), 3),  # any layer_sizes (models.py:328-349)
    (1, 1, 1, 1, (7,) * 9, 1),               # deep and tiny
    (2, 2, 40, 20, (48,), 11),               # wide io
    (1, 3, 300, 2, (), 1),
]
MLP_PATHS = [None, 'mfma', 'layers']      # PACOH_MLP_PATH: first implementation the dispatcher may pick (round 5: the VALU path is gone)


@pytest.fixture
def mlp_path(request):
    """(the library reads its switches once at load time: pacoh_reload_env makes it look again)"""
    from meta_learning_pacoh_amd import _lib
    old = os.environ.get('PACOH_MLP_PATH')
    if request.param is None:
        os.environ.pop('PACOH_MLP_PATH', None)
    else:
        os.environ['PACOH_MLP_PATH'] = request.param
    _lib.reload_env()
    yield request.param
    if old is None:
        os.environ.pop('PACOH_MLP_PATH', None)
    else:
        os.environ['PACOH_MLP_PATH'] = old
    _lib.reload_env()


@pytest.mark.parametrize('dtype', [torch.float32, torch.float64])
@pytest.mark.parametrize('mlp_path', MLP_PATHS, indirect=True)
@pytest.mark.parametrize('case', MLP_CASES)
def test_mlp_fwd_bwd(L, dtype, case, mlp_path):
    P, T, n, d_in, hidden, d_out = case
    B = T * P
    layout = O.nn_param_layout(d_in, d_out, hidden)
    Dn = sum(layout.values())
    g = torch.Generator().manual_seed(Dn)
    pad = 3                                                    # the block sits inside a wider particle matrix
    theta_full = 0.5 * torch.randn(P, Dn + 2 * pad, generator=g, dtype=dtype)
    x = torch.randn(T, n, d_in, generator=g, dtype=dtype)
    gout = torch.randn(B, n, d_out, generator=g, dtype=dtype)
    th_dev = theta_full.to(DEV)
    out = L.mlp_fwd(x.to(DEV), P, th_dev[:, pad:], Dn + 2 * pad, P, d_in, list(hidden), d_out, B, n)
    th = theta_full[:, pad:pad + Dn].double().clone().requires_grad_(True)
    ref = torch.stack([O.mlp_vectorized_forward(x[t].double(), th, d_in, d_out, hidden) for t in range(T)])  # [T,P,n,o]
    ref_flat = ref.reshape(B, n, d_out)
    assert relerr(out, ref_flat) < (1e-5 if dtype == torch.float32 else 1e-12)
    (ref_flat * gout.double()).sum().backward()
    d_full = torch.full((P, Dn + 2 * pad), 7.0, dtype=dtype, device=DEV)
    L.mlp_bwd(x.to(DEV), P, th_dev[:, pad:], Dn + 2 * pad, P, d_in, list(hidden), d_out, gout.to(DEV),
              d_full[:, pad:], Dn + 2 * pad, False, B, n)
    d_full = d_full.cpu()
    assert float((d_full[:, :pad] - 7).abs().max()) == 0 and float((d_full[:, pad + Dn:] - 7).abs().max()) == 0
    assert relerr(d_full[:, pad:pad + Dn], th.grad) < (2e-4 if dtype == torch.float32 else 1e-10)
    # accumulate=True adds on top
    d2 = torch.ones(P, Dn, dtype=dtype, device=DEV)
    th_c = theta_full[:, pad:pad + Dn].contiguous().to(DEV)
    L.mlp_bwd(x.to(DEV), P, th_c, Dn, P, d_in, list(hidden), d_out, gout.to(DEV), d2, Dn, True, B, n)
    assert relerr(d2.cpu() - 1, th.grad) < (2e-4 if dtype == torch.float32 else 1e-10)


MLP2_CASES = [  # P, T, n, d_in, hidden, x_div_is_P
    (20, 8, 64, 4, (32, 32), True),                  # cfg #3 shape
    (10, 2, 20, 1, (32, 32, 32, 32), True),          # experiments/meta_GPR_SVGD_base_exp.py defaults: 4 x 32, 20 points, 10 particles
    (3, 3, 37, 2, (32,), True),
    (4, 5, 16, 3, (20, 32, 11), False),              # inputs per problem (x_div = 1)
    (2, 3, 9, 2, (128, 128, 128, 128), True),        # not fused: two sequential general-path calls
    (2, 2, 12, 6, (32, 32), True),                   # d_in > 4: padded-io MFMA kernels, twice
    (5, 7, 23, 4, (32, 32), True),                   # rows per particle (161) end inside a 64-point tile
    (3, 33, 64, 4, (32, 32, 32), True),              # several tiles per wave, three hidden layers
]


@pytest.mark.parametrize('dtype', [torch.float32, torch.float64])
@pytest.mark.parametrize('case', MLP2_CASES)
def test_mlp2_mean_and_kernel_network_in_one_call(L, dtype, case):
    """pacoh_mlp2_fwd / pacoh_mlp2_bwd: the mean (d_out 1) and the kernel-feature (d_out 2) network of a step, blocks inside one
    particle matrix in the reference's order (mean_nn.* first, random_gp.py:33-46) vs the oracle, and vs the single-network calls"""
    P, T, n, d_in, hidden, shared_x = case
    B = T * P
    Dm = sum(O.nn_param_layout(d_in, 1, hidden).values())
    Dk = sum(O.nn_param_layout(d_in, 2, hidden).values())
    D = Dm + Dk + 3
    g = torch.Generator().manual_seed(Dm + n)
    theta = 0.5 * torch.randn(P, D, generator=g, dtype=dtype)
    x = torch.randn(T if shared_x else B, n, d_in, generator=g, dtype=dtype)
    x_div = P if shared_x else 1
    g_m = torch.randn(B, n, 1, generator=g, dtype=dtype)
    g_k = torch.randn(B, n, 2, generator=g, dtype=dtype)
    th_dev, x_dev = theta.to(DEV), x.to(DEV)
    mean, z = L.mlp2_fwd(x_dev, x_div, th_dev, P, d_in, list(hidden), 0, 1, Dm, 2, B, n)
    th = theta.double().clone().requires_grad_(True)
    xb = (x.unsqueeze(1).expand(T, P, n, d_in) if shared_x else x.reshape(T, P, n, d_in)).double()
    ref_m = torch.stack([O.mlp_vectorized_forward(xb[t], th[:, :Dm], d_in, 1, hidden) for t in range(T)]).reshape(B, n, 1)
    ref_k = torch.stack([O.mlp_vectorized_forward(xb[t], th[:, Dm:Dm + Dk], d_in, 2, hidden) for t in range(T)]).reshape(B, n, 2)
    tol_f, tol_b = (1e-5, 2e-4) if dtype == torch.float32 else (1e-12, 1e-10)
    assert relerr(mean, ref_m) < tol_f and relerr(z, ref_k) < tol_f
    ((ref_m * g_m.double()).sum() + (ref_k * g_k.double()).sum()).backward()
    grad = torch.full((P, D), 5.0, dtype=dtype, device=DEV)
    L.mlp2_bwd(x_dev, x_div, th_dev, P, d_in, list(hidden), 0, 1, g_m.to(DEV), Dm, 2, g_k.to(DEV), grad, False, B, n)
    assert float((grad[:, Dm + Dk:] - 5).abs().max()) == 0                       # outside the two blocks: untouched
    assert relerr(grad[:, :Dm + Dk], th.grad[:, :Dm + Dk]) < tol_b
    # the forward's activation stash (round 3) replaces the backward's recomputation of the hidden layers above the first: same outputs, same
    # gradient up to the rounding of a different instruction order (none: the stashed registers are the recomputed ones)
    stash = L.mlp2_stash(x_dev, P, d_in, list(hidden), 1, 2, B, n)
    fused = dtype == torch.float32 and d_in <= 4 and len(hidden) <= 4 and max(hidden) <= 32
    # fused kernels: every hidden layer but the first is parked (one layer: nothing to park); round 6: the layer-by-layer path (fp64, wide or
    # deep networks) keeps its packed weights + hidden activations too; the d_in 5 .. 16 MFMA kernels keep nothing
    mfma_path = dtype == torch.float32 and not fused and len(hidden) <= 2 and d_in <= 16 and max(hidden) <= 32
    assert (stash is not None) == ((fused and len(hidden) > 1) or not (fused or mfma_path))
    if stash is not None:
        stash.fill_(0xff)                                                         # (NaN patterns: every block read must have been written)
        mean_s, z_s = L.mlp2_fwd(x_dev, x_div, th_dev, P, d_in, list(hidden), 0, 1, Dm, 2, B, n, stash=stash)
        assert torch.equal(mean_s, mean) and torch.equal(z_s, z)
        grad_s = torch.full((P, D), 5.0, dtype=dtype, device=DEV)
        L.mlp2_bwd(x_dev, x_div, th_dev, P, d_in, list(hidden), 0, 1, g_m.to(DEV), Dm, 2, g_k.to(DEV), grad_s, False, B, n, stash=stash)
        assert bool(torch.isfinite(grad_s).all())
        assert relerr(grad_s[:, :Dm + Dk], th.grad[:, :Dm + Dk]) < tol_b
        assert relerr(grad_s, grad) < 1e-6
    # the single-network entry points give the same numbers
    m1 = L.mlp_fwd(x_dev, x_div, th_dev, D, P, d_in, list(hidden), 1, B, n)
    z1 = L.mlp_fwd(x_dev, x_div, th_dev[:, Dm:], D, P, d_in, list(hidden), 2, B, n)
    assert relerr(m1, mean) < 1e-6 and relerr(z1, z) < 1e-6
    # ... and with their own activation stash (ONE network: pacoh_mlp_fwd_stash -> pacoh_mlp_bwd_hyper(stash)): same output, the kernel
    # network's gradient block as above, the hyper-parameter columns behind it reduced by the same call
    st1 = L.mlp_stash(x_dev, P, d_in, list(hidden), 2, B, n)
    assert (st1 is not None) == ((fused and len(hidden) > 1) or not (fused or mfma_path))
    T_h = B // P
    d_ls, d_nz = torch.randn(T_h, P, 2, generator=g, dtype=dtype).to(DEV), torch.randn(T_h, P, generator=g, dtype=dtype).to(DEV)
    for stash1 in ([None, st1] if st1 is not None else [None]):
        if stash1 is not None:
            stash1.fill_(0xff)
        z2 = L.mlp_fwd(x_dev, x_div, th_dev[:, Dm:], D, P, d_in, list(hidden), 2, B, n, stash=stash1)
        assert torch.equal(z2, z1)
        grad1 = torch.full((P, D), 5.0, dtype=dtype, device=DEV)
        L.mlp_bwd_hyper(x_dev, x_div, th_dev, Dm, P, d_in, list(hidden), 2, g_k.to(DEV), grad1, B, n, T_h, Dm + Dk, 2, -1, Dm + Dk + 2, -1,
                        d_ls, None, d_nz, None, stash=stash1)
        assert bool(torch.isfinite(grad1).all()) and float((grad1[:, :Dm] - 5).abs().max()) == 0
        assert relerr(grad1[:, Dm:Dm + Dk], th.grad[:, Dm:Dm + Dk]) < tol_b
        sig = torch.sigmoid(th_dev[:, Dm + Dk:].double())
        assert relerr(grad1[:, Dm + Dk:Dm + Dk + 2], d_ls.double().sum(0) * sig[:, :2]) < tol_b
        assert relerr(grad1[:, Dm + Dk + 2], d_nz.double().sum(0) * sig[:, 2]) < tol_b


@pytest.mark.parametrize('tag', ['v4x32_d1_o2', 'v4x32_d4_o1', 'v3x32_d2_o2', 'v4x128_d2_o2', 'v_irregular_d3_o3', 's4x128'])
@pytest.mark.parametrize('dtype', [torch.float32, torch.float64])
def test_deep_mlp_matches_reference_fixture(L, golden_dir, tag, dtype):
    """forward outputs and parameter gradients of the REAL NeuralNetworkVectorized (4 x 32, 4 x 128, irregular layer_sizes) and of
    the real shared 4 x 128 NeuralNetwork of the PACOH-MAP launcher (deep_mlp_ref.npz, made by tests/golden/make_golden.py)"""
    fx = np.load(os.path.join(golden_dir, 'deep_mlp_ref.npz'))
    if tag == 's4x128':
        layers, theta = (128, 128, 128, 128), torch.from_numpy(fx['s4x128_theta']).reshape(1, -1)
        gq = torch.from_numpy(fx['s4x128_g']).unsqueeze(0)
        ref_out, ref_grad = torch.from_numpy(fx['s4x128_out']).unsqueeze(0), torch.from_numpy(fx['s4x128_grad']).reshape(1, -1)
    else:
        layers, theta = [int(v) for v in fx[tag + '_layers']], torch.from_numpy(fx[tag + '_theta'])
        gq, ref_out, ref_grad = (torch.from_numpy(fx[tag + k]) for k in ('_g', '_out', '_grad'))
    x = torch.from_numpy(fx[tag + '_x']).unsqueeze(0)                          # one task, shared by the P parameter sets
    P, D = theta.shape
    n, d_in, d_out = x.shape[1], x.shape[2], gq.shape[-1]
    th, xd = theta.to(dtype).to(DEV).contiguous(), x.to(dtype).to(DEV)
    out = L.mlp_fwd(xd, P, th, D, P, d_in, list(layers), d_out, P, n)
    assert relerr(out, ref_out) < 2e-5
    grad = torch.empty(P, D, dtype=dtype, device=DEV)
    L.mlp_bwd(xd, P, th, D, P, d_in, list(layers), d_out, gq.to(dtype).to(DEV).contiguous(), grad, D, False, P, n)
    assert relerr(grad, ref_grad) < 2e-4


def test_mlp_matches_reference_fixture(L, golden_dir):
    """per-particle MLP vs outputs of the REAL NeuralNetworkVectorized (fixture from the reference)"""
    fx = np.load(os.path.join(golden_dir, 'random_gp_ref.npz'))
    cfg = O.GPConfig(input_dim=4, covar_module='NN', mean_module='NN')
    theta = torch.from_numpy(fx['nn_nn_d4_theta']).to(DEV)
    x = torch.from_numpy(fx['nn_nn_d4_x']).unsqueeze(0).to(DEV)          # one task
    P, D = theta.shape
    lo, _ = cfg.slices['kernel_nn.fc_1.bias']
    out = L.mlp_fwd(x, P, theta[:, lo:], D, P, 4, [32, 32], 2, P, 9)
    assert relerr(out, torch.from_numpy(fx['nn_nn_d4_kernel_out'])) < 1e-5
    out = L.mlp_fwd(x, P, theta, D, P, 4, [32, 32], 1, P, 9)
    assert relerr(out, torch.from_numpy(fx['nn_nn_d4_mean_out'])) < 1e-5


# ------------------------------------------------------------------------------------------ misc
@pytest.mark.parametrize('dtype', [torch.float32, torch.float64])
def test_softplus_prior_adam(L, dtype):
    g = torch.Generator().manual_seed(0)
    raw = torch.randn(50, generator=g, dtype=dtype) * 5
    raw[0], raw[1] = 25.0, -30.0
    out = L.softplus_fwd(raw.to(DEV), 1e-3)
    assert maxrel(out, torch.nn.functional.softplus(raw.double()) + 1e-3) < (1e-6 if dtype == torch.float32 else 1e-14)
    gg = torch.randn(50, generator=g, dtype=dtype)
    d = L.softplus_bwd(raw.to(DEV), gg.to(DEV))
    rr = raw.double().clone().requires_grad_(True)
    (torch.nn.functional.softplus(rr) * gg.double()).sum().backward()      # torch: grad passes through above the threshold
    assert relerr(d, rr.grad) < (1e-6 if dtype == torch.float32 else 1e-14)

    cfg = O.GPConfig(input_dim=2, covar_module='NN', mean_module='NN', mean_nn_layers=(8, 8), kernel_nn_layers=(8, 8))
    pm, ps = O.hyperprior_mean_std(cfg.layout)
    theta = torch.randn(7, cfg.D, generator=g, dtype=dtype)
    th = theta.double().clone().requires_grad_(True)
    ref = O.hyperprior_log_prob(th, pm, ps)
    ref.sum().backward()
    grad = torch.full((7, cfg.D), 2.0, dtype=dtype, device=DEV)
    lp = L.prior_logprob_grad(theta.to(DEV), pm.to(dtype).to(DEV), ps.to(dtype).to(DEV), grad, 0.01)
    assert maxrel(lp, ref) < (1e-5 if dtype == torch.float32 else 1e-12)
    assert relerr(grad.cpu() - 2.0, 0.01 * th.grad) < (1e-5 if dtype == torch.float32 else 1e-12)
    # the captured-step form: score := scale[0] * score + prior_factor * d log prior, the scale read from device memory
    sc0 = torch.randn(7, cfg.D, generator=g, dtype=dtype)
    sc = sc0.to(DEV)
    L.prior_score_dev(theta.to(DEV), pm.to(dtype).to(DEV), ps.to(dtype).to(DEV), sc, 0.01, torch.tensor([0.37], dtype=dtype, device=DEV))
    assert relerr(sc, 0.37 * sc0.double() + 0.01 * th.grad) < (1e-5 if dtype == torch.float32 else 1e-12)

    # AdamW: 5 steps against torch.optim.AdamW
    p0 = torch.randn(300, generator=g, dtype=dtype)
    pt = p0.clone().requires_grad_(True)
    opt = torch.optim.AdamW([pt], lr=1e-2, weight_decay=0.2)
    pd, m, v = p0.to(DEV), torch.zeros(300, dtype=dtype, device=DEV), torch.zeros(300, dtype=dtype, device=DEV)
    for step in range(1, 6):
        gr = torch.randn(300, generator=g, dtype=dtype)
        pt.grad = gr.clone()
        opt.step()
        L.adam_step(pd, gr.to(DEV), m, v, 1e-2, step, weight_decay=0.2)
    assert maxrel(pd, pt.detach()) < (2e-6 if dtype == torch.float32 else 1e-13)


def test_svgd_phi_matches_reference_fixture(L, golden_dir):
    """phi vs the output of the REAL meta_learn/svgd.py (SVGD.phi + RBF_Kernel), fixed and median bandwidth"""
    fx = np.load(os.path.join(golden_dir, 'svgd_ref.npz'))
    for tag in ['small_median', 'small_fixed', 'cfg3_median', 'cfg3_fixed', 'se_median']:
        bw_arg = float(fx[tag + '_bw_arg'])
        X, score = torch.from_numpy(fx[tag + '_X']).to(DEV), torch.from_numpy(fx[tag + '_score']).to(DEV)
        phi, bw, _ = L.svgd_phi(X, score, None if bw_arg < 0 else bw_arg)
        assert abs(float(bw) - float(fx[tag + '_bw'])) < 1e-5 * float(fx[tag + '_bw'])
        ref = torch.from_numpy(fx[tag + '_phi'])
        assert float((phi.cpu() - ref).abs().max()) < 1e-2 * float(ref.abs().max()), tag     # fp32 bar
        assert relerr(phi, ref) < 1e-4, tag
        X64, s64 = X.double(), torch.from_numpy(fx[tag + '_score64']).to(DEV)
        phi64, _, _ = L.svgd_phi(X64, s64, None if bw_arg < 0 else bw_arg, neg=True)
        assert relerr(-phi64, torch.from_numpy(fx[tag + '_phi64'])) < 1e-9, tag


def test_reduce_tasks(L):
    a = torch.randn(13, 4, 9, dtype=torch.float64)
    out = torch.ones(4, 9, dtype=torch.float64, device=DEV)
    L.reduce_tasks(a.to(DEV), out, scale=0.5, accumulate=True)
    assert relerr(out, 1 + 0.5 * a.sum(0)) < 1e-13


def test_vi_sample_and_grad_match_reference_and_autograd(L, golden_dir):
    """theta / log q vs RandomGPPosterior.rsample / .log_prob of the real reference (fixture), ELBO gradient vs autograd"""
    fx = np.load(os.path.join(golden_dir, 'random_gp_ref.npz'))
    loc, scale = torch.from_numpy(fx['vi_init_loc']), torch.from_numpy(fx['vi_init_scale'])
    theta_ref, logq_ref = torch.from_numpy(fx['vi_rsample']), torch.from_numpy(fx['vi_logq'])
    eps = (theta_ref - loc) / torch.exp(scale)
    post = torch.stack([loc, scale]).to(DEV)
    theta, log_q = L.vi_sample(post, eps.to(DEV).contiguous())
    assert relerr(theta, theta_ref) < 1e-6 and relerr(log_q, logq_ref) < 1e-5
    S, D = 5, 37
    g = torch.Generator().manual_seed(4)
    post64 = torch.randn(2, D, generator=g, dtype=torch.float64) * 0.3
    eps64 = torch.randn(S, D, generator=g, dtype=torch.float64)
    A = torch.randn(D, D, generator=g, dtype=torch.float64)
    leaf = post64.clone().requires_grad_(True)
    th = leaf[0] + torch.exp(leaf[1]) * eps64
    logp = -0.5 * ((th @ A) ** 2).sum(-1)                           # any differentiable log-density
    logq = (-0.5 * eps64 ** 2 - leaf[1] - 0.5 * np.log(2 * np.pi)).sum(-1)
    (-(logp - 0.01 * logq).mean()).backward()
    thd = th.detach().clone().requires_grad_(True)
    score = torch.autograd.grad((-0.5 * ((thd @ A) ** 2).sum(-1)).sum(), thd)[0]
    grad = L.vi_grad(post64.to(DEV), eps64.to(DEV), score.to(DEV), 0.01)
    assert relerr(grad, leaf.grad) < 1e-12
    y = torch.ones(10, dtype=torch.float64, device=DEV)
    L.axpy(y, torch.arange(10, dtype=torch.float64, device=DEV), -0.5)
    assert relerr(y, 1 - 0.5 * torch.arange(10, dtype=torch.float64)) < 1e-15


@pytest.mark.parametrize('tag', ['small_median', 'small_fixed', 'tiny_median', 'cfg3_median', 'p20_fixed', 'p10_median'])
def test_svgd_phi_imq_matches_reference_fixture(L, golden_dir, tag):
    """phi vs the output of the REAL meta_learn/svgd.py (SVGD.phi + IMQSteinKernel, svgd.py:63-97), including the
    gradient that the reference's autograd sends through the per-dimension median bandwidth"""
    fx = np.load(os.path.join(golden_dir, 'svgd_imq_ref.npz'))
    bw_arg = float(fx[tag + '_bw_arg'])
    bw = None if bw_arg < 0 else bw_arg
    for dt, sfx, tol in ((torch.float32, '', 2e-5), (torch.float64, '64', 1e-11)):
        X, mu, s = (torch.from_numpy(fx[tag + k]).to(dt).cuda() for k in ('_X', '_mu', '_s'))
        score = (-(X - mu) / s ** 2).contiguous()
        phi, h, _ = L.svgd_phi_imq(X, score, 0.5, -0.5, bw)
        assert relerr(phi, torch.from_numpy(fx[tag + '_phi' + sfx])) < tol, (tag, dt)
        nphi, _, _ = L.svgd_phi_imq(X, score, 0.5, -0.5, bw, neg=True)
        assert torch.equal(nphi, -phi)
        if bw is None:                               # bandwidths: exact order statistics of the squared differences
            h_o, _, _ = O.svgd_imq_bandwidth(X.cpu())
            assert relerr(h, h_o) < (1e-6 if dt == torch.float32 else 1e-14)


def test_svgd_phi_imq_other_exponents_and_limits(L):
    g = torch.Generator().manual_seed(4)
    X = torch.randn(33, 101, generator=g, dtype=torch.float64)
    score = torch.randn(33, 101, generator=g, dtype=torch.float64)
    for alpha, beta, bw in ((1.3, -1.0, None), (0.2, -0.25, 0.7)):
        phi, _, _ = L.svgd_phi_imq(X.cuda(), score.cuda(), alpha, beta, bw)
        phi_o, _ = O.svgd_phi_imq_closed_form(X, score, alpha, beta, bw)
        assert relerr(phi, phi_o) < 1e-11
    with pytest.raises(RuntimeError):
        L.svgd_phi_imq(X.cuda(), score.cuda(), -1.0, -0.5, None)           # alpha must be positive (svgd.py:72)
    with pytest.raises(RuntimeError):
        L.svgd_phi_imq(torch.zeros(1025, 8).cuda(), torch.zeros(1025, 8).cuda())  # P <= PACOH_SVGD_MAX_PARTICLES


@pytest.mark.parametrize('tag', ['p80_median', 'p130_fixed', 'p200_median', 'p65_median_wide'])
def test_svgd_phi_imq_beyond_64_particles_matches_reference_fixture(L, golden_dir, tag):
    """IMQ-SVGD with 65 - 200 particles (the reference has no limit, svgd.py:63-99) vs phi of the REAL meta_learn/svgd.py
    (svgd_imq_large_ref.npz, made by tests/golden/make_golden.py imq_large): pair enumeration on the fly instead of a pair table,
    fewer dimensions per block in the phi kernel"""
    fx = np.load(os.path.join(golden_dir, 'svgd_imq_large_ref.npz'))
    bw_arg = float(fx[tag + '_bw_arg'])
    bw = None if bw_arg < 0 else bw_arg
    for dt, sfx, tol in ((torch.float32, '', 5e-5), (torch.float64, '64', 1e-11)):
        X, mu, s = (torch.from_numpy(fx[tag + k]).to(dt).cuda() for k in ('_X', '_mu', '_s'))
        score = (-(X - mu) / s ** 2).contiguous()
        phi, h, _ = L.svgd_phi_imq(X, score, 0.5, -0.5, bw)
        assert relerr(phi, torch.from_numpy(fx[tag + '_phi' + sfx])) < tol, (tag, dt)
        if bw is None:
            h_o, _, _ = O.svgd_imq_bandwidth(X.cpu())
            assert relerr(h, h_o) < (1e-6 if dt == torch.float32 else 1e-14)


def test_svgd_phi_imq_many_particles_vs_oracle(L):
    """600 particles (fp64): the phi kernel runs 8 dimensions per block there; against the oracle's closed form"""
    g = torch.Generator().manual_seed(8)
    X = torch.randn(600, 37, generator=g, dtype=torch.float64)
    score = torch.randn(600, 37, generator=g, dtype=torch.float64)
    phi, h, _ = L.svgd_phi_imq(X.cuda(), score.cuda(), 0.5, -0.5, None)
    phi_o, h_o = O.svgd_phi_imq_closed_form(X, score, 0.5, -0.5, None)
    assert relerr(phi, phi_o) < 1e-10 and relerr(h, h_o) < 1e-13


def test_vi_full_covariance_matches_reference_fixture(L, golden_dir):
    """theta / log q / ELBO gradient vs RandomGPPosterior(cov_type='full') of the real reference (fixture)"""
    fx = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(golden_dir, 'vi_full_ref.npz')).items()}
    D = fx['init_loc'].shape[0]
    post = torch.cat([fx['init_loc'].reshape(1, D), fx['tril']]).to(DEV).contiguous()
    eps = torch.linalg.solve_triangular(torch.tril(fx['tril']).double(), (fx['rsample'] - fx['init_loc']).double().t(),
                                        upper=False).t().float().contiguous()
    theta, log_q = L.vi_sample(post, eps.to(DEV), full=True)
    assert relerr(theta, fx['rsample']) < 1e-6 and relerr(log_q, fx['logq']) < 1e-5
    score = (-(theta.cpu() @ fx['A'].t()) @ fx['A']).contiguous()
    grad = L.vi_grad(post, eps.to(DEV), score.to(DEV), 0.01, full=True)
    assert relerr(grad[0], fx['grad_loc']) < 1e-5 and relerr(grad[1:], fx['grad_tril']) < 1e-5
    assert float(torch.triu(grad[1:], 1).abs().max()) == 0.0


@pytest.mark.parametrize('dtype', [torch.float32, torch.float64])
def test_vi_full_covariance_headline_size(L, dtype):
    S, D = 10, 2534
    g = torch.Generator().manual_seed(8)
    loc = torch.randn(D, generator=g, dtype=dtype) * 0.1
    tril = torch.diag(torch.rand(D, generator=g, dtype=dtype) * 0.05 + 0.05) + 0.001 * torch.randn(D, D, generator=g, dtype=dtype)
    eps, score = torch.randn(S, D, generator=g, dtype=dtype), torch.randn(S, D, generator=g, dtype=dtype)
    post = torch.cat([loc.reshape(1, D), tril]).to(DEV)
    theta, log_q = L.vi_sample(post, eps.to(DEV), full=True)
    th_o, lq_o = O.vi_full_sample(loc.double(), tril.double(), eps.double())
    tol = 1e-5 if dtype == torch.float32 else 1e-13
    assert relerr(theta, th_o) < tol and relerr(log_q, lq_o) < tol
    grad = L.vi_grad(post, eps.to(DEV), score.to(DEV), 0.01, full=True)
    gl, gt = O.vi_full_grad(tril.double(), eps.double(), score.double(), 0.01)
    assert relerr(grad[0], gl) < tol and relerr(grad[1:], gt) < tol


def test_gather_tasks(L):
    """pacoh_gather_tasks == three index_selects (task batch of a step, drawn with replacement)"""
    g = torch.Generator().manual_seed(2)
    for dtype in (torch.float32, torch.float64):
        x = torch.randn(11, 9, 3, generator=g, dtype=dtype).to(DEV)
        y = torch.randn(11, 9, generator=g, dtype=dtype).to(DEV)
        nv = torch.randint(1, 10, (11,), generator=g, dtype=torch.int32).to(DEV)
        idx = torch.tensor([10, 0, 3, 3, 7, 10], dtype=torch.int64, device=DEV)
        ox, oy, onv = L.gather_tasks(x, y, nv, idx)
        assert torch.equal(ox, x.index_select(0, idx)) and torch.equal(oy, y.index_select(0, idx)) and torch.equal(onv, nv.index_select(0, idx))
        ox2, oy2, onv2 = L.gather_tasks(x, y, None, idx)
        assert onv2 is None and torch.equal(ox2, ox) and torch.equal(oy2, oy)


def test_allreduce_sum_single_rank_communicator(L):
    """pacoh_comm_* / pacoh_allreduce_sum (RCCL on the launch stream): a one-rank communicator sums to itself, stays ordered with
    the kernels enqueued around it, and a second communicator can coexist; bad arguments are refused before RCCL is touched"""
    from meta_learning_pacoh_amd import parallel
    comm = parallel.RcclComm()
    assert comm.world_size == 1 and comm.handle.value
    for dtype in (torch.float32, torch.float64):
        buf, score, lik = parallel.packed_score_buffer(20, 2534, dtype, DEV)
        ref = torch.randn(buf.numel(), dtype=dtype, generator=torch.Generator().manual_seed(4)).to(DEV)
        buf.copy_(ref)
        L.axpy(buf, ref, 1.0)                       # kernel before, collective, kernel after: one stream, no host sync
        comm.all_reduce_(buf)
        L.axpy(buf, ref, -1.0)
        assert torch.equal(buf, ref)
        assert torch.equal(score, ref[:20 * 2534].view(20, 2534)) and torch.equal(lik, ref[20 * 2534:])
    other = parallel.RcclComm()
    other.all_reduce_(buf)
    other.close()
    lib = L.load_library()
    assert lib.pacoh_allreduce_sum(None, 4, 0, comm.handle, None) == -1
    assert lib.pacoh_allreduce_sum(buf.data_ptr(), 4, 7, comm.handle, None) == -3
    assert lib.pacoh_allreduce_sum(buf.data_ptr(), 4, 0, None, None) == -1
    comm.close()


@pytest.mark.parametrize('dtype', [torch.float32, torch.float64])
@pytest.mark.parametrize('optimizer,bandwidth,with_prior', [('Adam', None, True), ('SGD', 0.8, True), ('Adam', 1.7, False)])
def test_svgd_update_fused(L, dtype, optimizer, bandwidth, with_prior):
    """pacoh_svgd_update (prior score + phi + optimizer in one kernel) == oracle phi on the prior-augmented score + torch optimizer"""
    g = torch.Generator().manual_seed(17)
    P, D, pf, lr = 7, 301, 0.3, 1e-2
    X = torch.randn(P, D, generator=g, dtype=torch.float64)
    score = torch.randn(P, D, generator=g, dtype=torch.float64)
    mu, sd = torch.randn(D, generator=g, dtype=torch.float64), torch.rand(D, generator=g, dtype=torch.float64) + 0.5
    Xo = X.clone().requires_grad_(True)
    opt = torch.optim.Adam([Xo], lr=lr) if optimizer == 'Adam' else torch.optim.SGD([Xo], lr=lr)
    Xd, m, v = X.to(dtype).to(DEV), torch.zeros(P, D, dtype=dtype, device=DEV), torch.zeros(P, D, dtype=dtype, device=DEV)
    ws = None
    for step in (1, 2, 3):
        s_tot = score + (pf * (-(Xo.detach() - mu) / sd ** 2) if with_prior else 0.0)
        phi, bw_o = O.svgd_phi_closed_form(Xo.detach(), s_tot, bandwidth)
        Xo.grad = -phi
        opt.step()
        Xd, bw, ws = L.svgd_update(Xd, score.to(dtype).to(DEV), mu.to(dtype).to(DEV) if with_prior else None,
                                   sd.to(dtype).to(DEV) if with_prior else None, pf, bandwidth, optimizer, lr, step, m, v, workspace=ws)
        assert abs(float(bw) - float(bw_o)) < 1e-5 * float(bw_o)
    tol = 2e-4 if dtype == torch.float32 else 1e-10
    assert relerr(Xd, Xo.detach()) < tol


@pytest.mark.parametrize('P', [65, 100, 200])
def test_svgd_many_particles(L, P):
    """more than 64 particles: the median comes from the bisection kernel instead of the register sort -- phi and the fused update
    against the oracle (numpy.median of the full P x P matrix), odd and even P, fp32 and fp64"""
    g = torch.Generator().manual_seed(P)
    D, pf, lr = 77, 0.3, 1e-2
    X = torch.randn(P, D, generator=g, dtype=torch.float64)
    score = torch.randn(P, D, generator=g, dtype=torch.float64)
    mu, sd = torch.randn(D, generator=g, dtype=torch.float64), torch.rand(D, generator=g, dtype=torch.float64) + 0.5
    phi_o, bw_o = O.svgd_phi_closed_form(X, score, None)
    for dtype, tol in ((torch.float64, 1e-10), (torch.float32, 2e-4)):
        phi, bw, _ = L.svgd_phi(X.to(dtype).to(DEV), score.to(dtype).to(DEV), None)
        assert abs(float(bw) - float(bw_o)) < (1e-12 if dtype == torch.float64 else 1e-5) * float(bw_o)
        assert relerr(phi, phi_o) < tol
        # fused step (prior score + phi + SGD), host scalars and device scalars
        s_tot = score + pf * (-(X - mu) / sd ** 2)
        phi_p, _ = O.svgd_phi_closed_form(X, s_tot, None)
        args = [t.to(dtype).to(DEV) for t in (X, score, mu, sd)]
        m, v = torch.zeros(P, D, dtype=dtype, device=DEV), torch.zeros(P, D, dtype=dtype, device=DEV)
        Xn, bw2, _ = L.svgd_update(args[0], args[1], args[2], args[3], pf, None, 'SGD', lr, 1, m, v)
        assert relerr(Xn, X + lr * phi_p) < tol and abs(float(bw2) - float(bw_o)) < 1e-5 * float(bw_o)
        Xd = args[0].clone()
        sc = torch.tensor(L.step_scalars(1.0, lr, 1), dtype=dtype, device=DEV)
        L.svgd_update_dev(Xd, args[1], args[2], args[3], pf, None, 'SGD', sc, m, v)
        assert relerr(Xd, X + lr * phi_p) < tol


@pytest.mark.parametrize('dtype', [torch.float32, torch.float64])
@pytest.mark.parametrize('P,optimizer,kernel', [(7, 'Adam', 'RBF'), (70, 'SGD', 'RBF'), (5, 'Adam', 'COS')])
def test_pipelined_step_entry_points(L, dtype, P, optimizer, kernel):
    """pacoh_svgd_dist_advance + pacoh_svgd_update_next (csrc/step_tail.h) through the ABI: counter protocol, the update against the
    oracle's phi + torch optimizer with the scalars of the row the counter selects, softplus of the updated hyper-parameter entries,
    the NEXT row's scalars and gathered (ragged) task batch -- two consecutive steps"""
    from types import SimpleNamespace
    g = torch.Generator().manual_seed(23 + P)
    D, pf, T, n, d, tb, f = 61, 0.3, 9, 6, 2, 4, 3
    off_ls, off_os, off_noise, floor = 50, 57, 58, 1e-3
    X = torch.randn(P, D, generator=g, dtype=torch.float64)
    mu, sd = torch.randn(D, generator=g, dtype=torch.float64), torch.rand(D, generator=g, dtype=torch.float64) + 0.5
    tasks = SimpleNamespace(x=torch.randn(T, n, d, generator=g, dtype=torch.float64).to(dtype).to(DEV),
                            y=torch.randn(T, n, generator=g, dtype=torch.float64).to(dtype).to(DEV),
                            n_valid=torch.randint(1, n + 1, (T,), generator=g).to(torch.int32).to(DEV), ragged=True)
    rows = 3
    idx_all = torch.randint(0, T, (rows, tb), generator=g).to(DEV)
    sc_rows = [L.step_scalars(0.5 + 0.1 * k, 1e-2 * 0.9 ** k, k + 1) for k in range(rows)]
    sc_all = torch.tensor(sc_rows, dtype=dtype, device=DEV)
    feed = SimpleNamespace(tb=tb, idx_all=idx_all, sc_all=sc_all, ctr=torch.full((1,), -1, dtype=torch.int64, device=DEV),
                           sc2=torch.zeros(2, L.SC_COUNT, dtype=dtype, device=DEV),
                           batch=SimpleNamespace(x=torch.zeros(tb, n, d, dtype=dtype, device=DEV), y=torch.zeros(tb, n, dtype=dtype, device=DEV),
                                                 n_valid=torch.zeros(tb, dtype=torch.int32, device=DEV)),
                           hyp=(torch.zeros(P, f, dtype=dtype, device=DEV), torch.zeros(P, dtype=dtype, device=DEV),
                                torch.zeros(P, dtype=dtype, device=DEV)))
    feed.sc2[0] = sc_all[0]                                   # (what the chunk's prologue leaves)
    hyper = (off_ls, f, off_os, off_noise, floor, L.KERNEL_COSINE if kernel == 'COS' else L.KERNEL_RBF)
    Xo = X.clone().requires_grad_(True)
    opt = torch.optim.Adam([Xo], lr=1.0) if optimizer == 'Adam' else torch.optim.SGD([Xo], lr=1.0)
    Xd, m, v = X.to(dtype).to(DEV), torch.zeros(P, D, dtype=dtype, device=DEV), torch.zeros(P, D, dtype=dtype, device=DEV)
    ws, bw_out = L.svgd_update_workspace(Xd), torch.zeros(1, dtype=dtype, device=DEV)
    tol = 3e-4 if dtype == torch.float32 else 1e-10
    for k in range(2):
        score = torch.randn(P, D, generator=g, dtype=torch.float64)
        L.svgd_dist_advance(Xd, ws, feed.ctr)
        assert int(feed.ctr) == k
        d2 = ws.view(dtype)[:P * P].reshape(P, P).double().cpu()
        assert relerr(d2, torch.cdist(Xo.detach(), Xo.detach()) ** 2) < (1e-5 if dtype == torch.float32 else 1e-12)
        s_tot = sc_rows[k][0] * score + pf * (-(Xo.detach() - mu) / sd ** 2)
        phi, bw_o = O.svgd_phi_closed_form(Xo.detach(), s_tot, None)
        for grp in opt.param_groups:
            grp['lr'] = sc_rows[k][1]
        Xo.grad = -phi
        opt.step()
        ahead = P <= 64 and k == 1                             # second step: the bandwidth comes from the workgroup riding in hyper_bwd
        if ahead:
            Tt = 3
            dl, dn = torch.randn(Tt, P, f, dtype=dtype, device=DEV), torch.randn(Tt, P, dtype=dtype, device=DEV)
            L.hyper_bwd(Xd, Tt, off_ls, f, -1, off_noise, -1, dl, None, dn, None, torch.zeros(P, D, dtype=dtype, device=DEV),
                        kernel=hyper[5], svgd_bw=(ws, P, D))
            slot = ws.view(dtype)[P * P + P * D + 2]
            assert abs(float(slot) - float(bw_o)) < 1e-5 * float(bw_o)
        L.svgd_update_next(Xd, score.to(dtype).to(DEV), mu.to(dtype).to(DEV), sd.to(dtype).to(DEV), pf, None, optimizer, m, v, ws,
                           bw_out, feed, tasks, hyper, bandwidth_ready=ahead)
        assert int(feed.ctr) == k                             # (the update reads the counter, the next forward advances it)
        assert abs(float(bw_out) - float(bw_o)) < 1e-5 * float(bw_o)
        assert relerr(Xd, Xo.detach()) < tol
        # hyper-parameters of the UPDATED particles, from the very values the kernel stored
        sp = torch.nn.functional.softplus
        ls_ref = sp(Xd[:, off_ls:off_ls + 1]).expand(P, f) if kernel == 'COS' else sp(Xd[:, off_ls:off_ls + f])
        assert relerr(feed.hyp[0], ls_ref) < 1e-6 and relerr(feed.hyp[1], sp(Xd[:, off_os])) < 1e-6
        assert relerr(feed.hyp[2], sp(Xd[:, off_noise]) + floor) < 1e-6
        # the next row: scalars in the other ping-pong row, tasks gathered
        assert torch.equal(feed.sc2[(k + 1) & 1], sc_all[k + 1]) and torch.equal(feed.sc2[k & 1], sc_all[k])
        nxt = idx_all[k + 1]
        assert torch.equal(feed.batch.x, tasks.x[nxt]) and torch.equal(feed.batch.y, tasks.y[nxt])
        assert torch.equal(feed.batch.n_valid, tasks.n_valid[nxt])


@pytest.mark.parametrize('dtype', [torch.float32, torch.float64])
@pytest.mark.parametrize('kernel', ['RBF', 'COS'])
def test_step_begin_vi_equals_sample_plus_hyper_fwd(L, dtype, kernel):
    """pacoh_step_begin_vi: the step's samples, log q and transformed hyper-parameters from the SAME launch that selects the step's
    row and gathers its tasks -- bit for bit what pacoh_vi_sample + pacoh_hyper_fwd + pacoh_step_begin give, for row 0 and row 1"""
    from types import SimpleNamespace
    g = torch.Generator().manual_seed(41)
    S, D, T, n, d, tb, f = 6, 300, 7, 5, 2, 3, 3
    off_ls, off_noise, floor = 200, 290, 1e-3
    code = L.KERNEL_COSINE if kernel == 'COS' else L.KERNEL_RBF
    post = torch.randn(2, D, generator=g, dtype=torch.float64).to(dtype).to(DEV)
    tasks = SimpleNamespace(x=torch.randn(T, n, d, generator=g, dtype=torch.float64).to(dtype).to(DEV),
                            y=torch.randn(T, n, generator=g, dtype=torch.float64).to(dtype).to(DEV), n_valid=None, ragged=False)
    rows = 2
    feed = SimpleNamespace(tb=tb, idx_all=torch.randint(0, T, (rows, tb), generator=g).to(DEV),
                           sc_all=torch.rand(rows, L.SC_COUNT, generator=g, dtype=torch.float64).to(dtype).to(DEV),
                           aux_all=torch.randn(rows, S, D, generator=g, dtype=torch.float64).to(dtype).to(DEV),
                           ctr=torch.zeros(1, dtype=torch.int64, device=DEV), ticket=torch.zeros(1, dtype=torch.int32, device=DEV),
                           sc=torch.zeros(L.SC_COUNT, dtype=dtype, device=DEV), aux=torch.zeros(S, D, dtype=dtype, device=DEV))
    for row in range(rows):
        feed.ctr.fill_(row)
        out = (torch.zeros(tb, n, d, dtype=dtype, device=DEV), torch.zeros(tb, n, dtype=dtype, device=DEV), None)
        theta, log_q = torch.zeros(S, D, dtype=dtype, device=DEV), torch.zeros(S, dtype=dtype, device=DEV)
        hyp = (torch.zeros(S, f, dtype=dtype, device=DEV), None, torch.zeros(S, dtype=dtype, device=DEV))
        L.step_begin_vi(feed, tasks, out, post, theta, log_q, (off_ls, f, -1, off_noise, floor, code), hyp)
        th_ref, lq_ref = L.vi_sample(post, feed.aux_all[row].contiguous())
        ls_ref, _, noise_ref = L.hyper_fwd(th_ref, off_ls, f, -1, off_noise, floor, kernel=code)
        assert torch.equal(theta, th_ref) and torch.equal(log_q, lq_ref)
        assert torch.equal(hyp[0], ls_ref) and torch.equal(hyp[2], noise_ref)
        assert torch.equal(feed.aux, feed.aux_all[row]) and torch.equal(feed.sc, feed.sc_all[row])
        assert torch.equal(out[0], tasks.x[feed.idx_all[row]]) and torch.equal(out[1], tasks.y[feed.idx_all[row]])
        assert int(feed.ctr) == row                           # (advance = False: the step's update launch advances the counter)


# ------------------------------------------------------------------------------------------ predictive cdf / quantiles / calibration
def test_mixture_cdf_icdf_calib_match_reference_fixture(L, golden_dir):
    """pacoh_mixture_cdf / _icdf / pacoh_calib_error vs EqualWeightedMixtureDist.cdf / .icdf, AffineTransformedDistribution and
    _calib_error of the REAL reference (fixture mixture_quantiles_ref.npz)"""
    fx = np.load(os.path.join(golden_dir, 'mixture_quantiles_ref.npz'))
    ym, ys = float(fx['y_mean'][0]), float(fx['y_std'][0])
    for dtype in (torch.float32, torch.float64):
        mus = torch.from_numpy(fx['mus']).to(dtype).to(DEV)
        var = (torch.from_numpy(fx['sig']).to(dtype) ** 2).to(DEV)
        val, q = torch.from_numpy(fx['val']).to(dtype).to(DEV), torch.from_numpy(fx['q']).to(dtype).to(DEV)
        cdf = L.mixture_cdf(mus, var, val, ym, ys)
        np.testing.assert_allclose(cdf.cpu().numpy(), fx['mix_cdf'], rtol=2e-5, atol=2e-7)
        np.testing.assert_allclose(L.mixture_icdf(mus, var, q, ym, ys).cpu().numpy(), fx['mix_icdf'], rtol=0, atol=1e-5)
        for key, level in (('mix_icdf_05', 0.05), ('mix_icdf_95', 0.95)):
            x = L.mixture_icdf(mus, var, torch.full((40,), level, dtype=dtype, device=DEV), ym, ys)
            np.testing.assert_allclose(x.cpu().numpy(), fx[key], rtol=0, atol=1e-5)
        assert abs(float(L.calib_error(cdf)) - float(fx['mix_calib'])) < 1e-6
        c1 = L.mixture_cdf(mus[:1].contiguous(), var[:1].contiguous(), val, ym, ys)
        np.testing.assert_allclose(c1.cpu().numpy(), fx['single_cdf'], rtol=2e-5, atol=2e-7)
        x1 = L.mixture_icdf(mus[:1].contiguous(), var[:1].contiguous(), q, ym, ys, closed_form=True)
        np.testing.assert_allclose(x1.cpu().numpy(), fx['single_icdf'], rtol=3e-6)
        assert abs(float(L.calib_error(c1)) - float(fx['single_calib'])) < 1e-6


@pytest.mark.parametrize('dtype', [torch.float32, torch.float64])
@pytest.mark.parametrize('P,m', [(1, 1), (3, 50), (20, 500), (100, 333), (7, 1500), (64, 2048)])
def test_mixture_cdf_icdf_calib_match_oracle(L, dtype, P, m):
    """same three entry points vs the oracle's restatement on random mixtures (all lane-sharing widths of the quantile kernel)"""
    from oracle import pacoh_oracle as O
    g = torch.Generator().manual_seed(100 * P + m)
    mus = torch.randn(P, m, generator=g, dtype=torch.float64) * 0.7
    var = (torch.rand(P, m, generator=g, dtype=torch.float64) * 0.8 + 0.05) ** 2
    val = torch.randn(m, generator=g, dtype=torch.float64) * 1.5 + 0.3
    q = torch.rand(m, generator=g, dtype=torch.float64) * 0.98 + 0.01
    ym, ys = 0.3, 1.4
    d = lambda t: t.to(dtype).to(DEV)
    cdf = L.mixture_cdf(d(mus), d(var), d(val), ym, ys)
    cdf_o = O.mixture_cdf(mus.to(dtype).double(), var.to(dtype).double(), val.to(dtype).double(), ym, ys)
    assert float((cdf.double().cpu() - cdf_o).abs().max()) < (3e-6 if dtype == torch.float32 else 1e-13)
    x = L.mixture_icdf(d(mus), d(var), d(q), ym, ys)
    x_o = O.mixture_icdf(mus.to(dtype).double(), var.to(dtype).double(), q.to(dtype).double(), ym, ys)
    assert bool(torch.isfinite(x).all())
    assert float((x.double().cpu() - x_o).abs().max()) < (2e-5 if dtype == torch.float32 else 2.1e-6)
    # the quantiles invert the cdf
    back = L.mixture_cdf(d(mus), d(var), x, ym, ys)
    assert float((back.double().cpu() - q).abs().max()) < (2e-5 if dtype == torch.float32 else 3e-6)
    assert abs(float(L.calib_error(cdf)) - float(O.calib_error(cdf.cpu()))) < 1e-6
    if P == 1:
        xc = L.mixture_icdf(d(mus), d(var), d(q), ym, ys, closed_form=True)
        xo = O.gaussian_icdf(mus[0].to(dtype), var[0].to(dtype), q.to(dtype), ym, ys)
        assert relerr(xc, xo) < (1e-5 if dtype == torch.float32 else 1e-12)
        assert float((xc - x).abs().max()) < 1e-5


def test_mixture_icdf_stopping_rule_and_limits(L):
    """the reference's search (util.py:9-42) never reaches eps = 1e-6 once one fp32 ulp of the quantile exceeds 2e-6 (|y| >= 32):
    it then returns NaN for EVERY element after max_iter rounds -- the kernel reports the same outcome (and detects the stall at once);
    a standard-normal quantile (reference tests/test_utils.py:243-260); argument limits"""
    from oracle import pacoh_oracle as O
    mus = torch.zeros(2, 3)
    var = torch.ones(2, 3)
    q = torch.tensor([0.3, 0.5, 0.975])
    x = L.mixture_icdf(mus.to(DEV), var.to(DEV), q.to(DEV), 0.0, 1.0)
    assert abs(float(x[2]) - 1.959964) < 1e-5 and abs(float(x[1])) < 2e-6
    far = L.mixture_icdf(mus.to(DEV), var.to(DEV), q.to(DEV), 100.0, 1.0)
    far_o = O.mixture_icdf(mus, var, q, 100.0, 1.0)
    assert bool(torch.isnan(far_o).all()) and bool(torch.isnan(far).all())
    ok64 = L.mixture_icdf(mus.double().to(DEV), var.double().to(DEV), q.double().to(DEV), 100.0, 1.0)
    assert abs(float(ok64[2]) - 101.959964) < 1e-5
    few = L.mixture_icdf(mus.to(DEV), var.to(DEV), q.to(DEV), 0.0, 1.0, max_iter=10)         # cannot converge in 10 halvings of 2e8
    assert bool(torch.isnan(few).all())
    lib = L.load_library()
    big = torch.zeros(1, 2049, device=DEV)
    assert lib.pacoh_mixture_icdf(big.data_ptr(), big.data_ptr(), big.data_ptr(), big.data_ptr(), 0.0, 1.0, -1e8, 1e8, 1e-6, 10000, 0,
                                  1, 2049, 0, None) == -2
    assert lib.pacoh_mixture_icdf(big.data_ptr(), big.data_ptr(), big.data_ptr(), big.data_ptr(), 0.0, 1.0, -1e8, 1e8, 1e-6, 10000, 1,
                                  2, 8, 0, None) == -1
    assert lib.pacoh_mixture_cdf(big.data_ptr(), big.data_ptr(), big.data_ptr(), big.data_ptr(), 0.0, 0.0, 1, 1, 8, 0, None) == -1


def test_launch_stream_follows_torch_current_stream(L):
    """every C-ABI call goes to torch's CURRENT stream (side streams, graph capture): the raw accessor the binding uses must agree
    with the public API inside and outside a torch.cuda.stream() block"""
    assert (L._stream().value or 0) == torch.cuda.current_stream().cuda_stream
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        assert (L._stream().value or 0) == side.cuda_stream
        x = torch.ones(1000, device=DEV)
        y = torch.zeros(1000, device=DEV)
        L.axpy(y, x, 2.0)
    side.synchronize()
    assert float(y.sum()) == 2000.0
    assert (L._stream().value or 0) == torch.cuda.current_stream().cuda_stream


# ------------------------------------------------------------------------------------------ kernel families (round 3)
def _cos_problem(T, P, n, f, dtype, seed):
    g = torch.Generator().manual_seed(seed)
    B = T * P
    z = torch.randn(B, n, f, generator=g, dtype=torch.float64)
    mean = 0.3 * torch.randn(B, n, generator=g, dtype=torch.float64)
    y = torch.randn(T, n, generator=g, dtype=torch.float64)
    ls = torch.rand(P, f, generator=g, dtype=torch.float64) + 0.8          # period lengths (per dimension: the device allows ARD)
    # cos(pi |x - x'| / p) is positive semi-definite in one dimension only: beyond it the noise has to carry the matrix
    # (|lambda_min(os K)| <= os n: with os <= 0.05 and n <= 33 below 1.65 < noise)
    os_ = (0.2 + 0.3 * torch.rand(P, generator=g, dtype=torch.float64)) * (1.0 if f == 1 else 0.1)
    noise = (0.2 if f == 1 else 2.0) + torch.rand(P, generator=g, dtype=torch.float64)
    return [t.to(dtype) for t in (z, mean, y, ls, os_, noise)]


@pytest.mark.parametrize('dtype', [torch.float32, torch.float64])
@pytest.mark.parametrize('case', [(2, 2, 20, 1, False), (1, 3, 33, 3, False), (2, 1, 150, 1, False), (1, 2, 24, 2, True)])
def test_cosine_kernel_family_gram_lml_gradients_predict(L, dtype, case):
    """PACOH_KERNEL_COSINE (gpytorch.kernels.CosineKernel: the kernel object of the reference's tests/test_GPR.py:95-101) through
    the C ABI -- the family rides in the high bits of the `f` argument: Gram, LML, every gradient and the posterior predictive vs the
    oracle's torch restatement and its autograd, on the LDS-resident general kernel (n <= 128) and the HBM-resident path (n = 150 or
    forced)"""
    T, P, n, f, force_dense = case
    B = T * P
    z, mean, y, ls, os_, noise = _cos_problem(T, P, n, f, dtype, seed=11 * n + f)
    K = L.gram_rbf_ard(z.to(DEV), 1, z.to(DEV), 1, ls.to(DEV), os_.to(DEV), noise.to(DEV), True, B, P, kernel=L.KERNEL_COSINE)
    lsb = ls.unsqueeze(0).expand(T, P, f).reshape(B, 1, f).double()
    osb = os_.unsqueeze(0).expand(T, P).reshape(B).double()
    nb = noise.unsqueeze(0).expand(T, P).reshape(B).double()
    Kref = osb.reshape(B, 1, 1) * O.gram_cosine(z.double(), z.double(), lsb) + nb.reshape(B, 1, 1) * torch.eye(n, dtype=torch.float64)
    assert relerr(K, Kref) < (2e-6 if dtype == torch.float32 else 1e-13)
    gl = torch.rand(B, dtype=dtype) + 0.5
    leaves = [t.double().clone().requires_grad_(True) for t in (z, mean, ls, os_, noise)]
    yy = y.unsqueeze(1).expand(T, P, n).reshape(B, n).double()
    ref = O.gp_mll(leaves[0], leaves[1], yy, leaves[2].unsqueeze(0).expand(T, P, f).reshape(B, 1, f),
                   leaves[3].unsqueeze(0).expand(T, P).reshape(B), leaves[4].unsqueeze(0).expand(T, P).reshape(B), kernel='cos')
    (ref * gl.double()).sum().backward()
    old = L.FORCE_DENSE
    L.FORCE_DENSE = force_dense
    try:
        out = L.gp_lml_fwdbwd(z.to(DEV), 1, mean.to(DEV), L.MEAN_VECTOR, y.to(DEV), P, ls.to(DEV), os_.to(DEV), noise.to(DEV), B, P,
                              g_lml=gl.to(DEV), kernel=L.KERNEL_COSINE)
        lml_f, _, _, info_f = L.gp_lml_fwd(z.to(DEV), 1, mean.to(DEV), L.MEAN_VECTOR, y.to(DEV), P, ls.to(DEV), os_.to(DEV), noise.to(DEV),
                                           B, P, kernel=L.KERNEL_COSINE)
        g = torch.Generator().manual_seed(5)
        m = 9
        zt = torch.randn(B, m, f, generator=g, dtype=dtype)
        mt = 0.2 * torch.randn(B, m, generator=g, dtype=dtype)
        mu, var, cov, info_p = L.gp_predict(z.to(DEV), 1, mean.to(DEV), L.MEAN_VECTOR, y.to(DEV), P, zt.to(DEV), 1, mt.to(DEV), ls.to(DEV),
                                            os_.to(DEV), noise.to(DEV), B, P, want_cov=True, kernel=L.KERNEL_COSINE)
    finally:
        L.FORCE_DENSE = old
    lml, d_z, d_mean, d_ls, d_os, d_noise, info = out
    assert int(info.abs().max()) == 0 and int(info_f.abs().max()) == 0 and int(info_p.abs().max()) == 0
    t_l, t_g = (2e-4, 5e-3) if dtype == torch.float32 else (1e-10, 1e-8)
    assert maxrel(lml, ref) < t_l and maxrel(lml_f, ref) < t_l
    assert relerr(d_z, leaves[0].grad * 1.0) < t_g
    assert relerr(d_mean, leaves[1].grad) < t_g
    assert relerr(d_ls.reshape(T, P, f).sum(0), leaves[2].grad) < t_g
    assert relerr(d_os.reshape(T, P).sum(0), leaves[3].grad) < t_g
    assert relerr(d_noise.reshape(T, P).sum(0), leaves[4].grad) < t_g
    rm, rc = O.gp_predict(z.double(), mean.double(), yy, zt.double(), mt.double(), lsb, osb, nb, kernel='cos')
    tol = 5e-3 if dtype == torch.float32 else 1e-9
    assert relerr(mu, rm) < tol and relerr(var, torch.diagonal(rc, dim1=-2, dim2=-1)) < tol and relerr(cov, rc) < tol
    # an unknown family is refused before anything is launched
    lib = L.load_library()
    assert lib.pacoh_gram_rbf_ard(z.to(DEV).data_ptr(), 1, z.to(DEV).data_ptr(), 1, ls.to(DEV).data_ptr(), None, None, 0, K.data_ptr(), B, P,
                                  n, n, f | (2 << L.KERNEL_SHIFT), L.dtype_code(K), None) == -2
