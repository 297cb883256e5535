"""Left-looking dense Cholesky (csrc/dense_ll.hip) through pacoh_mvn_logprob_dense, against plain torch fp64 on the CPU:
log-density, alpha = A^-1 r, the factor left in the lower triangle, the inverse diagonal blocks left in the upper triangle
(the format trtri_dense_kernel and the backward solve read), non-PSD input.  Reference semantics:
torch.distributions.MultivariateNormal.log_prob as used at meta_learn/random_gp.py:83-85 and abstract.py:134-163."""
import math
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture()
def L():
    from meta_learning_pacoh_amd import _lib
    _lib.load_library()
    return _lib


def _spd(B, n, seed, cond=1.0):
    g = torch.Generator().manual_seed(seed)
    X = torch.randn(B, n, 6, generator=g, dtype=torch.float64)
    d2 = ((X[:, :, None, :] - X[:, None, :, :]) ** 2).sum(-1)
    return torch.exp(-0.5 * d2 / 4.0) + cond * 0.3 * torch.eye(n, dtype=torch.float64), torch.randn(B, n, generator=g, dtype=torch.float64)


SIZES64 = [98, 128, 130, 160, 192, 200, 256, 258, 320, 384, 430, 448, 500, 512]
SIZES32 = [100, 128, 132, 192, 256, 260, 384, 448, 500, 512]


@pytest.mark.parametrize('n', SIZES64)
def test_chol_ll_fp64(L, n):
    B = 3
    A, r = _spd(B, n, seed=n)
    Ad = A.cuda()
    logp, alpha, info = L.mvn_logprob_dense(Ad, r.cuda(), scale=1.0 / n, want_alpha=True)
    torch.cuda.synchronize()
    assert int(info.min()) == 0 and int(info.max()) == 0
    Lref = torch.linalg.cholesky(A)
    ref = torch.distributions.MultivariateNormal(torch.zeros(B, n, dtype=torch.float64), scale_tril=Lref).log_prob(r) / n
    assert torch.allclose(logp.cpu(), ref, rtol=1e-10, atol=1e-10), (logp.cpu(), ref)
    aref = torch.cholesky_solve(r.unsqueeze(-1), Lref).squeeze(-1)
    assert float((alpha.cpu() - aref).abs().max() / aref.abs().max()) < 1e-9
    Lgpu = torch.tril(Ad.cpu())
    assert float((Lgpu - Lref).abs().max()) < 1e-10
    # inverse of every 32 x 32 diagonal block, strictly lower part, stored transposed above the diagonal
    U = Ad.cpu()
    for k0 in range(0, n, 32):
        kb = min(32, n - k0)
        Zref = torch.linalg.inv(Lref[:, k0:k0 + kb, k0:k0 + kb])
        got = U[:, k0:k0 + kb, k0:k0 + kb].transpose(-1, -2)
        m = torch.tril(torch.ones(kb, kb, dtype=torch.bool), -1)
        assert float((got[:, m] - Zref[:, m]).abs().max()) < 1e-9, k0


@pytest.mark.parametrize('n', SIZES32)
def test_chol_ll_fp32(L, n):
    B = 2
    A, r = _spd(B, n, seed=1000 + n, cond=2.0)
    Ad = A.float().cuda()
    logp, alpha, info = L.mvn_logprob_dense(Ad, r.float().cuda(), scale=1.0, want_alpha=True)
    torch.cuda.synchronize()
    assert int(info.max()) == 0 and int(info.min()) == 0
    A32 = A.float().double()
    Lref = torch.linalg.cholesky(A32)
    ref = torch.distributions.MultivariateNormal(torch.zeros(B, n, dtype=torch.float64), scale_tril=Lref).log_prob(r.float().double())
    assert torch.allclose(logp.cpu().double(), ref, rtol=2e-4), (logp.cpu(), ref)
    aref = torch.cholesky_solve(r.float().double().unsqueeze(-1), Lref).squeeze(-1)
    assert float((alpha.cpu().double() - aref).abs().max() / aref.abs().max()) < 2e-3
    assert float((torch.tril(Ad.cpu()).double() - Lref).abs().max()) < 2e-4


def test_chol_ll_forward_only_and_u(L):
    """alpha_out = NULL (log-density only) and the values agree with the alpha-producing call"""
    n, B = 256, 4
    A, r = _spd(B, n, seed=7)
    lp1, _, info1 = L.mvn_logprob_dense(A.cuda(), r.cuda(), want_alpha=False)
    lp2, _, info2 = L.mvn_logprob_dense(A.cuda(), r.cuda(), want_alpha=True)
    torch.cuda.synchronize()
    assert torch.equal(lp1, lp2) and int(info1.max()) == 0


def test_chol_ll_not_psd(L):
    n, B = 192, 3
    A, r = _spd(B, n, seed=3)
    A[1, 150, 150] = -1.0                      # an indefinite matrix among good ones
    logp, alpha, info = L.mvn_logprob_dense(A.cuda(), r.cuda(), want_alpha=True)
    torch.cuda.synchronize()
    assert info.cpu().tolist() == [0, -1, 0]
    assert math.isnan(float(logp[1])) and not math.isnan(float(logp[0])) and not math.isnan(float(logp[2]))
    assert bool(torch.isnan(alpha[1]).all())


def test_chol_ll_is_the_kernel_that_ran(L):
    """PACOH_CHOL_LL=0 selects the right-looking kernel: both must agree to rounding (and differ in the last bits, or the switch is dead)"""
    import subprocess
    import sys
    code = ("import torch, sys; sys.path.insert(0, '.'); from meta_learning_pacoh_amd import _lib as L\n"
            "g = torch.Generator().manual_seed(5); X = torch.randn(2, 384, 6, generator=g, dtype=torch.float64)\n"
            "A = torch.exp(-0.125 * ((X[:, :, None] - X[:, None]) ** 2).sum(-1)) + 0.3 * torch.eye(384, dtype=torch.float64)\n"
            "r = torch.randn(2, 384, generator=g, dtype=torch.float64)\n"
            "lp, al, info = L.mvn_logprob_dense(A.cuda(), r.cuda(), want_alpha=True); torch.cuda.synchronize()\n"
            "print(repr(float(lp[0])), repr(float(al[1, 17])))\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for flag in ('1', '0'):
        env = dict(os.environ, PACOH_CHOL_LL=flag)
        outs.append(subprocess.run([sys.executable, '-c', code], cwd=root, env=env, capture_output=True, text=True, check=True).stdout.split())
    a, b = [float(v) for v in outs[0]], [float(v) for v in outs[1]]
    assert abs(a[0] - b[0]) <= 1e-11 * abs(b[0]) and abs(a[1] - b[1]) <= 1e-9 * max(abs(b[1]), 1e-3)
