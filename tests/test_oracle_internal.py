"""Cross-checks of the oracle's GP core against independent formulations available here
(torch.distributions.MultivariateNormal, scipy.stats.multivariate_normal, torch.autograd)."""
import numpy as np
import scipy.stats
import torch

from oracle import pacoh_oracle as O


def _problem(P=4, n=24, f=3, seed=0, dtype=torch.float64):
    g = torch.Generator().manual_seed(seed)
    z = torch.randn(P, n, f, generator=g, dtype=dtype)
    m = 0.3 * torch.randn(P, n, generator=g, dtype=dtype)
    y = torch.randn(P, n, generator=g, dtype=dtype)
    ls = torch.nn.functional.softplus(torch.randn(P, 1, f, generator=g, dtype=dtype))
    noise = torch.nn.functional.softplus(torch.randn(P, generator=g, dtype=dtype) - 1)
    os_ = torch.nn.functional.softplus(torch.randn(P, generator=g, dtype=dtype))
    return z, m, y, ls, os_, noise


def test_mll_matches_torch_and_scipy_mvn():
    z, m, y, ls, os_, noise = _problem()
    mll = O.gp_mll(z, m, y, ls, os_, noise)
    n = z.shape[1]
    for p in range(z.shape[0]):
        K = os_[p] * O.gram_rbf_ard(z[p], z[p], ls[p]) + noise[p] * torch.eye(n, dtype=z.dtype)
        ref = torch.distributions.MultivariateNormal(m[p], covariance_matrix=K).log_prob(y[p]) / n
        assert abs(float(mll[p] - ref)) < 1e-12 * max(1.0, abs(float(ref)))
        ref2 = scipy.stats.multivariate_normal(m[p].numpy(), K.numpy()).logpdf(y[p].numpy()) / n
        assert abs(float(mll[p]) - ref2) < 1e-10 * max(1.0, abs(ref2))


def test_closed_form_grads_match_autograd():
    z, m, y, ls, os_, noise = _problem(P=1, n=17, f=2, seed=3)
    z, m, y, ls, os_, noise = z[0], m[0], y[0], ls[0, 0], os_[0], noise[0]
    leaves = [t.clone().requires_grad_(True) for t in (z, m, ls, os_, noise)]
    mll = O.gp_mll(leaves[0], leaves[1], y, leaves[2], leaves[3], leaves[4])
    gz, gm, gl, go, gn = torch.autograd.grad(mll, leaves)
    cf = O.gp_mll_grads_closed_form(z, m, y, ls, os_, noise)
    assert abs(float(cf['mll'] - mll)) < 1e-13
    for a, b in [(gz, cf['d_z']), (gm, cf['d_mean']), (gl, cf['d_lengthscale']),
                 (go, cf['d_outputscale']), (gn, cf['d_noise'])]:
        assert float((a - b).abs().max()) < 1e-13


def test_predict_matches_joint_gaussian_conditioning():
    z, m, y, ls, os_, noise = _problem(P=2, n=12, f=2, seed=5)
    zt, mt = z[:, :5] + 0.1, m[:, :5] * 0.5
    mean, cov = O.gp_predict(z, m, y, zt, mt, ls, os_, noise)
    for p in range(2):
        zz = torch.cat([z[p], zt[p]])
        K = os_[p] * O.gram_rbf_ard(zz, zz, ls[p]) + noise[p] * torch.eye(17, dtype=z.dtype)
        Kxx, Kxs, Kss = K[:12, :12], K[:12, 12:], K[12:, 12:]
        mu = mt[p] + Kxs.T @ torch.linalg.solve(Kxx, y[p] - m[p])
        S = Kss - Kxs.T @ torch.linalg.solve(Kxx, Kxs)
        assert float((mean[p] - mu).abs().max()) < 1e-10
        assert float((cov[p] - S).abs().max()) < 1e-10


def test_batched_meta_log_prob_equals_task_loop():
    cfg = O.GPConfig(input_dim=2, mean_module='NN', covar_module='NN', mean_nn_layers=(8, 8),
                     kernel_nn_layers=(8, 8))
    pm, ps = O.hyperprior_mean_std(cfg.layout)
    torch.manual_seed(0)
    theta = O.hyperprior_sample(cfg.layout, pm, ps, 3).double()
    tasks = [(torch.randn(10, 2, dtype=torch.float64), torch.randn(10, dtype=torch.float64)) for _ in range(4)]
    a = O.meta_log_prob(theta, tasks, cfg, pm, ps, 0.01, loop=True)
    b = O.meta_log_prob(theta, tasks, cfg, pm, ps, 0.01, loop=False)
    assert float((a - b).abs().max()) < 1e-10


def test_jitter_retry_on_rank_deficient_gram():
    z = torch.zeros(6, 2)                      # all points identical -> K = ones, singular w/o noise
    L = O.psd_safe_cholesky(O.gram_rbf_ard(z, z, torch.ones(2)))
    assert torch.isfinite(L).all()
