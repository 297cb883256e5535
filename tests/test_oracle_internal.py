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


def test_svgd_vi_flavour_restated_independently_with_numpy_and_scipy():
    """The SVGD / VI flavour of the GP (SEKernelLight with unit outputscale, plain-softplus noise without a floor, the m~/(m~+T)
    pre-factor, prior_factor on the hyper-prior: models.py:418-446, random_gp.py:54-89,204-222) has no recorded reference output to
    pin it (gpytorch is absent).  Second, independent restatement: numpy loops over the reference's flattened parameter layout
    (bias before weight, row-major [out, in]) + scipy.stats.multivariate_normal, against the oracle's torch code."""
    d, hidden, f = 3, (8, 8), 2
    cfg = O.GPConfig(input_dim=d, mean_module='NN', covar_module='NN', mean_nn_layers=hidden, kernel_nn_layers=hidden)
    pm, ps = O.hyperprior_mean_std(cfg.layout, 0.5, 3.0)
    torch.manual_seed(4)
    theta = O.hyperprior_sample(cfg.layout, pm, ps, 3).double()
    rs = np.random.RandomState(2)
    tasks = [(rs.randn(n, d), rs.randn(n)) for n in (7, 11, 11, 5)]

    def net(vec, x, d_out):                              # bias BEFORE weight per layer (models.py:319-323), tanh hidden layers
        h, pos, prev = x, 0, d
        widths = list(hidden) + [d_out]
        for li, width in enumerate(widths):
            b = vec[pos:pos + width]
            W = vec[pos + width:pos + width + width * prev].reshape(width, prev)
            pos += width * (prev + 1)
            h = h @ W.T + b
            if li < len(widths) - 1:
                h = np.tanh(h)
            prev = width
        assert pos == len(vec)
        return h

    def softplus(v):
        return np.log1p(np.exp(v))

    th = theta.numpy()
    Dm = sum(O.nn_param_layout(d, 1, hidden).values())
    Dk = sum(O.nn_param_layout(d, f, hidden).values())
    assert th.shape[1] == Dm + Dk + f + 1
    total = np.zeros(3)
    for x, y in tasks:
        n = len(y)
        ref_t = O.vectorized_gp_mll(theta, torch.from_numpy(x), torch.from_numpy(y), cfg).numpy()
        for p in range(3):
            m = net(th[p, :Dm], x, 1)[:, 0]
            z = net(th[p, Dm:Dm + Dk], x, f)
            ell, s2 = softplus(th[p, Dm + Dk:Dm + Dk + f]), softplus(th[p, Dm + Dk + f])
            zs = z / ell
            K = np.exp(-0.5 * ((zs[:, None, :] - zs[None, :, :]) ** 2).sum(-1))          # unit outputscale
            val = scipy.stats.multivariate_normal(m, K + s2 * np.eye(n), allow_singular=False).logpdf(y) / n
            assert abs(val - ref_t[p]) < 1e-9 * max(1.0, abs(val))
            total[p] += val
    sizes = np.array([len(y) for _, y in tasks], dtype=np.float64)
    hm = 1.0 / np.mean(1.0 / sizes)
    pre = hm / (hm + len(tasks))                          # T = the BATCH length (random_gp.py:209)
    logprior = scipy.stats.norm(pm.numpy(), ps.numpy()).logpdf(th).sum(1)
    want = 0.01 * logprior + pre * total
    got = O.meta_log_prob(theta, [(torch.from_numpy(x), torch.from_numpy(y)) for x, y in tasks], cfg, pm, ps, 0.01, loop=True)
    # (the pre-factor is a float32 quantity in the reference and in the oracle: random_gp.py:209-212)
    assert np.abs(got.numpy() - want).max() < 1e-6 * np.abs(want).max()


def test_cosine_kernel_restatement_against_numpy():
    """gpytorch.kernels.CosineKernel restated (k = os cos(pi |x - x'| / p)): plain numpy on the same points, symmetry, unit diagonal,
    and the LML through scipy"""
    rs = np.random.RandomState(1)
    x = rs.randn(9, 1)
    p, os_, s2 = 0.8, 0.6, 0.3
    K = O.gram_cosine(torch.from_numpy(x), torch.from_numpy(x), torch.tensor([p], dtype=torch.float64), os_).numpy()
    want = os_ * np.cos(np.pi * np.abs(x - x.T) / p)
    assert np.abs(K - want).max() < 1e-12 and np.abs(np.diag(K) - os_).max() < 1e-12
    y = rs.randn(9)
    got = float(O.gp_mll(torch.from_numpy(x), torch.zeros(9, dtype=torch.float64), torch.from_numpy(y),
                         torch.tensor([[p]], dtype=torch.float64), os_, s2, kernel='cos'))
    ref = scipy.stats.multivariate_normal(np.zeros(9), want + s2 * np.eye(9)).logpdf(y) / 9
    assert abs(got - ref) < 1e-10 * max(1.0, abs(ref))
