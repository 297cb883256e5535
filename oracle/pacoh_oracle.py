"""CPU oracle for the PACOH task-GP hot path  --  TEST INFRASTRUCTURE ONLY.

This module is a plain torch/numpy (CPU) restatement of the arithmetic of the reference's
batched GP log-marginal-likelihood / posterior path.  It is
the *checker* for the HIP kernels in ``meta_learning_pacoh_amd/csrc``; it is never the thing
that is shipped or measured.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.  The product package
(``meta_learning_pacoh_amd``) must not import anything from ``oracle/``.

Pinning status ("how do we know the oracle equals the reference?"):
  * the dense GP algebra of the reference lives in the third-party dependency ``gpytorch``
    (unpinned in requirements.txt:5 / setup.py:18, era v1.0.x), which is absent from this image,
    so it is restated from its published algorithm (ExactMarginalLogLikelihood = MVN log-prob / n,
    exact GP posterior, softplus constraints);
  * PINNED against genuine reference output: the PACOH-MAP trajectory and the single-task
    GPRegressionLearned log recorded in ``demo.ipynb`` (tests/test_oracle_golden_demo.py reproduces
    both logs to the printed digits),
  * PINNED against the imported reference: ``meta_learn/svgd.py`` (SVGD phi with the RBF and the IMQ
    particle kernel, median heuristics, the gradient through the IMQ median bandwidth) and, under
    import shims, ``meta_learn/models.py`` / ``random_gp.py`` (parameter layout, hyper-prior sampling +
    log-prob, vectorised MLP forward, diagonal and full-covariance VI posterior: init stream, rsample,
    log_prob, autograd gradient; EqualWeightedMixtureDist / AffineTransformedDistribution cdf and icdf) and
    ``meta_learn/abstract.py`` (_calib_error), with ``meta_learn/util.py`` (the quantile bisection) underneath
    -- fixtures in ``tests/golden/*.npz`` produced by ``tests/golden/make_golden.py``;
  * UNPINNED by any recorded reference output (restated from the source only): the SVGD/VI GP
    flavour end-to-end values (unit outputscale, no noise floor, m~/(m~+T) pre-factor; the mixture
    predictive's moments, cdf and quantiles ARE pinned, the GP posterior feeding it is not).  See DESIGN.md "Oracle".

All ``file:line`` citations are relative to the reference repository root.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

LOG_2PI = math.log(2.0 * math.pi)


# --------------------------------------------------------------------------------------
# A1  data preparation                                   meta_learn/abstract.py:212-258
# --------------------------------------------------------------------------------------

def handle_input_dimensionality(x, y=None):
    """meta_learn/util.py:44-58 -- make x [n,d] and y [n,1]."""
    if x.ndim == 1:
        x = np.expand_dims(x, -1)
    assert x.ndim == 2
    if y is None:
        return x
    if y.ndim == 1:
        y = np.expand_dims(y, -1)
    assert x.shape[0] == y.shape[0] and y.ndim == 2
    return x, y


def compute_normalization_stats(meta_train_tuples, normalize_data=True):
    """meta_learn/abstract.py:212-221 -- pooled mean / std(+1e-8) over all tasks."""
    xs, ys = zip(*[handle_input_dimensionality(x, y) for x, y in meta_train_tuples])
    X, Y = np.concatenate(xs, axis=0), np.concatenate(ys, axis=0)
    if normalize_data:
        return (np.mean(X, axis=0), np.std(X, axis=0) + 1e-8,
                np.mean(Y, axis=0), np.std(Y, axis=0) + 1e-8)
    return np.zeros(X.shape[1]), np.ones(X.shape[1]), np.zeros(Y.shape[1]), np.ones(Y.shape[1])


def normalize(x, stats, y=None):
    """meta_learn/abstract.py:223-233."""
    x_mean, x_std, y_mean, y_std = stats
    xn = (x - x_mean[None, :]) / x_std[None, :]
    if y is None:
        return xn
    return xn, (y - y_mean[None, :]) / y_std[None, :]


def prepare_task(x, y, stats, dtype=torch.float32):
    """meta_learn/abstract.py:243-258 -- normalise, flatten y, cast (reference: float32)."""
    x, y = handle_input_dimensionality(x, y)
    xn, yn = normalize(x, stats, y)
    assert yn.shape[1] == 1
    # the reference casts float64 numpy -> float32 torch; keep that rounding step, then upcast
    xt = torch.from_numpy(xn).float().to(dtype)
    yt = torch.from_numpy(yn.flatten()).float().to(dtype)
    return xt, yt


# --------------------------------------------------------------------------------------
# A2  feature / mean networks                            meta_learn/models.py:190-227, 279-384
# --------------------------------------------------------------------------------------

def nn_param_layout(input_dim, output_dim, layer_sizes):
    """Flattened per-particle parameter layout of NeuralNetworkVectorized
    (meta_learn/models.py:319-323, 351-384): per layer BIAS first, then WEIGHT flattened
    row-major [out, in]."""
    layout = OrderedDict()
    prev = input_dim
    for i, size in enumerate(layer_sizes):
        layout['fc_%i.bias' % (i + 1)] = size
        layout['fc_%i.weight' % (i + 1)] = size * prev
        prev = size
    layout['out.bias'] = output_dim
    layout['out.weight'] = output_dim * prev
    return layout


def mlp_vectorized_forward(x, theta, input_dim, output_dim, layer_sizes):
    """NeuralNetworkVectorized.forward / LinearVectorized.forward (models.py:295-317, 343-349).

    x: [P, n, in] (or [n, in], tiled over P, models.py:305-309); theta: [P, D_net] in
    ``nn_param_layout`` order.  Returns [P, n, out]."""
    P = theta.shape[0]
    if x.ndim == 2:
        x = x.unsqueeze(0).expand(P, -1, -1)
    assert x.ndim == 3 and x.shape[0] == P
    sizes = list(layer_sizes) + [output_dim]
    h, prev, idx = x, input_dim, 0
    for li, size in enumerate(sizes):
        b = theta[:, idx:idx + size]
        idx += size
        W = theta[:, idx:idx + size * prev].reshape(P, size, prev)
        idx += size * prev
        h = torch.bmm(h, W.permute(0, 2, 1)) + b[:, None, :]
        if li < len(sizes) - 1:
            h = torch.tanh(h)
        prev = size
    assert idx == theta.shape[1]
    return h


def mlp_shared_forward(x, weights):
    """NeuralNetwork.forward (models.py:211-217) with an explicit list [(W,b), ...];
    tanh on all but the last layer."""
    h = x
    for li, (W, b) in enumerate(weights):
        h = F.linear(h, W, b)
        if li < len(weights) - 1:
            h = torch.tanh(h)
    return h


# --------------------------------------------------------------------------------------
# parameter vector layout of VectorizedGP               meta_learn/random_gp.py:24-51
# --------------------------------------------------------------------------------------

def gp_param_layout(input_dim, mean_module='NN', covar_module='NN', mean_nn_layers=(32, 32),
                    kernel_nn_layers=(32, 32), feature_dim=2):
    """Order of blocks in the flattened prior-parameter vector (random_gp.py:33-51):
    mean block(s) first (mean_nn.* or constant_mean), then kernel_nn.*, lengthscale_raw,
    noise_raw.  NB: SVGD/VI learners never forward ``feature_dim`` (GPR_meta_svgd.py:167-170),
    so it is always VectorizedGP's default 2 there."""
    layout = OrderedDict()
    if mean_module == 'NN':
        for k, v in nn_param_layout(input_dim, 1, mean_nn_layers).items():
            layout['mean_nn.' + k] = v
    elif mean_module == 'constant':
        layout['constant_mean'] = 1
    else:
        raise NotImplementedError
    if covar_module == 'NN':
        for k, v in nn_param_layout(input_dim, feature_dim, kernel_nn_layers).items():
            layout['kernel_nn.' + k] = v
        layout['lengthscale_raw'] = feature_dim
    elif covar_module == 'SE':
        layout['lengthscale_raw'] = input_dim
    else:
        raise NotImplementedError
    layout['noise_raw'] = 1
    return layout


def layout_slices(layout):
    out, idx = OrderedDict(), 0
    for k, v in layout.items():
        out[k] = (idx, idx + v)
        idx += v
    return out, idx


def hyperprior_mean_std(layout, weight_prior_std=0.5, bias_prior_std=3.0, dtype=torch.float64):
    """Gaussian hyper-prior per block (random_gp.py:126-151): constant_mean ~ N(0,1),
    lengthscale_raw ~ N(0,1), noise_raw ~ N(-1,1), NN weights N(0,weight_prior_std),
    NN biases N(0,bias_prior_std)."""
    _, D = layout_slices(layout)
    mean = torch.zeros(D, dtype=dtype)
    std = torch.ones(D, dtype=dtype)
    idx = 0
    for name, size in layout.items():
        if name == 'noise_raw':
            mean[idx:idx + size] = -1.0
        elif 'mean_nn' in name or 'kernel_nn' in name:
            std[idx:idx + size] = weight_prior_std if 'weight' in name else bias_prior_std
        idx += size
    return mean, std


def hyperprior_log_prob(theta, prior_mean, prior_std):
    """CatDist.log_prob of independent Normals (models.py:159-181) = sum of all elementwise
    Normal log-densities.  theta: [P, D] -> [P]."""
    z = (theta - prior_mean) / prior_std
    return (-0.5 * z * z - torch.log(prior_std) - 0.5 * LOG_2PI).sum(-1)


def hyperprior_sample(layout, prior_mean, prior_std, n_particles, generator=None):
    """CatDist._sample (models.py:183-184): block after block ``Normal(loc,scale).sample((P,))``
    (= torch.normal(loc.expand, scale.expand)) on the torch CPU generator, concatenated."""
    blocks, idx = [], 0
    for name, size in layout.items():
        loc = prior_mean[idx:idx + size].float().expand(n_particles, size)
        scale = prior_std[idx:idx + size].float().expand(n_particles, size)
        blocks.append(torch.normal(loc, scale, generator=generator))
        idx += size
    return torch.cat(blocks, dim=-1)


# --------------------------------------------------------------------------------------
# A4-A6  Gram build, noise, exact marginal log-likelihood
# --------------------------------------------------------------------------------------

def sq_dist_scaled(z1, z2, lengthscale):
    """sum_k ((z1_ik - z2_jk)/l_k)^2 by direct differences.  z1 [...,n,f], z2 [...,m,f],
    lengthscale [...,1,f] or [f].  (gpytorch's Kernel.covar_dist uses the
    |a|^2+|b|^2-2ab form, clamped at 0 -- equal in exact arithmetic)"""
    a = (z1 / lengthscale).unsqueeze(-2)
    b = (z2 / lengthscale).unsqueeze(-3)
    return ((a - b) ** 2).sum(-1)


def gram_rbf_ard(z1, z2, lengthscale, outputscale=1.0):
    """A4: SEKernelLight.forward (models.py:428-446) / ScaleKernel(RBFKernel(ard))
    (GPR_meta_mll.py:218,223): K_ij = os * exp(-0.5 * sum_k ((z_ik - z_jk)/l_k)^2)."""
    return outputscale * torch.exp(-0.5 * sq_dist_scaled(z1, z2, lengthscale))


def gram_cosine(z1, z2, period, outputscale=1.0):
    """gpytorch.kernels.CosineKernel [gpytorch-upstream; absent here, restated from its documented definition]:
    K_ij = os * cos(pi * |z_i - z_j| / period_length) -- the kernel object the reference's own suite hands to the single-task
    learner (tests/test_GPR.py:95-101; GPR_mll.py:41 accepts any gpytorch.kernels.Kernel).  `period` broadcasts like a
    lengthscale ([...,1,1] or [...,1,f] with equal entries)."""
    d2 = sq_dist_scaled(z1, z2, period)
    return outputscale * torch.cos(math.pi * torch.sqrt(d2.clamp_min(1e-30)))


def gram_family(z1, z2, lengthscale, outputscale=1.0, kernel='rbf'):
    return gram_cosine(z1, z2, lengthscale, outputscale) if kernel == 'cos' else gram_rbf_ard(z1, z2, lengthscale, outputscale)


def _as_batched(v, like):
    v = torch.as_tensor(v, dtype=like.dtype)
    return v


def psd_safe_cholesky(A):
    """gpytorch.utils.cholesky.psd_safe_cholesky [gpytorch-upstream]: plain Cholesky, on
    failure retry with diagonal jitter 1e-6 (f32) / 1e-8 (f64), x10 up to 3 times."""
    L, info = torch.linalg.cholesky_ex(A)
    if not bool((info > 0).any()):
        return L
    jitter = 1e-6 if A.dtype == torch.float32 else 1e-8
    eye = torch.eye(A.shape[-1], dtype=A.dtype)
    for i in range(3):
        L, info = torch.linalg.cholesky_ex(A + (jitter * 10 ** i) * eye)
        if not bool((info > 0).any()):
            return L
    raise RuntimeError('matrix not positive definite after jitter')


def gp_mll(z, mean, y, lengthscale, outputscale, noise, kernel='rbf'):
    """A5+A6: per-datapoint exact log marginal likelihood
    (gpytorch.mlls.ExactMarginalLogLikelihood, call sites GPR_meta_mll.py:72,111-113 and
    random_gp.py:83-85):  MVN(mean, K + noise*I).log_prob(y) / n.

    z [...,n,f]; mean,y [...,n]; lengthscale [...,1,f]; outputscale,noise [...] or scalars.
    Returns [...]."""
    n = z.shape[-2]
    K = gram_family(z, z, lengthscale, 1.0, kernel)
    os_ = torch.as_tensor(outputscale, dtype=z.dtype)
    nz = torch.as_tensor(noise, dtype=z.dtype)
    if os_.ndim > 0:
        os_ = os_.reshape(os_.shape + (1, 1))
    if nz.ndim > 0:
        nz = nz.reshape(nz.shape + (1, 1))
    Ky = os_ * K + nz * torch.eye(n, dtype=z.dtype)
    L = psd_safe_cholesky(Ky)
    r = (y - mean).unsqueeze(-1)
    alpha = torch.cholesky_solve(r, L)
    quad = (r * alpha).sum((-2, -1))
    logdet = 2.0 * torch.log(torch.diagonal(L, dim1=-2, dim2=-1)).sum(-1)
    return -0.5 * (quad + logdet + n * LOG_2PI) / n


def gp_mll_grads_closed_form(z, mean, y, lengthscale, outputscale, noise):
    """Closed-form gradients of ``gp_mll``, used to check the HIP
    backward kernel independently of autograd:
      G = 0.5*(alpha alpha^T - Ky^-1)/n,  K = noise-free Gram (incl. outputscale)
      d/dl_k   = sum_ij G_ij K_ij (z_ik-z_jk)^2 / l_k^3
      d/dnoise = tr G
      d/dos    = sum_ij G_ij K_ij / os
      d/dz_ik  = 2 sum_j G_ij K_ij (z_jk - z_ik)/l_k^2
      d/dmean  = alpha/n
    Single (unbatched) problem: z [n,f], lengthscale [f]."""
    n, f = z.shape
    K = outputscale * torch.exp(-0.5 * sq_dist_scaled(z, z, lengthscale))
    Ky = K + noise * torch.eye(n, dtype=z.dtype)
    L = torch.linalg.cholesky(Ky)
    r = (y - mean).unsqueeze(-1)
    alpha = torch.cholesky_solve(r, L)
    Kinv = torch.cholesky_inverse(L)
    G = 0.5 * (alpha @ alpha.T - Kinv) / n
    M = G * K
    diff = z.unsqueeze(1) - z.unsqueeze(0)                    # [i,j,k] = z_ik - z_jk
    d_ls = (M.unsqueeze(-1) * diff ** 2).sum((0, 1)) / lengthscale ** 3
    d_noise = torch.trace(G)
    d_os = M.sum() / outputscale
    d_z = 2.0 * (M.unsqueeze(-1) * (-diff)).sum(1) / lengthscale ** 2
    d_mean = alpha.squeeze(-1) / n
    mll = -0.5 * ((r * alpha).sum() + 2 * torch.log(torch.diagonal(L)).sum() + n * LOG_2PI) / n
    return dict(mll=mll, d_z=d_z, d_mean=d_mean, d_lengthscale=d_ls, d_outputscale=d_os,
                d_noise=d_noise)


# --------------------------------------------------------------------------------------
# A3 + A2 + A6  VectorizedGP.forward (train branch)      meta_learn/random_gp.py:54-89
# --------------------------------------------------------------------------------------

class GPConfig:
    """Static description of the vectorised GP prior (random_gp.py:24-51)."""

    def __init__(self, input_dim, mean_module='NN', covar_module='NN', mean_nn_layers=(32, 32),
                 kernel_nn_layers=(32, 32), feature_dim=2):
        self.input_dim, self.mean_module, self.covar_module = input_dim, mean_module, covar_module
        self.mean_nn_layers, self.kernel_nn_layers = tuple(mean_nn_layers), tuple(kernel_nn_layers)
        self.feature_dim = feature_dim
        self.layout = gp_param_layout(input_dim, mean_module, covar_module, mean_nn_layers,
                                      kernel_nn_layers, feature_dim)
        self.slices, self.D = layout_slices(self.layout)

    def block(self, theta, prefix):
        keys = [k for k in self.layout if k.startswith(prefix)]
        lo, hi = self.slices[keys[0]][0], self.slices[keys[-1]][1]
        return theta[..., lo:hi]


def consume_vectorized_gp_init_rng(cfg):
    """Advance the torch CPU generator exactly as constructing the reference's VectorizedGP does:
    every LinearVectorized draws torch.normal(0,1,(in*out,)), then weight.uniform_ and
    bias.uniform_ (models.py:283-293); mean_nn is built before kernel_nn (random_gp.py:33-46).
    The values are discarded by the reference too (set_parameters_as_vector overwrites them,
    SURVEY appendix) -- only the stream position matters for seed-for-seed reproducibility."""
    def net(in_dim, out_dim, layers):
        prev = in_dim
        for size in list(layers) + [out_dim]:
            w = torch.normal(0, 1, size=(prev * size,))
            w.uniform_(-1.0, 1.0)
            torch.empty(size).uniform_(-1.0, 1.0)
            prev = size
    if cfg.mean_module == 'NN':
        net(cfg.input_dim, 1, cfg.mean_nn_layers)
    if cfg.covar_module == 'NN':
        net(cfg.input_dim, cfg.feature_dim, cfg.kernel_nn_layers)


def vectorized_gp_features(theta, x, cfg):
    """Mean values m [P,n], kernel inputs z [P,n,f], lengthscale [P,1,f], noise [P] from the
    flattened particles (random_gp.py:56-74; models.py:505-519).  SVGD/VI flavour: plain
    softplus for lengthscale and noise (NO floor), outputscale == 1 (models.py:420)."""
    P = theta.shape[0]
    xP = x.unsqueeze(0).expand(P, -1, -1) if x.ndim == 2 else x
    if cfg.mean_module == 'NN':
        m = mlp_vectorized_forward(xP, cfg.block(theta, 'mean_nn.'), cfg.input_dim, 1,
                                   cfg.mean_nn_layers).squeeze(-1)
    else:
        m = cfg.block(theta, 'constant_mean').expand(P, xP.shape[1])
    if cfg.covar_module == 'NN':
        z = mlp_vectorized_forward(xP, cfg.block(theta, 'kernel_nn.'), cfg.input_dim,
                                   cfg.feature_dim, cfg.kernel_nn_layers)
    else:
        z = xP
    ls = F.softplus(cfg.block(theta, 'lengthscale_raw')).unsqueeze(1)
    noise = F.softplus(cfg.block(theta, 'noise_raw')).squeeze(-1)
    return m, z, ls, noise


def vectorized_gp_mll(theta, x, y, cfg):
    """``_, mll = VectorizedGP.forward(x, y)`` (random_gp.py:54-85): per-particle,
    per-datapoint MLL of ONE task.  theta [P,D], x [n,d], y [n] -> [P]."""
    m, z, ls, noise = vectorized_gp_features(theta, x, cfg)
    return gp_mll(z, m, y.unsqueeze(0).expand(theta.shape[0], -1), ls, 1.0, noise)


# --------------------------------------------------------------------------------------
# A7  meta objective                                     meta_learn/random_gp.py:204-222
# --------------------------------------------------------------------------------------

def meta_pre_factor(task_sizes):
    """m~/(m~+T) with m~ the harmonic mean of the batch's dataset sizes and T the BATCH
    length (random_gp.py:209-212)."""
    sizes = torch.tensor([float(s) for s in task_sizes])
    hm = 1.0 / torch.mean(1.0 / sizes)
    return float(hm / (hm + len(task_sizes)))


def meta_log_prob(theta, tasks, cfg, prior_mean, prior_std, prior_factor, loop=True):
    """RandomGPMeta.log_prob (random_gp.py:221-222): prior_factor*log p(theta) +
    pre_factor * sum_t mll_t(theta).  tasks: list of (x[n,d], y[n]).  ``loop=True`` is the
    reference's serial loop over tasks (random_gp.py:215-217); ``loop=False`` batches all
    equal-size tasks in one call (same arithmetic, used only as the strong CPU baseline)."""
    pre = meta_pre_factor([x.shape[-2] for x, _ in tasks])
    if loop:
        mll_sum = torch.stack([vectorized_gp_mll(theta, x, y, cfg) for x, y in tasks], -1).sum(-1)
    else:
        P, T = theta.shape[0], len(tasks)
        X = torch.stack([x for x, _ in tasks])                        # [T,n,d]
        Y = torch.stack([y for _, y in tasks])                        # [T,n]
        n, d = X.shape[1:]
        th = theta.unsqueeze(0).expand(T, -1, -1).reshape(T * P, -1)
        xs = X.unsqueeze(1).expand(-1, P, -1, -1).reshape(T * P, n, d)
        ys = Y.unsqueeze(1).expand(-1, P, -1).reshape(T * P, n)
        m, z, ls, noise = vectorized_gp_features(th, xs, cfg)
        mll_sum = gp_mll(z, m, ys, ls, 1.0, noise).reshape(T, P).sum(0)
    return prior_factor * hyperprior_log_prob(theta, prior_mean.to(theta.dtype),
                                              prior_std.to(theta.dtype)) + pre * mll_sum


def meta_score(theta, tasks, cfg, prior_mean, prior_std, prior_factor, loop=True):
    """score = grad_theta sum_p log_prob_p   (svgd.py:15-16)."""
    th = theta.detach().clone().requires_grad_(True)
    lp = meta_log_prob(th, tasks, cfg, prior_mean, prior_std, prior_factor, loop=loop)
    (score,) = torch.autograd.grad(lp.sum(), th)
    return lp.detach(), score


# --------------------------------------------------------------------------------------
# A9  SVGD                                               meta_learn/svgd.py:12-59,103-107
# --------------------------------------------------------------------------------------

def svgd_norm_sq(X, Y):
    """svgd.py:103-107 (matmul form, kept because the median heuristic sees its rounding)."""
    XX, XY, YY = X.matmul(X.t()), X.matmul(Y.t()), Y.matmul(Y.t())
    return -2 * XY + XX.diag().unsqueeze(1) + YY.diag().unsqueeze(0)


def svgd_bandwidth(dnorm2, bandwidth=None):
    """RBF_Kernel._bandwidth (svgd.py:44-51): sqrt(median(d^2 incl. the zero diagonal) /
    (2 ln(P+1))) via numpy median, unless a fixed bandwidth is given."""
    if bandwidth is not None:
        return bandwidth
    d = dnorm2.detach().cpu().numpy()
    h = np.median(d) / (2 * np.log(d.shape[0] + 1))
    return np.sqrt(h).item()


def svgd_phi_closed_form(X, score, bandwidth=None):
    """SVGD.phi with RBF_Kernel (svgd.py:12-23, 53-59) in closed form:
       gamma = 1/(1e-8 + 2 bw^2); k_ij = exp(-gamma |x_i-x_j|^2)
       phi_i = (sum_j k_ij s_j + 2 gamma sum_j k_ij (x_i - x_j)) / P."""
    dn = svgd_norm_sq(X, X)
    bw = svgd_bandwidth(dn, bandwidth)
    gamma = 1.0 / (1e-8 + 2 * bw ** 2)
    K = torch.exp(-gamma * dn)
    grad_K = 2 * gamma * (K.sum(1, keepdim=True) * X - K @ X)
    return (K @ score + grad_K) / X.shape[0], bw


def svgd_imq_bandwidth(X):
    """IMQSteinKernel._bandwidth (svgd.py:78-89): per-dimension LOWER median (torch.median) of the squared
    coordinate differences over the pairs i < j, divided by log(P+1).  Also returns the pair (a, b), a < b,
    that attains the median in each dimension -- the reference's autograd differentiates through it."""
    P = X.shape[0]
    iu, ju = torch.triu_indices(P, P, offset=1)                      # row-major pairs i < j
    sq = (X[ju] - X[iu]) ** 2                                        # [npairs, D]
    k = (sq.shape[0] - 1) // 2
    srt, order = torch.sort(sq, dim=0, stable=True)
    med, arg = srt[k], order[k]
    return med / math.log(P + 1), iu[arg], ju[arg]


def svgd_phi_imq_closed_form(X, score, alpha=0.5, beta=-0.5, bandwidth=None):
    """SVGD.phi with IMQSteinKernel (svgd.py:12-23, 63-97) in closed form.
       base_ij = alpha + sum_d (x_jd - x_id)^2 / h_d,  k_ij = base_ij^beta,  kb_ij = beta base_ij^(beta-1)
       grad_K[j,d] = -sum_i kb_ij 2 (x_jd - x_id) / h_d                              (direct term)
                     + [j == b_d] (sum_il kb_il (x_ld - x_id)^2 / h_d^2) * 2 (x_bd - x_ad) / log(P+1)
       The second term is the derivative through the median bandwidth (the reference builds h from the
       differentiable `norm_sq`, whose X-side is the later particle b of the median pair (a, b), svgd.py:85-87,92);
       it is absent for a fixed bandwidth.  phi = (K score + grad_K) / P."""
    P = X.shape[0]
    diff = X.unsqueeze(0) - X.unsqueeze(1)                           # [i, j, d] = x_j - x_i
    nsq = diff ** 2
    if bandwidth is None:
        h, a_idx, b_idx = svgd_imq_bandwidth(X)
    else:
        h = torch.as_tensor(bandwidth, dtype=X.dtype)
    base = alpha + (nsq / h).sum(-1)
    K = torch.exp(beta * torch.log(base))
    Kb = beta * K / base
    grad_K = -(Kb.unsqueeze(-1) * 2 * diff / h).sum(0)               # sum over i -> [j, d]
    if bandwidth is None:
        S = -(Kb.unsqueeze(-1) * nsq).sum((0, 1)) / h ** 2           # d sum(K) / d h_d
        d_idx = torch.arange(X.shape[1])
        dh = 2 * (X[b_idx, d_idx] - X[a_idx, d_idx]) / math.log(P + 1)
        grad_K[b_idx, d_idx] -= S * dh
    return (K @ score + grad_K) / P, h


# --------------------------------------------------------------------------------------
# A10  VI                                               GPR_meta_vi.py:216-224, random_gp.py:224-251
# --------------------------------------------------------------------------------------

def vi_neg_elbo(loc, log_scale, eps, tasks, cfg, prior_mean, prior_std, prior_factor, loop=True):
    """get_neg_elbo with the diagonal Gaussian posterior: theta_s = loc + exp(log_scale)*eps_s;
    elbo_s = log p(theta_s) - prior_factor * log q(theta_s); loss = -mean_s elbo_s."""
    scale = torch.exp(log_scale)
    theta = loc + scale * eps
    log_q = (-0.5 * eps ** 2 - log_scale - 0.5 * LOG_2PI).sum(-1)
    lp = meta_log_prob(theta, tasks, cfg, prior_mean, prior_std, prior_factor, loop=loop)
    return -(lp - prior_factor * log_q).mean()


def vi_full_init(D):
    """RandomGPPosterior.__init__ for cov_type='full' (random_gp.py:244,249-250): loc ~ N(0, 0.1), then
    tril_cov = diag(U(0.05, 0.1)), drawn in this order from the torch CPU generator."""
    loc = torch.normal(0.0, 0.1, size=(D,))
    tril = torch.diag(torch.ones(D).uniform_(0.05, 0.1))
    return loc, tril


def vi_full_sample(loc, tril_cov, eps):
    """MultivariateNormal(loc, scale_tril=tril(tril_cov)).rsample / .log_prob (random_gp.py:251) with the
    standard-normal draw eps made explicit: theta_s = loc + L eps_s,
    log q(theta_s) = -0.5 |L^-1 (theta_s - loc)|^2 - sum_d log L_dd - D/2 log(2 pi)."""
    Lt = torch.tril(tril_cov)
    theta = loc + eps @ Lt.t()
    D = loc.shape[0]
    log_q = -0.5 * (eps ** 2).sum(-1) - torch.log(torch.diagonal(Lt)).sum() - 0.5 * D * LOG_2PI
    return theta, log_q


def vi_full_grad(tril_cov, eps, score, prior_factor):
    """gradient of  -mean_s [log p(theta_s) - prior_factor log q(theta_s)]  w.r.t. (loc, tril_cov) given the
    per-sample score d log p / d theta_s (closed form of the autograd backward of GPR_meta_vi.py:220-224):
       d/d loc = -mean_s score_s;   d/d L_ij (i >= j) = -mean_s score_si eps_sj - [i == j] prior_factor / L_ii;
    entries above the diagonal get exactly 0 (torch.tril mask)."""
    S = eps.shape[0]
    Lt = torch.tril(tril_cov)
    g_tril = torch.tril(-(score.t() @ eps) / S)
    g_tril = g_tril - torch.diag(prior_factor / torch.diagonal(Lt))
    return -score.mean(0), g_tril


# --------------------------------------------------------------------------------------
# A11  posterior predictive + eval metrics
# --------------------------------------------------------------------------------------

def gp_predict(z_ctx, m_ctx, y_ctx, z_tst, m_tst, lengthscale, outputscale, noise, kernel='rbf'):
    """Exact GP posterior predictive incl. observation noise (eval-mode ExactGP +
    likelihood [gpytorch-upstream]; call sites GPR_meta_mll.py:174-181, GPR_meta_svgd.py:203-212):
      mu* = m* + K*x (Kxx + s2 I)^-1 (y - mx);  S* = K** - K*x (Kxx + s2 I)^-1 Kx* + s2 I.
    Batched over leading dims.  Returns (mean [...,m], cov [...,m,m])."""
    n, m = z_ctx.shape[-2], z_tst.shape[-2]
    os_ = torch.as_tensor(outputscale, dtype=z_ctx.dtype)
    nz = torch.as_tensor(noise, dtype=z_ctx.dtype)
    if os_.ndim > 0:
        os_ = os_.reshape(os_.shape + (1, 1))
    if nz.ndim > 0:
        nz = nz.reshape(nz.shape + (1, 1))
    Kxx = os_ * gram_family(z_ctx, z_ctx, lengthscale, 1.0, kernel) + nz * torch.eye(n, dtype=z_ctx.dtype)
    Kxs = os_ * gram_family(z_ctx, z_tst, lengthscale, 1.0, kernel)
    Kss = os_ * gram_family(z_tst, z_tst, lengthscale, 1.0, kernel)
    L = psd_safe_cholesky(Kxx)
    alpha = torch.cholesky_solve((y_ctx - m_ctx).unsqueeze(-1), L)
    mean = m_tst + (Kxs.transpose(-1, -2) @ alpha).squeeze(-1)
    V = torch.linalg.solve_triangular(L, Kxs, upper=False)
    cov = Kss - V.transpose(-1, -2) @ V + nz * torch.eye(m, dtype=z_ctx.dtype)
    return mean, cov


def mvn_log_prob(value, mean, cov):
    L = psd_safe_cholesky(cov)
    r = (value - mean).unsqueeze(-1)
    a = torch.linalg.solve_triangular(L, r, upper=False)
    return -0.5 * ((a * a).sum((-2, -1)) + 2 * torch.log(torch.diagonal(L, dim1=-2, dim2=-1)).sum(-1)
                   + value.shape[-1] * LOG_2PI)


def calib_error(cdf_vals):
    """meta_learn/abstract.py:260-272: RMSE between empirical frequencies of
    cdf(y) <= level and the 20 levels linspace(0.05,0.95)."""
    conf = torch.linspace(0.05, 0.95, 20)
    emp = torch.sum(cdf_vals.flatten()[:, None].float() <= conf, dim=0).float() / cdf_vals.numel()
    return torch.sqrt(torch.mean((emp - conf) ** 2))


def eval_metrics(mean_n, cov_n, test_y, y_mean, y_std):
    """RegressionModelMetaLearned.eval (abstract.py:134-163) on a (mixture of) Gaussian
    predictive(s) given in NORMALISED space: mean_n [P,m] (or [m]), cov_n [P,m,m] (or [m,m]).
    Un-normalisation = AffineTransformedDistribution (models.py:15-43).
    Returns (avg joint log-lik per test point, rmse, calibration error)."""
    if mean_n.ndim == 1:
        mean_n, cov_n, single = mean_n[None], cov_n[None], True
    else:
        single = False
    P, m = mean_n.shape
    y_mean_t, y_std_t = float(np.asarray(y_mean).reshape(-1)[0]), float(np.asarray(y_std).reshape(-1)[0])
    ty = torch.as_tensor(test_y, dtype=mean_n.dtype).flatten()
    ty_n = (ty - y_mean_t) / y_std_t
    lp = mvn_log_prob(ty_n.unsqueeze(0).expand(P, -1), mean_n, cov_n) - m * math.log(y_std_t)
    ll = lp[0] if single else torch.logsumexp(lp, 0) - math.log(P)     # models.py:117-122
    avg_ll = ll / m
    mean_p = mean_n * y_std_t + y_mean_t
    std_p = torch.sqrt(torch.diagonal(cov_n, dim1=-2, dim2=-1)) * y_std_t
    rmse = torch.sqrt(torch.mean((mean_p.mean(0) - ty) ** 2))
    cdf = torch.distributions.Normal(mean_p, std_p).cdf(ty.unsqueeze(0)).mean(0)  # models.py:124-131
    return float(avg_ll), float(rmse), float(calib_error(cdf))


def mixture_cdf(mean_n, var_n, value, y_mean, y_std):
    """EqualWeightedMixtureDist.cdf (models.py:124-131) of the un-normalised components (AffineTransformedDistribution,
    models.py:15-43): mean over the P components of Phi((value - (y_mean + y_std mu)) / (y_std sigma)).  mean_n, var_n [P,m]."""
    y_mean_t, y_std_t = float(np.asarray(y_mean).reshape(-1)[0]), float(np.asarray(y_std).reshape(-1)[0])
    value = torch.as_tensor(value, dtype=mean_n.dtype).flatten()
    z = (value.unsqueeze(0) - (mean_n * y_std_t + y_mean_t)) / (torch.sqrt(var_n) * y_std_t)
    return (0.5 * (1 + torch.erf(z / math.sqrt(2.0)))).mean(0)


def mixture_icdf(mean_n, var_n, quantile, y_mean, y_std, lo=-1e8, hi=1e8, eps=1e-6, max_iter=10000):
    """EqualWeightedMixtureDist.icdf (models.py:136-140): the interval-halving root search of util.py:9-42 on
    cdf(x) - quantile, every element starting from [lo, hi], stopping when the LARGEST half-width is <= eps and
    returning the last midpoint (NaN for every element past max_iter).  One Gaussian (P == 1): see gaussian_icdf."""
    q = torch.as_tensor(quantile, dtype=mean_n.dtype).flatten()
    left, right = torch.full_like(q, lo), torch.full_like(q, hi)
    for _ in range(int(max_iter)):          # the reference gives up (NaN) when round max_iter + 1 would be needed
        mid = (left + right) / 2
        below = mixture_cdf(mean_n, var_n, mid, y_mean, y_std) - q < 0
        left, right = torch.where(below, mid, left), torch.where(below, right, mid)
        if not float((right - left).abs().max()) / 2 > eps:
            return mid
    return torch.full_like(q, float('nan'))


def gaussian_icdf(mean_n, var_n, quantile, y_mean, y_std):
    """AffineTransformedDistribution(Normal).icdf (models.py:15-43): y_mean + y_std (mu + sigma sqrt(2) erfinv(2q - 1))"""
    y_mean_t, y_std_t = float(np.asarray(y_mean).reshape(-1)[0]), float(np.asarray(y_std).reshape(-1)[0])
    q = torch.as_tensor(quantile, dtype=mean_n.dtype).flatten()
    return y_mean_t + y_std_t * (mean_n + torch.sqrt(var_n) * math.sqrt(2.0) * torch.erfinv(2 * q - 1))


def mixture_mean_std(mean_n, cov_n, y_mean, y_std):
    """EqualWeightedMixtureDist.mean / .stddev (models.py:90-115) after un-normalisation."""
    y_mean_t, y_std_t = float(np.asarray(y_mean).reshape(-1)[0]), float(np.asarray(y_std).reshape(-1)[0])
    mean_p = mean_n * y_std_t + y_mean_t
    var_p = torch.diagonal(cov_n, dim1=-2, dim2=-1) * y_std_t ** 2
    mu = mean_p.mean(0)
    var = ((mean_p - mu) ** 2).mean(0) + var_p.mean(0)
    return mu, torch.sqrt(var)


# --------------------------------------------------------------------------------------
# A8 + A12  PACOH-MAP restated end-to-end               meta_learn/GPR_meta_mll.py
# --------------------------------------------------------------------------------------

class MapOracle:
    """Plain-torch restatement of GPRegressionMetaLearned (GPR_meta_mll.py:12-264) for the
    default configuration (NN mean + NN kernel or SE / constant / zero).  Its ``meta_fit`` log
    is checked digit by digit against demo.ipynb:115-127 (tests/test_oracle_golden_demo.py)."""

    def __init__(self, meta_train_data, lr_params=1e-3, weight_decay=0.0, feature_dim=2,
                 num_iter_fit=10000, covar_module='NN', mean_module='NN', mean_nn_layers=(32, 32),
                 kernel_nn_layers=(32, 32), task_batch_size=5, normalize_data=True,
                 lr_decay=1.0, random_seed=None, dtype=torch.float32):
        self.dtype = dtype
        if random_seed is not None:                                   # abstract.py:125-129
            torch.manual_seed(random_seed)
            self.rds = np.random.RandomState(random_seed + 1)
        else:
            self.rds = np.random
        data = [handle_input_dimensionality(x, y) for x, y in meta_train_data]
        self.input_dim = data[0][0].shape[-1]
        self.stats = compute_normalization_stats(data, normalize_data)
        self.num_iter_fit, self.task_batch_size = num_iter_fit, task_batch_size
        self.covar_module, self.mean_module = covar_module, mean_module

        def make_net(out_dim, layers):                                # models.py:190-209
            mods, prev = [], self.input_dim
            for s in layers:
                mods.append(torch.nn.Linear(prev, s)); prev = s
            mods.append(torch.nn.Linear(prev, out_dim))
            return [m.to(dtype) for m in mods]

        params = []
        # kernel net is built BEFORE the mean net (GPR_meta_mll.py:214-231) -> RNG order
        if covar_module == 'NN':
            self.kernel_net = make_net(feature_dim, kernel_nn_layers)
            for m in self.kernel_net: params += [m.weight, m.bias]
            ls_dim = feature_dim
        else:
            self.kernel_net, ls_dim = None, (1 if covar_module == 'COS' else self.input_dim)     # COS: one period_length
        self.kernel = 'cos' if covar_module == 'COS' else 'rbf'
        if mean_module == 'NN':
            self.mean_net = make_net(1, mean_nn_layers)
            for m in self.mean_net: params += [m.weight, m.bias]
        else:
            self.mean_net = None
        # raw GP hyper-parameters all start at 0 (gpytorch defaults) [gpytorch-upstream]
        self.raw_lengthscale = torch.zeros(1, ls_dim, dtype=dtype, requires_grad=True)
        self.raw_outputscale = torch.zeros((), dtype=dtype, requires_grad=True)
        self.raw_noise = torch.zeros(1, dtype=dtype, requires_grad=True)
        params += [self.raw_lengthscale, self.raw_outputscale]
        if mean_module == 'constant':
            self.constant_mean = torch.zeros(1, dtype=dtype, requires_grad=True)
            params.append(self.constant_mean)
        params.append(self.raw_noise)
        self.params = params
        # AdamW with weight decay on EVERY group (GPR_meta_mll.py:255)
        self.optimizer = torch.optim.AdamW(params, lr=lr_params, weight_decay=weight_decay)
        self.scheduler = (torch.optim.lr_scheduler.StepLR(self.optimizer, 1000, gamma=lr_decay)
                          if lr_decay < 1.0 else None)
        self.tasks = [prepare_task(x, y, self.stats, dtype) for x, y in data]

    # ---- model pieces -------------------------------------------------------------
    def hypers(self):
        ls = F.softplus(self.raw_lengthscale)
        os_ = F.softplus(self.raw_outputscale)
        noise = F.softplus(self.raw_noise) + 1e-3                    # GreaterThan(1e-3), :54-55
        return ls, os_, noise.squeeze(0)

    def features(self, x):
        z = mlp_shared_forward(x, [(m.weight, m.bias) for m in self.kernel_net]) \
            if self.kernel_net is not None else x
        if self.mean_net is not None:
            m = mlp_shared_forward(x, [(l.weight, l.bias) for l in self.mean_net]).squeeze(-1)
        elif self.mean_module == 'constant':
            m = self.constant_mean.expand(x.shape[0])
        else:
            m = torch.zeros(x.shape[0], dtype=x.dtype)
        return z, m

    def task_mll(self, x, y):
        z, m = self.features(x)
        ls, os_, noise = self.hypers()
        return gp_mll(z, m, y, ls, os_, noise, kernel=self.kernel)

    # ---- training loop ------------------------------------------------------------
    def meta_fit(self, valid_tuples=None, log_period=500, n_iter=None, log_fn=None):
        """GPR_meta_mll.py:82-147; returns the list of (itr, avg_loss, ll, rmse, calib)."""
        n_iter = self.num_iter_fit if n_iter is None else n_iter
        cum_loss, log = 0.0, []
        for itr in range(1, n_iter + 1):
            self.optimizer.zero_grad()
            idx = self.rds.randint(0, len(self.tasks), self.task_batch_size)   # == choice(), A12
            loss = 0.0
            for i in idx:
                loss = loss - self.task_mll(*self.tasks[i])
            loss.backward()
            self.optimizer.step()
            if self.scheduler is not None:
                self.scheduler.step()
            cum_loss += float(loss.detach())
            if itr == 1 or itr % log_period == 0:
                avg = cum_loss / (log_period if itr > 1 else 1.0)
                cum_loss = 0.0
                rec = (itr, avg) + (self.eval_datasets(valid_tuples) if valid_tuples is not None else ())
                log.append(rec)
                if log_fn is not None:
                    log_fn(rec)
        return log

    # ---- prediction / evaluation --------------------------------------------------
    def predict_normalized(self, cx, cy, tx):
        cx, cy = prepare_task(cx, cy, self.stats, self.dtype)
        tx = torch.from_numpy(normalize(handle_input_dimensionality(tx), self.stats)).float().to(self.dtype)
        with torch.no_grad():
            zc, mc = self.features(cx)
            zt, mt = self.features(tx)
            ls, os_, noise = self.hypers()
            return gp_predict(zc, mc, cy, zt, mt, ls, os_, noise, kernel=self.kernel)

    def eval(self, cx, cy, tx, ty):
        mean, cov = self.predict_normalized(cx, cy, tx)
        return eval_metrics(mean, cov, np.asarray(ty), self.stats[2], self.stats[3])

    def eval_datasets(self, tuples):
        res = np.array([self.eval(*t) for t in tuples])
        return tuple(res.mean(0))


class SingleTaskOracle(MapOracle):
    """Plain-torch restatement of the single-task learner GPRegressionLearned (meta_learn/GPR_mll.py:11-216,
    RegressionModel abstract.py:7-115).  It differs from PACOH-MAP in: normalisation statistics of the one
    training set, the default GaussianLikelihood noise bound 1e-4 [gpytorch-upstream] (GPR_mll.py:88), AdamW's
    DEFAULT weight decay 1e-2 on every non-NN group (GPR_mll.py:57,69,82,92), only the groups selected by
    `learning_mode` being trained, and ReduceLROnPlateau stepped at the log lines (GPR_mll.py:104-107,148).
    Its fit() log is checked against the recorded demo.ipynb output (tests/test_oracle_golden_demo.py)."""

    def __init__(self, train_x, train_t, learning_mode='both', lr=1e-3, weight_decay=0.0, feature_dim=2,
                 num_iter_fit=1000, covar_module='NN', mean_module='NN', mean_nn_layers=(32, 32),
                 kernel_nn_layers=(32, 32), normalize_data=True, lr_scheduler=True, random_seed=None,
                 dtype=torch.float32):
        super().__init__([(train_x, train_t)], lr_params=lr, weight_decay=weight_decay, feature_dim=feature_dim,
                         num_iter_fit=num_iter_fit, covar_module=covar_module, mean_module=mean_module,
                         mean_nn_layers=mean_nn_layers, kernel_nn_layers=kernel_nn_layers, task_batch_size=1,
                         normalize_data=normalize_data, random_seed=random_seed, dtype=dtype)
        groups = []
        if self.kernel_net is not None:
            groups.append({'params': [q for m in self.kernel_net for q in (m.weight, m.bias)], 'weight_decay': weight_decay})
        if self.mean_net is not None:
            groups.append({'params': [q for m in self.mean_net for q in (m.weight, m.bias)], 'weight_decay': weight_decay})
        groups.append({'params': [self.raw_noise]})
        if learning_mode in ('learn_kernel', 'both'):
            groups.append({'params': [self.raw_lengthscale, self.raw_outputscale]})
        if learning_mode in ('learn_mean', 'both') and mean_module == 'constant':
            groups.append({'params': [self.constant_mean]})
        self.optimizer = torch.optim.AdamW(groups, lr=lr)              # default weight_decay = 1e-2
        self.scheduler = torch.optim.lr_scheduler.ReduceLROnPlateau(self.optimizer, mode='max',
                                                                    factor=0.2 if lr_scheduler else 1.0)
        self.train_x, self.train_t = train_x, train_t

    def hypers(self):
        ls = F.softplus(self.raw_lengthscale)
        os_ = F.softplus(self.raw_outputscale)
        noise = F.softplus(self.raw_noise) + 1e-4                      # GreaterThan(1e-4) default
        return ls, os_, noise.squeeze(0)

    def fit(self, valid_x=None, valid_t=None, log_period=500, n_iter=None):
        """GPR_mll.py:111-168 -> list of (itr, loss[, valid_ll, rmse, calib])"""
        n_iter = self.num_iter_fit if n_iter is None else n_iter
        log = []
        for itr in range(1, n_iter + 1):
            self.optimizer.zero_grad()
            loss = -self.task_mll(*self.tasks[0])
            loss.backward()
            self.optimizer.step()
            if itr == 1 or itr % log_period == 0:
                rec = (itr, float(loss.detach()))
                if valid_x is not None:
                    ev = self.eval(self.train_x, self.train_t, valid_x, valid_t)
                    self.scheduler.step(ev[0])
                    rec = rec + tuple(ev)
                log.append(rec)
        return log

    def predict_single(self, test_x):
        return self.predict_normalized(self.train_x, self.train_t, test_x)


# --------------------------------------------------------------------------------------
# synthetic task generators used by bench.py / tests
# --------------------------------------------------------------------------------------

def sinusoid_tasks_nd(n_tasks, n, d, seed0=1000):
    """d-dimensional sinusoid-of-mean tasks (parameters as
    experiments/data_sim.py:242-248): per task t, RandomState(seed0+t)."""
    tasks = []
    for t in range(n_tasks):
        rs = np.random.RandomState(seed0 + t)
        X = rs.uniform(-5, 5, size=(n, d))
        amp = rs.uniform(0.7, 1.3)
        x_shift = rs.normal(0.0, 0.1)
        y_shift = rs.normal(5.0, 0.1)
        slope = rs.normal(0.5, 0.2)
        xm = X.mean(axis=1, keepdims=True)
        Y = slope * xm + amp * np.sin(1.5 * (xm - x_shift)) + y_shift + 0.1 * rs.normal(size=(n, 1))
        tasks.append((X, Y))
    return tasks


class SinusoidDataset:
    """Restatement of experiments/data_sim.py:203-248 (same RNG call order), so that tests and
    the demo golden trajectory do not need the reference on the GPU box."""

    def __init__(self, random_state, amp_low=0.7, amp_high=1.3, period_low=1.5, period_high=1.5,
                 x_shift_mean=0.0, x_shift_std=0.1, y_shift_mean=5.0, y_shift_std=0.1,
                 slope_mean=0.5, slope_std=0.2, noise_std=0.1, x_low=-5, x_high=5):
        self.rs = random_state
        self.__dict__.update(locals())

    def _sample_sinusoid(self):
        amp = self.rs.uniform(self.amp_low, self.amp_high)
        x_shift = self.rs.normal(loc=self.x_shift_mean, scale=self.x_shift_std)
        y_shift = self.rs.normal(loc=self.y_shift_mean, scale=self.y_shift_std)
        slope = self.rs.normal(loc=self.slope_mean, scale=self.slope_std)
        period = self.rs.uniform(self.period_low, self.period_high)
        return lambda x: slope * x + amp * np.sin(period * (x - x_shift)) + y_shift

    def generate_meta_train_data(self, n_tasks, n_samples):
        out = []
        for _ in range(n_tasks):
            f = self._sample_sinusoid()
            X = self.rs.uniform(self.x_low, self.x_high, size=(n_samples, 1))
            Y = f(X) + self.noise_std * self.rs.normal(size=f(X).shape)
            out.append((X, Y))
        return out

    def generate_meta_test_data(self, n_tasks, n_samples_context, n_samples_test):
        out = []
        for _ in range(n_tasks):
            f = self._sample_sinusoid()
            X = self.rs.uniform(self.x_low, self.x_high, size=(n_samples_context + n_samples_test, 1))
            Y = f(X) + self.noise_std * self.rs.normal(size=f(X).shape)
            c = n_samples_context
            out.append((X[:c], Y[:c], X[c:], Y[c:]))
        return out
