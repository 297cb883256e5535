"""Generate the golden fixtures in tests/golden/*.npz from the REAL reference code.

Run only in the build container (needs /root/reference; nothing here travels to or runs on the
GPU box -- only the produced .npz/.json data files do):

    python tests/golden/make_golden.py

What is imported from the reference:
  * meta_learn/svgd.py            -- imports as shipped (numpy + torch only)          -> A9 fixtures
  * meta_learn/models.py, random_gp.py -- imported under *import shims* for the absent
    third-party modules gpytorch / pyro / absl (dummy base classes only; they do not compute
    anything).  Gives the real parameter layout, hyper-prior sampling / log-prob and the
    per-particle MLP forward                                                        -> A2/A7 fixtures
  * experiments/data_sim.py       -- SinusoidDataset (numpy only)                     -> A1 fixtures
The shims cannot execute the GP algebra itself (A4-A6 *is* gpytorch); that part of the oracle is
pinned by the recorded demo.ipynb log instead (demo_log.json, transcribed here from
demo.ipynb:115-127,164-166).
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch

REF = '/root/reference'
OUT = os.path.dirname(os.path.abspath(__file__))


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def install_shims():
    """Dummy stand-ins so that `import gpytorch/pyro/absl` succeed; only class *names* used as
    base classes by models.py / random_gp.py exist, none of them computes anything."""
    def mod(name):
        m = types.ModuleType(name)
        sys.modules[name] = m
        return m

    class _Dummy(torch.nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

    g = mod('gpytorch')
    for sub in ['means', 'kernels', 'functions', 'utils', 'utils.broadcasting', 'likelihoods',
                'likelihoods.noise_models', 'models', 'models.approximate_gp', 'variational',
                'mlls', 'distributions']:
        m = mod('gpytorch.' + sub)
        parent = sys.modules['gpytorch.' + sub.rsplit('.', 1)[0]] if '.' in sub else g
        setattr(parent, sub.rsplit('.', 1)[-1], m)
    g.means.Mean = _Dummy
    g.means.ZeroMean = _Dummy
    g.kernels.Kernel = _Dummy
    g.functions.RBFCovariance = object
    g.utils.broadcasting._mul_broadcast_shape = lambda *a: None
    g.likelihoods.noise_models._HomoskedasticNoiseBase = _Dummy
    g.likelihoods._GaussianLikelihoodBase = _Dummy
    g.models.ExactGP = _Dummy
    g.models.approximate_gp.ApproximateGP = _Dummy
    g.variational.CholeskyVariationalDistribution = _Dummy
    g.variational.VariationalStrategy = _Dummy

    p = mod('pyro')
    pd = mod('pyro.distributions')
    p.distributions = pd

    class Normal(torch.distributions.Normal):                 # pyro adds .to_event()
        def to_event(self, n):
            return torch.distributions.Independent(self, n)
    pd.Normal = Normal
    pd.LogNormal = torch.distributions.LogNormal
    pd.Independent = torch.distributions.Independent

    a = mod('absl')
    fl = mod('absl.flags')
    a.flags = fl
    fl.FLAGS = types.SimpleNamespace(is_parsed=lambda: False)

    cfg = mod('config')
    cfg.device = torch.device('cpu')


def make_vi_full():
    """RandomGPPosterior(cov_type='full') (random_gp.py:249-251): init stream, rsample, log_prob and the autograd
    gradient of a reparameterised objective -> vi_full_ref.npz"""
    install_shims()
    sys.path.insert(0, REF)
    import meta_learn.random_gp as random_gp
    fx = {}
    torch.manual_seed(30)
    rgp = random_gp.RandomGPMeta(size_in=2, prior_factor=0.01, weight_prior_std=0.5, bias_prior_std=3.0,
                                 covar_module_str='NN', mean_module_str='constant', kernel_nn_layers=(4,))
    post = random_gp.RandomGPPosterior(rgp.parameter_shapes(), cov_type='full')
    fx['init_loc'] = post.loc.detach().numpy()
    fx['init_tril'] = post.tril_cov.detach().numpy().copy()
    # perturb the strictly-lower AND upper parts so that the tril() mask and the off-diagonal algebra are exercised
    gen = torch.Generator().manual_seed(5)
    with torch.no_grad():
        post.tril_cov.add_(0.008 * torch.randn(post.tril_cov.shape, generator=gen))
    fx['tril'] = post.tril_cov.detach().numpy()
    samp = post.rsample(sample_shape=(6,))
    fx['rsample'] = samp.detach().numpy()
    logq = post.log_prob(samp)
    fx['logq'] = logq.detach().numpy()
    # stand-in log-density with a known score: log p(theta) = -0.5 |A theta|^2  (the GP part is pinned elsewhere)
    D = post.loc.shape[0]
    A = torch.randn(D, D, generator=gen) / D ** 0.5
    fx['A'] = A.numpy()
    loss = -(-0.5 * ((samp @ A.t()) ** 2).sum(-1) - 0.01 * logq).mean()
    loss.backward()
    fx['loss'] = np.array(loss.item())
    fx['grad_loc'] = post.loc.grad.numpy()
    fx['grad_tril'] = post.tril_cov.grad.numpy()
    np.savez_compressed(os.path.join(OUT, 'vi_full_ref.npz'), **fx)


def make_mixture_quantiles():
    """EqualWeightedMixtureDist.cdf / .icdf (models.py:127-140, bisection of util.py:9-42), the single-Gaussian
    AffineTransformedDistribution cdf / icdf (models.py:15-43) and _calib_error (abstract.py:260-272) -> mixture_quantiles_ref.npz"""
    install_shims()
    sys.path.insert(0, REF)
    import meta_learn.models as models
    import meta_learn.abstract as abstract
    fx = {}
    gen = torch.Generator().manual_seed(11)
    P, m = 6, 40
    mus, sig = torch.randn(P, m, generator=gen) * 0.8 + 1.5, torch.rand(P, m, generator=gen) * 0.6 + 0.05
    y_mean, y_std = np.array([4.2]), np.array([1.7])
    base = torch.distributions.Normal(mus, sig)
    affine = models.AffineTransformedDistribution(base, normalization_mean=y_mean, normalization_std=y_std)
    mix = models.EqualWeightedMixtureDist(affine, batched=True, num_dists=P)
    val = torch.randn(m, generator=gen) * 2.0 + 6.5
    q = torch.rand(m, generator=gen) * 0.98 + 0.01
    fx['mus'], fx['sig'], fx['y_mean'], fx['y_std'] = mus.numpy(), sig.numpy(), y_mean, y_std
    fx['val'], fx['q'] = val.numpy(), q.numpy()
    fx['mix_cdf'] = mix.cdf(val).numpy()
    fx['mix_icdf'] = mix.icdf(q.clone()).numpy()
    fx['mix_icdf_05'] = mix.icdf(torch.ones(m) * 0.05).numpy()
    fx['mix_icdf_95'] = mix.icdf(torch.ones(m) * 0.95).numpy()
    fx['mix_calib'] = np.array(abstract._calib_error(mix, val).item())
    single = models.AffineTransformedDistribution(torch.distributions.Normal(mus[0], sig[0]), normalization_mean=y_mean,
                                                  normalization_std=y_std)
    fx['single_cdf'] = single.cdf(val).numpy()
    fx['single_icdf'] = single.icdf(q).numpy()
    fx['single_calib'] = np.array(abstract._calib_error(single, val).item())
    np.savez_compressed(os.path.join(OUT, 'mixture_quantiles_ref.npz'), **fx)
    print('mixture_quantiles_ref.npz written')


def make_deep_mlp():
    """The networks the reference's launchers actually build -- NeuralNetworkVectorized 4 x 32 (experiments/
    meta_GPR_SVGD_base_exp.py:29-30,83), the shared NeuralNetwork 4 x 128 of PACOH-MAP (meta_GPR_mll_base_exp.py:29-30) and an
    irregular layer_sizes tuple (models.py:328-349 takes any) -- forward outputs and the autograd gradient of sum(out * g)
    with respect to the flattened parameters -> deep_mlp_ref.npz"""
    install_shims()
    sys.path.insert(0, REF)
    import meta_learn.models as models
    fx = {}
    cases = {'v4x32_d1_o2': (1, 2, (32, 32, 32, 32), 5, 23), 'v4x32_d4_o1': (4, 1, (32, 32, 32, 32), 3, 70),
             'v3x32_d2_o2': (2, 2, (32, 32, 32), 4, 19), 'v4x128_d2_o2': (2, 2, (128, 128, 128, 128), 2, 37),
             'v_irregular_d3_o3': (3, 3, (40, 17, 128, 9, 64), 3, 21)}
    for tag, (d_in, d_out, layers, P, n) in cases.items():
        gen = torch.Generator().manual_seed(len(tag) + d_in + n)
        net = models.NeuralNetworkVectorized(d_in, d_out, layer_sizes=layers)
        D = sum(int(v[-1]) for v in net.parameter_shapes().values())
        scale = torch.cat([torch.full((int(v[-1]),), 3.0 if k.endswith('bias') else 0.5) for k, v in net.parameter_shapes().items()])
        theta = (torch.randn(P, D, generator=gen) * scale * 0.6).requires_grad_(True)
        net.set_parameters_as_vector(theta)
        x = torch.randn(n, d_in, generator=gen)
        g = torch.randn(P, n, d_out, generator=gen)
        out = net(x)                                          # 2-D inputs are tiled over the P parameter sets (models.py:305-309)
        (out * g).sum().backward()
        fx[tag + '_theta'], fx[tag + '_x'], fx[tag + '_g'] = theta.detach().numpy(), x.numpy(), g.numpy()
        fx[tag + '_out'], fx[tag + '_grad'] = out.detach().numpy(), theta.grad.numpy()
        fx[tag + '_layers'] = np.array(layers)
    # the shared-weight torch.nn network of PACOH-MAP at the launcher's 4 x 128, its parameters flattened bias-before-weight
    torch.manual_seed(28)
    net = models.NeuralNetwork(input_dim=1, output_dim=2, layer_sizes=(128, 128, 128, 128))
    names = ['fc_1', 'fc_2', 'fc_3', 'fc_4', 'out']
    gen = torch.Generator().manual_seed(4)
    x = torch.randn(10, 1, generator=gen)                     # 2 tasks x 5 points, the launcher's batch (meta_GPR_mll_base_exp.py:33,40)
    g = torch.randn(10, 2, generator=gen)
    out = net(x)
    (out * g).sum().backward()
    fx['s4x128_theta'] = torch.cat([torch.cat([getattr(net, k).bias.detach().reshape(-1), getattr(net, k).weight.detach().reshape(-1)])
                                    for k in names]).numpy()
    fx['s4x128_grad'] = torch.cat([torch.cat([getattr(net, k).bias.grad.reshape(-1), getattr(net, k).weight.grad.reshape(-1)])
                                   for k in names]).numpy()
    fx['s4x128_x'], fx['s4x128_g'], fx['s4x128_out'] = x.numpy(), g.numpy(), out.detach().numpy()
    np.savez_compressed(os.path.join(OUT, 'deep_mlp_ref.npz'), **fx)
    print('deep_mlp_ref.npz written')


def make_imq_large():
    """IMQ-SVGD beyond 64 particles (round 3: the IMQ entry point takes up to 1024 like the RBF one): phi of the REAL
    meta_learn/svgd.py (SVGD.phi + IMQSteinKernel, svgd.py:12-23,63-99), median and fixed bandwidth, fp32 and fp64
        python tests/golden/make_golden.py imq_large"""
    svgd = _load('ref_svgd', os.path.join(REF, 'meta_learn', 'svgd.py'))

    class QuadLogProb:
        def __init__(self, mu, s):
            self.mu, self.s = mu, s

        def log_prob(self, X):
            return (-0.5 * ((X - self.mu) / self.s) ** 2).sum(-1)

    fx = {}
    for tag, (P, D, bw) in {'p80_median': (80, 150, None), 'p130_fixed': (130, 64, 1.2), 'p200_median': (200, 40, None),
                            'p65_median_wide': (65, 260, None)}.items():
        gen = torch.Generator().manual_seed(7 + P + D)
        X = torch.randn(P, D, generator=gen) * 0.7
        mu = torch.randn(D, generator=gen)
        s = torch.rand(D, generator=gen) + 0.5
        for sfx, dt in (('', torch.float32), ('64', torch.float64)):
            Xd, mud, sd = X.to(dt), mu.to(dt), s.to(dt)
            kern = svgd.IMQSteinKernel(bandwidth=bw)
            phi = svgd.SVGD(QuadLogProb(mud, sd), kern, optimizer=None).phi(Xd)
            fx[tag + '_phi' + sfx] = phi.detach().numpy()
        fx[tag + '_X'], fx[tag + '_mu'], fx[tag + '_s'] = X.numpy(), mu.numpy(), s.numpy()   # score = -(X - mu) / s^2
        fx[tag + '_bw_arg'] = np.array(-1.0 if bw is None else bw)
    np.savez_compressed(os.path.join(OUT, 'svgd_imq_large_ref.npz'), **fx)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == 'imq_large':
        return make_imq_large()
    if len(sys.argv) > 1 and sys.argv[1] == 'mixture_quantiles':
        return make_mixture_quantiles()
    if len(sys.argv) > 1 and sys.argv[1] == 'deep_mlp':
        return make_deep_mlp()
    if len(sys.argv) > 1 and sys.argv[1] == 'vi_full':
        return make_vi_full()
    assert os.path.isdir(REF), 'reference not mounted -- fixtures can only be regenerated in the build container'
    sys.path.insert(0, REF)

    # ------------------------------------------------------------------ A9: svgd.py as shipped
    svgd = _load('ref_svgd', os.path.join(REF, 'meta_learn', 'svgd.py'))

    class QuadLogProb:
        """stub target density: log p(x) = -0.5 * sum((x-mu)^2 / s^2) (so that score is known)"""
        def __init__(self, mu, s):
            self.mu, self.s = mu, s

        def log_prob(self, X):
            return (-0.5 * ((X - self.mu) / self.s) ** 2).sum(-1)

    fx = {}
    for tag, (P, D, bw) in {'small_median': (5, 7, None), 'small_fixed': (5, 7, 0.5),
                            'cfg3_median': (20, 2534, None), 'cfg3_fixed': (20, 2534, 2.0),
                            'se_median': (10, 6, None)}.items():
        gen = torch.Generator().manual_seed(1234 + P + D)
        X = torch.randn(P, D, generator=gen) * 0.7
        mu = torch.randn(D, generator=gen)
        s = torch.rand(D, generator=gen) + 0.5
        dist = QuadLogProb(mu, s)
        kern = svgd.RBF_Kernel(bandwidth=bw)
        phi = svgd.SVGD(dist, kern, optimizer=None).phi(X)
        score = -(X - mu) / s ** 2
        dn = svgd.norm_sq(X, X)
        fx[tag + '_X'] = X.numpy()
        fx[tag + '_score'] = score.numpy()
        fx[tag + '_phi'] = phi.detach().numpy()
        fx[tag + '_bw'] = np.array(kern._bandwidth(dn), dtype=np.float64)
        fx[tag + '_K'] = kern(X, X).detach().numpy()
        fx[tag + '_bw_arg'] = np.array(-1.0 if bw is None else bw)
        # float64 version of the same call
        X64, mu64, s64 = X.double(), mu.double(), s.double()
        phi64 = svgd.SVGD(QuadLogProb(mu64, s64), svgd.RBF_Kernel(bandwidth=bw), None).phi(X64)
        fx[tag + '_phi64'] = phi64.detach().numpy()
        fx[tag + '_score64'] = (-(X64 - mu64) / s64 ** 2).numpy()
    np.savez_compressed(os.path.join(OUT, 'svgd_ref.npz'), **fx)

    # ------------------------------------------------------------------ A9 (IMQ particle kernel)
    fx = {}
    for tag, (P, D, bw) in {'small_median': (5, 7, None), 'small_fixed': (5, 7, 0.5), 'tiny_median': (4, 3, None),
                            'cfg3_median': (20, 2534, None), 'p20_fixed': (20, 300, 1.5),
                            'p10_median': (10, 642, None)}.items():
        gen = torch.Generator().manual_seed(99 + P + D)
        X = torch.randn(P, D, generator=gen) * 0.7
        mu = torch.randn(D, generator=gen)
        s = torch.rand(D, generator=gen) + 0.5
        for sfx, dt in (('', torch.float32), ('64', torch.float64)):
            Xd, mud, sd = X.to(dt), mu.to(dt), s.to(dt)
            kern = svgd.IMQSteinKernel(bandwidth=bw)
            phi = svgd.SVGD(QuadLogProb(mud, sd), kern, optimizer=None).phi(Xd)
            fx[tag + '_phi' + sfx] = phi.detach().numpy()
            fx[tag + '_K' + sfx] = kern(Xd, Xd).detach().numpy()
        fx[tag + '_X'], fx[tag + '_mu'], fx[tag + '_s'] = X.numpy(), mu.numpy(), s.numpy()   # score = -(X - mu) / s^2
        fx[tag + '_bw_arg'] = np.array(-1.0 if bw is None else bw)
    np.savez_compressed(os.path.join(OUT, 'svgd_imq_ref.npz'), **fx)
    if len(sys.argv) > 1 and sys.argv[1] == 'imq':
        return

    # ------------------------------------------------------------------ A2/A7 under import shims
    install_shims()
    import meta_learn.models as models                         # noqa: E402
    import meta_learn.random_gp as random_gp                   # noqa: E402

    fx = {}
    cases = {
        'nn_nn_d4': dict(size_in=4, covar_module_str='NN', mean_module_str='NN'),
        'se_const_d4': dict(size_in=4, covar_module_str='SE', mean_module_str='constant'),
        'se_nn_d1': dict(size_in=1, covar_module_str='SE', mean_module_str='NN'),
        'nn_const_d2_small': dict(size_in=2, covar_module_str='NN', mean_module_str='constant',
                                  kernel_nn_layers=(8, 12)),
    }
    layouts = {}
    for tag, kw in cases.items():
        torch.manual_seed(7)
        rgp = random_gp.RandomGPMeta(prior_factor=0.01, weight_prior_std=0.5, bias_prior_std=3.0, **kw)
        shapes = rgp.parameter_shapes()
        layouts[tag] = [(k, int(v[0])) for k, v in shapes.items()]
        torch.manual_seed(11)
        P = 6
        theta = rgp.sample_params_from_prior(shape=(P,))
        fx[tag + '_theta'] = theta.numpy()
        fx[tag + '_logprior'] = rgp._log_prob_prior(theta).numpy()
        # per-particle MLP forward through the reference's NeuralNetworkVectorized
        gp = rgp.get_forward_fn(theta)
        x = torch.randn(9, kw['size_in'], generator=torch.Generator().manual_seed(5))
        xP = x.view(1, 9, -1).repeat(P, 1, 1)
        fx[tag + '_x'] = x.numpy()
        if kw['mean_module_str'] == 'NN':
            fx[tag + '_mean_out'] = gp.mean_nn(xP).detach().numpy()
        if kw['covar_module_str'] == 'NN':
            fx[tag + '_kernel_out'] = gp.kernel_nn(xP).detach().numpy()
    # the SVGD learner's particle initialisation stream: torch.manual_seed(seed) then one prior draw
    # (GPR_meta_svgd.py:182 after abstract.py:125-129)
    torch.manual_seed(30)
    rgp = random_gp.RandomGPMeta(size_in=1, prior_factor=0.01, weight_prior_std=0.5, bias_prior_std=3.0,
                                 covar_module_str='NN', mean_module_str='NN')
    fx['svgd_init_seed30_d1_P10'] = rgp.sample_params_from_prior(shape=(10,)).numpy()
    # VI posterior init + one rsample (random_gp.py:244-248, GPR_meta_vi.py:220)
    torch.manual_seed(30)
    rgp = random_gp.RandomGPMeta(size_in=1, prior_factor=0.01, weight_prior_std=0.5, bias_prior_std=3.0,
                                 covar_module_str='SE', mean_module_str='constant')
    post = random_gp.RandomGPPosterior(rgp.parameter_shapes(), cov_type='diag')
    samp = post.rsample(sample_shape=(4,))
    fx['vi_init_loc'] = post.loc.detach().numpy()
    fx['vi_init_scale'] = post.scale.detach().numpy()
    fx['vi_rsample'] = samp.detach().numpy()
    fx['vi_logq'] = post.log_prob(samp).detach().numpy()
    # harmonic-mean pre-factor on a ragged batch (random_gp.py:209-212), arithmetic only
    sizes = torch.tensor([5., 7., 12., 5.])
    hm = 1. / (torch.mean(1. / sizes))
    fx['prefactor_ragged'] = np.array(float(hm / (hm + 4)))
    # EqualWeightedMixtureDist mean/std/log_prob/cdf on a batched Normal (models.py:74-134)
    gen = torch.Generator().manual_seed(3)
    mus, sig = torch.randn(5, 8, generator=gen), torch.rand(5, 8, generator=gen) + 0.3
    mix = models.EqualWeightedMixtureDist(torch.distributions.Normal(mus, sig), batched=True, num_dists=5)
    val = torch.randn(8, generator=gen)
    fx['mix_mus'], fx['mix_sig'], fx['mix_val'] = mus.numpy(), sig.numpy(), val.numpy()
    fx['mix_mean'], fx['mix_std'] = mix.mean.numpy(), mix.stddev.numpy()
    fx['mix_cdf'] = mix.cdf(val).numpy()
    np.savez_compressed(os.path.join(OUT, 'random_gp_ref.npz'), **fx)
    with open(os.path.join(OUT, 'param_layouts.json'), 'w') as f:
        json.dump(layouts, f, indent=1)

    # ------------------------------------------------------------------ A1/A12: data + sampling
    import experiments.data_sim as data_sim                      # noqa: E402
    env = data_sim.SinusoidDataset(random_state=np.random.RandomState(26))
    train = env.generate_meta_train_data(n_tasks=20, n_samples=5)
    test = env.generate_meta_test_data(n_tasks=20, n_samples_context=5, n_samples_test=50)
    fx = {'train_x': np.stack([x for x, _ in train]), 'train_y': np.stack([y for _, y in train]),
          'test_cx': np.stack([t[0] for t in test]), 'test_cy': np.stack([t[1] for t in test]),
          'test_tx': np.stack([t[2] for t in test]), 'test_ty': np.stack([t[3] for t in test])}
    # rds_numpy.choice(list_of_dicts, size=B) == randint(0, T, B) (abstract.py:125-129, GPR_meta_mll.py:109)
    rds = np.random.RandomState(31)
    fx['choice_seed31'] = np.stack([rds.choice(np.arange(20), size=5) for _ in range(4)])
    np.savez_compressed(os.path.join(OUT, 'sinusoid_demo_data.npz'), **fx)

    # ------------------------------------------------------------------ recorded run of the reference
    demo_log = {
        'source': 'demo.ipynb:115-127 (meta_fit log) and demo.ipynb:164-166 (final test metrics)',
        'config': 'SinusoidDataset(RandomState(26)) 20x5 train, 20x(5+50) test; GPRegressionMetaLearned('
                  'weight_decay=0.2, num_iter_fit=12000, random_seed=30); log_period=1000',
        'log': [  # itr, avg loss, valid LL, valid RMSE, calib err
            [1, 5.755850, -1.559, 1.284, 0.138], [1000, 4.953671, -1.198, 0.849, 0.125],
            [2000, 3.324356, -0.725, 0.514, 0.138], [3000, 1.704049, -0.464, 0.420, 0.137],
            [4000, 1.019263, -0.321, 0.419, 0.126], [5000, 0.663038, -0.208, 0.366, 0.127],
            [6000, 0.437905, -0.121, 0.363, 0.121], [7000, 0.219458, -0.075, 0.345, 0.121],
            [8000, 0.124243, -0.016, 0.312, 0.130], [9000, -0.004427, 0.027, 0.306, 0.132],
            [10000, -0.118806, 0.011, 0.311, 0.129], [11000, -0.166441, 0.036, 0.308, 0.133],
            [12000, -0.203187, 0.033, 0.309, 0.134]],
        'final_test': {'ll': 0.03293633884750306, 'rmse': 0.3090165838599205,
                       'calib': 0.13442028164863587},
    }
    # GPRegressionLearned cell of demo.ipynb (single-task baseline), transcribed
    demo_log['single_task'] = {
        'source': 'demo.ipynb (GPRegressionLearned cell output)',
        'config': "x_context, y_context, x_test, y_test = meta_test_data[0]; GPRegressionLearned(x_context, y_context, "
                  "learning_mode='learn_mean', covar_module='SE', mean_module='constant', random_seed=30).fit(x_test, y_test)",
        'log': [[1, 1.436, -1.315, 1.402, 0.290], [500, 1.436, -1.309, 1.405, 0.296], [1000, 1.436, -1.309, 1.405, 0.296]]}
    with open(os.path.join(OUT, 'demo_log.json'), 'w') as f:
        json.dump(demo_log, f, indent=1)
    make_vi_full()
    make_mixture_quantiles()
    make_deep_mlp()
    print('fixtures written to', OUT)


if __name__ == '__main__':
    main()
