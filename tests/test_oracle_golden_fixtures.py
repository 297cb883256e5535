"""Pin the CPU oracle against fixtures produced by the REAL reference code
(tests/golden/make_golden.py: svgd.py as shipped; models.py / random_gp.py under import shims)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import pacoh_oracle as O

CASES = {
    'nn_nn_d4': dict(input_dim=4, covar_module='NN', mean_module='NN'),
    'se_const_d4': dict(input_dim=4, covar_module='SE', mean_module='constant'),
    'se_nn_d1': dict(input_dim=1, covar_module='SE', mean_module='NN'),
    'nn_const_d2_small': dict(input_dim=2, covar_module='NN', mean_module='constant',
                              kernel_nn_layers=(8, 12)),
}


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_param_layout_matches_reference(golden_dir):
    with open(os.path.join(golden_dir, 'param_layouts.json')) as f:
        ref = json.load(f)
    for tag, kw in CASES.items():
        cfg = O.GPConfig(**kw)
        assert [(k, v) for k, v in cfg.layout.items()] == [tuple(e) for e in ref[tag]], tag
    assert O.GPConfig(**CASES['nn_nn_d4']).D == 2534
    assert O.GPConfig(**CASES['se_const_d4']).D == 6
    assert O.GPConfig(**CASES['se_nn_d1']).D == 1155


def test_hyperprior_sample_and_logprob_match_reference(golden_dir):
    fx = _load(golden_dir, 'random_gp_ref.npz')
    for tag, kw in CASES.items():
        cfg = O.GPConfig(**kw)
        pm, ps = O.hyperprior_mean_std(cfg.layout, 0.5, 3.0)
        torch.manual_seed(11)
        theta = O.hyperprior_sample(cfg.layout, pm, ps, 6)
        np.testing.assert_array_equal(theta.numpy(), fx[tag + '_theta'])       # same RNG stream
        lp = O.hyperprior_log_prob(theta, pm.float(), ps.float())
        np.testing.assert_allclose(lp.numpy(), fx[tag + '_logprior'], rtol=2e-6)


def test_vectorized_mlp_forward_matches_reference(golden_dir):
    fx = _load(golden_dir, 'random_gp_ref.npz')
    for tag, kw in CASES.items():
        cfg = O.GPConfig(**kw)
        theta = torch.from_numpy(fx[tag + '_theta'])
        x = torch.from_numpy(fx[tag + '_x'])
        if cfg.mean_module == 'NN':
            out = O.mlp_vectorized_forward(x, cfg.block(theta, 'mean_nn.'), cfg.input_dim, 1, cfg.mean_nn_layers)
            np.testing.assert_allclose(out.numpy(), fx[tag + '_mean_out'], rtol=1e-5, atol=1e-5)
        if cfg.covar_module == 'NN':
            out = O.mlp_vectorized_forward(x, cfg.block(theta, 'kernel_nn.'), cfg.input_dim, 2, cfg.kernel_nn_layers)
            np.testing.assert_allclose(out.numpy(), fx[tag + '_kernel_out'], rtol=1e-5, atol=1e-5)


DEEP_CASES = ['v4x32_d1_o2', 'v4x32_d4_o1', 'v3x32_d2_o2', 'v4x128_d2_o2', 'v_irregular_d3_o3']


@pytest.mark.parametrize('tag', DEEP_CASES + ['s4x128'])
def test_deep_mlp_forward_and_gradient_match_reference(golden_dir, tag):
    """the launchers' 4 x 32 / 4 x 128 networks and an irregular layer_sizes tuple: outputs and parameter gradients of the real
    NeuralNetworkVectorized / NeuralNetwork (deep_mlp_ref.npz) vs the oracle's restatement + autograd"""
    fx = _load(golden_dir, 'deep_mlp_ref.npz')
    if tag == 's4x128':
        layers, theta = (128, 128, 128, 128), torch.from_numpy(fx['s4x128_theta']).reshape(1, -1)
        g = torch.from_numpy(fx['s4x128_g']).unsqueeze(0)
        ref_out, ref_grad = fx['s4x128_out'][None], fx['s4x128_grad'][None]
    else:
        layers, theta = tuple(int(v) for v in fx[tag + '_layers']), torch.from_numpy(fx[tag + '_theta'])
        g = torch.from_numpy(fx[tag + '_g'])
        ref_out, ref_grad = fx[tag + '_out'], fx[tag + '_grad']
    x = torch.from_numpy(fx[tag + '_x'])
    theta = theta.clone().requires_grad_(True)
    out = O.mlp_vectorized_forward(x, theta, x.shape[1], g.shape[-1], layers)
    assert np.allclose(out.detach().numpy(), ref_out, rtol=2e-5, atol=2e-5)
    (out * g).sum().backward()
    assert np.linalg.norm(theta.grad.numpy() - ref_grad) < 1e-5 * np.linalg.norm(ref_grad)


def test_svgd_particle_init_stream(golden_dir):
    fx = _load(golden_dir, 'random_gp_ref.npz')
    cfg = O.GPConfig(input_dim=1, covar_module='NN', mean_module='NN')
    pm, ps = O.hyperprior_mean_std(cfg.layout, 0.5, 3.0)
    torch.manual_seed(30)
    O.consume_vectorized_gp_init_rng(cfg)        # model construction precedes the prior draw
    theta = O.hyperprior_sample(cfg.layout, pm, ps, 10)
    np.testing.assert_array_equal(theta.numpy(), fx['svgd_init_seed30_d1_P10'])


def test_prefactor_ragged(golden_dir):
    fx = _load(golden_dir, 'random_gp_ref.npz')
    assert abs(O.meta_pre_factor([5, 7, 12, 5]) - float(fx['prefactor_ragged'])) < 1e-7


def test_svgd_phi_closed_form_matches_reference(golden_dir):
    fx = _load(golden_dir, 'svgd_ref.npz')
    for tag in ['small_median', 'small_fixed', 'cfg3_median', 'cfg3_fixed', 'se_median']:
        X, score = torch.from_numpy(fx[tag + '_X']), torch.from_numpy(fx[tag + '_score'])
        bw_arg = float(fx[tag + '_bw_arg'])
        phi, bw = O.svgd_phi_closed_form(X, score, None if bw_arg < 0 else bw_arg)
        assert abs(bw - float(fx[tag + '_bw'])) <= 1e-6 * abs(bw)
        ref = fx[tag + '_phi']
        assert np.abs(phi.numpy() - ref).max() <= 2e-5 * np.abs(ref).max(), tag
        # float64: closed form == autograd of the reference to round-off
        phi64, _ = O.svgd_phi_closed_form(X.double(), torch.from_numpy(fx[tag + '_score64']), None if bw_arg < 0 else bw_arg)
        ref64 = fx[tag + '_phi64']
        assert np.abs(phi64.numpy() - ref64).max() <= 1e-12 * np.abs(ref64).max(), tag


def test_mixture_mean_std_cdf(golden_dir):
    fx = _load(golden_dir, 'random_gp_ref.npz')
    mus, sig = torch.from_numpy(fx['mix_mus']), torch.from_numpy(fx['mix_sig'])
    cov = torch.diag_embed(sig ** 2)
    mu, std = O.mixture_mean_std(mus, cov, np.zeros(1), np.ones(1))
    np.testing.assert_allclose(mu.numpy(), fx['mix_mean'], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(std.numpy(), fx['mix_std'], rtol=1e-5)
    cdf = torch.distributions.Normal(mus, sig).cdf(torch.from_numpy(fx['mix_val'])).mean(0)
    np.testing.assert_allclose(cdf.numpy(), fx['mix_cdf'], rtol=1e-6)


def test_mixture_cdf_icdf_and_calibration(golden_dir):
    """oracle cdf / bisection quantiles / calibration error vs EqualWeightedMixtureDist, AffineTransformedDistribution and
    _calib_error of the reference (fixture made by make_golden.py mixture_quantiles from the real classes)"""
    fx = _load(golden_dir, 'mixture_quantiles_ref.npz')
    for dt, tol in ((torch.float32, 1e-5), (torch.float64, 1e-5)):          # the reference bisects to 1e-6 on an fp32 cdf (a few ulp at y ~ 8)
        mus, var = torch.from_numpy(fx['mus']).to(dt), torch.from_numpy(fx['sig']).to(dt) ** 2
        cdf = O.mixture_cdf(mus, var, fx['val'], fx['y_mean'], fx['y_std'])
        np.testing.assert_allclose(cdf.numpy(), fx['mix_cdf'], rtol=2e-5, atol=2e-7)
        for key, q in (('mix_icdf', fx['q']), ('mix_icdf_05', np.full(40, 0.05, np.float32)), ('mix_icdf_95', np.full(40, 0.95, np.float32))):
            x = O.mixture_icdf(mus, var, q, fx['y_mean'], fx['y_std'])
            np.testing.assert_allclose(x.numpy(), fx[key], rtol=0, atol=tol)
        assert abs(float(O.calib_error(cdf)) - float(fx['mix_calib'])) < 1e-6
        c1 = O.mixture_cdf(mus[:1], var[:1], fx['val'], fx['y_mean'], fx['y_std'])
        np.testing.assert_allclose(c1.numpy(), fx['single_cdf'], rtol=2e-5, atol=2e-7)
        np.testing.assert_allclose(O.gaussian_icdf(mus[0], var[0], fx['q'], fx['y_mean'], fx['y_std']).numpy(), fx['single_icdf'], rtol=2e-6)
        assert abs(float(O.calib_error(c1)) - float(fx['single_calib'])) < 1e-6


def test_sinusoid_dataset_and_task_sampling(golden_dir):
    fx = _load(golden_dir, 'sinusoid_demo_data.npz')
    env = O.SinusoidDataset(np.random.RandomState(26))
    train = env.generate_meta_train_data(20, 5)
    test = env.generate_meta_test_data(20, 5, 50)
    np.testing.assert_array_equal(np.stack([x for x, _ in train]), fx['train_x'])
    np.testing.assert_array_equal(np.stack([y for _, y in train]), fx['train_y'])
    np.testing.assert_array_equal(np.stack([t[2] for t in test]), fx['test_tx'])
    np.testing.assert_array_equal(np.stack([t[3] for t in test]), fx['test_ty'])
    rds = np.random.RandomState(31)
    draws = np.stack([rds.randint(0, 20, 5) for _ in range(4)])
    np.testing.assert_array_equal(draws, fx['choice_seed31'])
    assert list(draws[0]) == [18, 16, 2, 6, 10]


# ---- IMQ particle kernel (svgd.py:63-97), fixtures generated from the reference's svgd.py as shipped ----
IMQ_TAGS = ['small_median', 'small_fixed', 'tiny_median', 'cfg3_median', 'p20_fixed', 'p10_median']


def imq_case(golden_dir, tag, dt):
    fx = np.load(os.path.join(golden_dir, 'svgd_imq_ref.npz'))
    X, mu, s = (torch.from_numpy(fx[tag + k]).to(dt) for k in ('_X', '_mu', '_s'))
    sfx = '64' if dt == torch.float64 else ''
    bw = float(fx[tag + '_bw_arg'])
    return X, -(X - mu) / s ** 2, (None if bw < 0 else bw), torch.from_numpy(fx[tag + '_phi' + sfx]), torch.from_numpy(fx[tag + '_K' + sfx])


@pytest.mark.parametrize('tag', IMQ_TAGS)
@pytest.mark.parametrize('dt', [torch.float32, torch.float64])
def test_oracle_imq_phi_matches_reference(golden_dir, tag, dt):
    X, score, bw, phi_ref, _ = imq_case(golden_dir, tag, dt)
    phi, _ = O.svgd_phi_imq_closed_form(X, score, bandwidth=bw)
    tol = 1e-12 if dt == torch.float64 else 2e-5
    assert float((phi - phi_ref).norm() / phi_ref.norm()) < tol


def test_oracle_vi_full_covariance_matches_reference(golden_dir):
    """RandomGPPosterior(cov_type='full') of the real reference: init stream, rsample, log_prob, autograd gradient"""
    fx = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(golden_dir, 'vi_full_ref.npz')).items()}
    cfg = O.GPConfig(2, 'constant', 'NN', kernel_nn_layers=(4,))
    D = fx['init_loc'].shape[0]
    assert cfg.D == D
    torch.manual_seed(30)
    O.consume_vectorized_gp_init_rng(cfg)
    loc, tril = O.vi_full_init(D)
    assert torch.equal(loc, fx['init_loc']) and torch.equal(tril, fx['init_tril'])
    eps = torch.normal(torch.zeros(6, D), torch.ones(6, D))
    theta, log_q = O.vi_full_sample(loc, fx['tril'], eps)
    assert float((theta - fx['rsample']).abs().max()) < 1e-6
    assert float((log_q - fx['logq']).abs().max()) < 2e-5 * float(fx['logq'].abs().max())
    score = -(theta @ fx['A'].t()) @ fx['A']
    g_loc, g_tril = O.vi_full_grad(fx['tril'], eps, score, 0.01)
    assert float((g_loc - fx['grad_loc']).norm() / fx['grad_loc'].norm()) < 1e-5
    assert float((g_tril - fx['grad_tril']).norm() / fx['grad_tril'].norm()) < 1e-5
    assert float(torch.triu(g_tril, 1).abs().max()) == 0.0
    loss = -(-0.5 * ((theta @ fx['A'].t()) ** 2).sum(-1) - 0.01 * log_q).mean()
    assert abs(float(loss) - float(fx['loss'])) < 1e-5
