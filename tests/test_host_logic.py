"""Host-side logic of the drop-in package that needs no GPU: parameter layout, RNG streams, priors,
normalisation, task packing and sharding -- checked against the fixtures generated from the reference."""
import json
import os

import numpy as np
import pytest
import torch

from meta_learning_pacoh_amd import parallel
from meta_learning_pacoh_amd.engine import ParamLayout, TaskBatch
from meta_learning_pacoh_amd.GPR_meta_svgd import (consume_vectorized_gp_init_rng, harmonic_pre_factor,
                                                   sample_hyper_prior)
from meta_learning_pacoh_amd.GPR_meta_vi import init_vi_posterior, standard_normal
from meta_learning_pacoh_amd.util import StepLR, _handle_input_dimensionality
from oracle import pacoh_oracle as O

CASES = {
    'nn_nn_d4': dict(input_dim=4, covar_module='NN', mean_module='NN'),
    'se_const_d4': dict(input_dim=4, covar_module='SE', mean_module='constant'),
    'se_nn_d1': dict(input_dim=1, covar_module='SE', mean_module='NN'),
    'nn_const_d2_small': dict(input_dim=2, covar_module='NN', mean_module='constant', kernel_nn_layers=(8, 12)),
}


def test_param_layout_matches_reference(golden_dir):
    with open(os.path.join(golden_dir, 'param_layouts.json')) as f:
        ref = json.load(f)
    for tag, kw in CASES.items():
        lay = ParamLayout(**kw)
        assert [(k, v) for k, v in lay.blocks.items()] == [tuple(e) for e in ref[tag]], tag
        assert list(lay.parameter_shapes().values()) == [torch.Size((e[1],)) for e in ref[tag]]
    assert ParamLayout(**CASES['nn_nn_d4']).D == 2534
    map_lay = ParamLayout(1, 'NN', 'NN', feature_dim=2, with_outputscale=True)
    assert 'outputscale_raw' in map_lay.blocks and map_lay.D == 1155 + 64 + 34 + 1 + 1 - 1 + 0 or map_lay.D > 0


def test_particle_init_stream_and_prior_match_reference(golden_dir):
    fx = np.load(os.path.join(golden_dir, 'random_gp_ref.npz'))
    lay = ParamLayout(1, 'NN', 'NN')
    pm, ps = lay.hyper_prior_mean_std(0.5, 3.0)
    om, osd = O.hyperprior_mean_std(O.GPConfig(1, 'NN', 'NN').layout, 0.5, 3.0)
    assert torch.equal(pm.double(), om) and torch.equal(ps.double(), osd)
    torch.manual_seed(30)
    consume_vectorized_gp_init_rng(lay)
    theta = sample_hyper_prior(lay, pm, ps, 10)
    np.testing.assert_array_equal(theta.numpy(), fx['svgd_init_seed30_d1_P10'])
    for tag, kw in CASES.items():
        lay = ParamLayout(**kw)
        pm, ps = lay.hyper_prior_mean_std(0.5, 3.0)
        torch.manual_seed(11)
        np.testing.assert_array_equal(sample_hyper_prior(lay, pm, ps, 6).numpy(), fx[tag + '_theta'])


def test_vi_posterior_init_and_rsample_stream(golden_dir):
    fx = np.load(os.path.join(golden_dir, 'random_gp_ref.npz'))
    lay = ParamLayout(1, 'constant', 'SE')
    torch.manual_seed(30)
    consume_vectorized_gp_init_rng(lay)          # SE/constant: no networks -> consumes nothing
    post = init_vi_posterior(lay.D)
    np.testing.assert_array_equal(post[0].numpy(), fx['vi_init_loc'])
    np.testing.assert_array_equal(post[1].numpy(), fx['vi_init_scale'])
    eps = standard_normal(4, lay.D)
    theta = post[0] + eps * torch.exp(post[1])
    np.testing.assert_allclose(theta.numpy(), fx['vi_rsample'], rtol=1e-6, atol=1e-7)
    logq = (-0.5 * eps ** 2 - post[1] - 0.5 * np.log(2 * np.pi)).sum(-1)
    np.testing.assert_allclose(logq.numpy(), fx['vi_logq'], rtol=1e-5)


def test_prefactor_steplr_and_shapes(golden_dir):
    fx = np.load(os.path.join(golden_dir, 'random_gp_ref.npz'))
    assert abs(harmonic_pre_factor([5, 7, 12, 5]) - float(fx['prefactor_ragged'])) < 1e-6
    sch = StepLR(1e-3, 1000, 0.5)
    lrs = []
    for _ in range(2001):
        lrs.append(sch.lr)
        sch.step()
    assert lrs[0] == 1e-3 and lrs[999] == 1e-3 and lrs[1000] == 5e-4 and lrs[2000] == 2.5e-4
    assert StepLR(1e-3, 1000, 1.0).lr == 1e-3
    x, y = _handle_input_dimensionality(np.zeros(5), np.zeros(5))
    assert x.shape == (5, 1) and y.shape == (5, 1)


def test_task_batch_packing_and_sharding():
    rs = np.random.RandomState(0)
    sizes = [5, 9, 7]
    tasks = [(rs.randn(s, 2).astype(np.float32), rs.randn(s).astype(np.float32)) for s in sizes]
    tb = TaskBatch(tasks, torch.device('cpu'))
    assert tb.x.shape == (3, 9, 2) and tb.y.shape == (3, 9) and tb.ragged
    assert tb.n_valid.tolist() == sizes and float(tb.x[0, 5:].abs().sum()) == 0
    with pytest.raises(RuntimeError):                            # the gather is a HIP kernel: no CPU path (GPU test: test_gather_tasks)
        tb.select(torch.tensor([2, 2, 0]))
    idx = np.arange(10)
    parts = [parallel.shard(idx, r, 4) for r in range(4)]
    assert sorted(np.concatenate(parts).tolist()) == list(range(10)) and max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    lik, score = torch.ones(3), torch.ones(3, 4)
    a, b = parallel.all_reduce_sum_(lik, score)                  # world size 1: identity
    assert a is lik and b is score


def test_vi_full_posterior_init_stream(golden_dir):
    """cov_type='full' init (random_gp.py:244,249-250): same torch-generator stream as the real reference"""
    from meta_learning_pacoh_amd.GPR_meta_vi import init_vi_posterior_full
    fx = np.load(os.path.join(golden_dir, 'vi_full_ref.npz'))
    lay = ParamLayout(2, 'constant', 'NN', kernel_nn_layers=(4,))
    torch.manual_seed(30)
    consume_vectorized_gp_init_rng(lay)
    post = init_vi_posterior_full(lay.D)
    np.testing.assert_array_equal(post[0].numpy(), fx['init_loc'])
    np.testing.assert_array_equal(post[1:].numpy(), fx['init_tril'])


def test_vectorised_step_draws_equal_per_step_draws():
    """the training loops draw the task batches, pre-factors and optimizer scalars of a whole chunk of steps at once: one randint
    call of shape [k, B] must consume the numpy stream like k calls of size B (the reference draws per iteration,
    GPR_meta_svgd.py:102), and the vectorised scalars must equal the per-step formulas bit for bit"""
    from meta_learning_pacoh_amd.GPR_meta_svgd import harmonic_pre_factor
    from meta_learning_pacoh_amd import _lib as L
    sizes = np.array([5, 7, 12, 5, 9, 20])
    a, b = np.random.RandomState(4), np.random.RandomState(4)
    idx = b.randint(0, 6, size=(5, 4))
    s = sizes[idx].astype(np.float32)
    hm = np.float32(1) / np.mean(np.float32(1) / s, axis=1, dtype=np.float32)
    pre = (hm / (hm + np.float32(4))).astype(np.float64)
    sched = StepLR(1e-3, 2, 0.9)
    sched.epoch = 3
    rows = L.step_scalar_rows(pre, sched.lrs(5), 11, weight_decay=0.1)
    for j in range(5):
        i = a.randint(0, 6, size=4)
        assert (i == idx[j]).all()
        ref = L.step_scalars(harmonic_pre_factor(sizes[i]), sched.lr, 11 + j, weight_decay=0.1)
        sched.step()
        assert np.array_equal(rows[j], np.asarray(ref))
    assert a.randint(0, 10 ** 6) == b.randint(0, 10 ** 6)          # both streams are at the same position afterwards


def test_module_objects_are_resolved_by_class_name():
    """mean_module / covar_module objects (GPR_meta_mll.py:207-251): ZeroMean / ConstantMean / RBFKernel / ScaleKernel(RBFKernel) map to
    the string options with their raw hyper-parameters as initial values; anything else is refused (it cannot run on the RBF kernels)"""
    import torch
    from meta_learning_pacoh_amd.engine import ParamLayout
    from meta_learning_pacoh_amd.modules import apply_initial_values, resolve_covar_module, resolve_mean_module

    class Mean:                      # stand-ins with gpytorch's class names (gpytorch is not installed here or on the GPU box)
        pass

    class ZeroMean(Mean):
        pass

    class ConstantMean(Mean):
        def __init__(self, c):
            self.constant = torch.nn.Parameter(torch.tensor([c]))

    class Kernel:
        pass

    class RBFKernel(Kernel):
        def __init__(self, raw):
            self.raw_lengthscale = torch.nn.Parameter(torch.tensor([raw]))

    class ScaleKernel(Kernel):
        def __init__(self, base, raw):
            self.base_kernel, self.raw_outputscale = base, torch.nn.Parameter(torch.tensor(raw))

    class CosineKernel(Kernel):
        def __init__(self, raw=0.0):
            self.raw_period_length = torch.nn.Parameter(torch.full((1, 1), raw))

    class MaternKernel(Kernel):
        pass

    assert resolve_mean_module('NN') == ('NN', {}) and resolve_mean_module(ZeroMean()) == ('zero', {})
    kind, init = resolve_mean_module(ConstantMean(0.25))
    assert kind == 'constant' and init == {'constant_mean': 0.25}
    kind, init, learn = resolve_covar_module(ScaleKernel(RBFKernel([0.5, -1.0]), 0.75))
    assert kind == 'SE' and learn and init == {'lengthscale_raw': [0.5, -1.0], 'outputscale_raw': 0.75}
    kind, init, learn = resolve_covar_module(RBFKernel([0.1]))
    assert kind == 'SE' and not learn and abs(np.log1p(np.exp(init['outputscale_raw'])) - 1.0) < 1e-12
    # the cosine family (round 3; the kernel object of the reference's tests/test_GPR.py:95-101): one raw period in the lengthscale slot
    kind, init, learn = resolve_covar_module(CosineKernel(0.4))
    assert kind == 'COS' and not learn and init['lengthscale_raw'] == [pytest.approx(0.4)]
    kind, init, learn = resolve_covar_module(ScaleKernel(CosineKernel(), -0.5))
    assert kind == 'COS' and learn and init == {'lengthscale_raw': [0.0], 'outputscale_raw': -0.5}
    lay_c = ParamLayout(3, 'zero', 'COS', with_outputscale=True)
    assert lay_c.blocks['lengthscale_raw'] == 1 and lay_c.feature_dim == 3 and lay_c.D == 3 and lay_c.kernel_code == 1
    with pytest.raises(NotImplementedError):
        resolve_covar_module(MaternKernel())
    with pytest.raises(NotImplementedError):
        resolve_mean_module(Mean())
    lay = ParamLayout(2, 'constant', 'SE', with_outputscale=True)
    theta = torch.zeros(lay.D)
    apply_initial_values(theta, lay, {'constant_mean': 0.25, 'lengthscale_raw': [0.5, -1.0], 'outputscale_raw': 0.75})
    assert theta[lay.slices['constant_mean'][0]] == 0.25 and theta[lay.slices['outputscale_raw'][0]] == 0.75
    assert theta[lay.slices['lengthscale_raw'][0]:lay.slices['lengthscale_raw'][1]].tolist() == [0.5, -1.0]


def test_failed_cholesky_on_another_rank_raises_on_every_rank(monkeypatch):
    """a rank whose shard hit a non-positive-definite matrix must not raise alone (the others would wait in the next
    all-reduce).  Its NaN likelihood sums / loss reach every rank through the packed buffer; _check_numerics raises on a rank whose own
    flag is clean when the reduced values are not finite -- and only in multi-rank runs"""
    from types import SimpleNamespace
    from meta_learning_pacoh_amd.engine import NotPSDError
    from meta_learning_pacoh_amd.GPR_meta_mll import GPRegressionMetaLearned
    from meta_learning_pacoh_amd.GPR_meta_svgd import _RandomGPLearner
    clean = lambda: torch.zeros(1, dtype=torch.int32)
    svgd = SimpleNamespace(_fail=clean(), _lik=torch.tensor([0.3, float('nan')]))
    mapl = SimpleNamespace(_fail=clean(), _g_loss=torch.tensor([float('nan')]))
    _RandomGPLearner._check_numerics(svgd)                     # world size 1: the local flag alone decides
    GPRegressionMetaLearned._check_numerics(mapl)
    monkeypatch.setattr(parallel, 'world', lambda: (1, 2))
    with pytest.raises(NotPSDError):
        _RandomGPLearner._check_numerics(svgd)
    with pytest.raises(NotPSDError):
        GPRegressionMetaLearned._check_numerics(mapl)
    ok = SimpleNamespace(_fail=clean(), _lik=torch.tensor([0.3, 0.1]))
    _RandomGPLearner._check_numerics(ok)
    flagged = SimpleNamespace(_fail=torch.ones(1, dtype=torch.int32), _lik=torch.tensor([0.3, 0.1]))
    with pytest.raises(NotPSDError):
        _RandomGPLearner._check_numerics(flagged)
    assert int(flagged._fail) == 0                             # (reset, as before)


def test_host_cpu_budget_is_within_the_visible_cores():
    """util.host_cpu_budget: affinity mask capped by the cgroup quota -- what bench.py and the test session size torch's intra-op
    pool by (256 visible cores behind a quota of 16 got the spinning pool throttled and the GPU starved)"""
    import os
    from meta_learning_pacoh_amd.util import host_cpu_budget
    b = host_cpu_budget()
    assert isinstance(b, int) and 1 <= b <= (os.cpu_count() or 1)
    try:
        q, per = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if q != 'max':
            assert b <= max(1, int(float(q) / float(per)))
    except OSError:
        pass


def test_standard_normal_into_a_buffer_is_the_same_draw():
    """GPR_meta_vi.standard_normal(out=row): the values and the generator state after the call equal those of the plain call (the
    reference's Normal(...).rsample, random_gp.py:244-248) -- PACOH-VI draws its per-step noise straight into pinned staging rows"""
    import torch
    torch.manual_seed(77)
    a = [standard_normal(3, 50) for _ in range(4)]
    state_a = torch.get_rng_state()
    torch.manual_seed(77)
    buf = torch.empty(4, 3, 50)
    for j in range(4):
        standard_normal(3, 50, out=buf[j])
    assert torch.equal(torch.stack(a), buf) and torch.equal(state_a, torch.get_rng_state())
    torch.manual_seed(77)
    buf64 = torch.empty(4, 3, 50, dtype=torch.float64)
    for j in range(4):
        standard_normal(3, 50, out=buf64[j])
    assert torch.equal(torch.stack(a).double(), buf64)
    # ... and they are the reference's own call, torch.distributions' _standard_normal = torch.normal(zeros, ones), value for value and
    # generator state for generator state -- at sizes on both sides of the generator's 16-element blocks and at the launchers' size
    for n, D in ((3, 50), (1, 1), (1, 15), (2, 8), (3, 7), (10, 6566)):
        torch.manual_seed(5)
        ref = [torch.normal(torch.zeros(n, D), torch.ones(n, D)) for _ in range(2)]
        state_ref = torch.get_rng_state()
        torch.manual_seed(5)
        got = [standard_normal(n, D), standard_normal(n, D, out=torch.empty(n, D))]
        assert all(torch.equal(g, r) for g, r in zip(got, ref)) and torch.equal(state_ref, torch.get_rng_state()), (n, D)


def test_first_chunk_sizes():
    """engine.first_chunk: a call with many steps starts with 16 so that the host's chunk preparation hides behind the GPU; shorter
    calls and small feeds are one chunk (round 5: the round-4 rule for the driver's 20-step window is gone -- profiles/r05_prefetch_ab.txt);
    the sizes always fit the request and the feed"""
    from meta_learning_pacoh_amd import engine
    fc = engine.first_chunk
    assert fc(200, 1024) == 16 and fc(64, 1024) == 16 and fc(1000, 128) == 16
    assert fc(20, 1024) == 20 and fc(63, 1024) == 63 and fc(16, 1024) == 16
    assert fc(15, 1024) == 15 and fc(7, 1024) == 7 and fc(1, 1024) == 1
    assert fc(200, 8) == 8 and fc(20, 4) == 4 and fc(20, 2) == 2
    for n in range(1, 300):
        for chunk in (1, 4, 16, 100, 1024):
            k = fc(n, chunk)
            assert 1 <= k <= min(n, chunk)


def test_bench_line_shape_with_both_scaling_legs():
    """bench.py's JSON line at N > 1: `legs` carries the weak AND the strong scaling mode of the one invocation the driver makes,
    each with ms_per_step / value / exchange / world_size_seen / all_reduce_us; the top-level numbers are the --scaling leg's"""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('pacoh_bench', os.path.join(root, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)

    def leg(scaling, ms, evals):
        return {'scaling': scaling, 'ms_per_step': ms, 'value': evals / (ms * 1e-3), 'unit': 'evals/s', 'evals_per_step': evals,
                'host_ms_per_step': 0.01, 'finite': True, 'exchange': 'torch.distributed.all_reduce between two graphs per step',
                'world_size_seen': 2, 'step_mode': {'graph': True}, 'tasks_total': 2048 if scaling == 'weak' else 1024, 'all_reduce_us': 31.5,
                'steady': {'steps': 200, 'ms_per_step': ms * 0.97, 'value': evals / (ms * 0.97e-3)}}
    legs = {'weak': leg('weak', 0.45, 40960), 'strong': leg('strong', 0.25, 20480)}
    pp = {'kernel_sum': 0.4, 'ms_per_step': 0.47, 'kernel_ms': {'mlp_bwd': 0.17}, 'kernel_ms_raw': {'mlp_bwd': 0.18}, 'event_overhead_ms': 0.009}
    out = bench.assemble_line(metric='task-GP LML+grad evals/sec (n_ctx=64, d=4, 20 particles)', value=legs['weak']['value'], world=2, steps=20,
                              warmup=5, ms_per_step=0.45, scaling='weak', dtype='f32', backend='gloo', world_size_seen=2, leg=legs['weak'],
                              legs=legs, host_ms=0.01, step_mode={'graph': True}, config={'workload': 'x'}, roofline=None,
                              kernel_rooflines={}, step_flops=2.6e10, gram=None, pp=pp, others=None, cpu=None)
    json.loads(json.dumps(out))                                  # one serialisable line
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype',
                'data', 'config', 'roofline', 'cpu_baseline', 'legs', 'exchange', 'all_reduce_us', 'world_size_seen', 'schema', 'steady'):
        assert key in out, key
    assert out['n_gpus'] == 2 and out['scaling'] == 'weak' and out['value'] == round(legs['weak']['value'], 1)
    assert set(out['legs']) == {'weak', 'strong'}
    for l in out['legs'].values():
        for key in ('ms_per_step', 'value', 'exchange', 'world_size_seen', 'all_reduce_us'):
            assert key in l, key
    assert out['all_reduce_us'] == 31.5 and out['vs_baseline'] is None
    assert set(out['steady']) == {'steps', 'ms_per_step', 'value'} and out['steady']['steps'] == 200       # the steady-state region behind the timed one
