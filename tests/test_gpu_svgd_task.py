"""pacoh_svgd_task_step (round 6, csrc/map_task.hip with P parameter rows): the likelihood half of a PACOH-SVGD / PACOH-VI step for
under-filled grids -- forward of both networks, GP LML + gradient and both networks' backward of every (task, parameter row) problem
in ONE launch, then the slab reduction with the step's tail -- against the general launch sequence (networks forward -> GP ->
networks backward -> slab reduction) on the same operands, against the CPU oracle at the reference launchers' own shape
(experiments/meta_GPR_SVGD_base_exp.py:28-49: 2 tasks x 10 particles, 20 points, 4 x 32 networks), and through the learners.
Reference lines: random_gp.py:204-222 (the sum over the batch), random_gp.py:54-89, svgd.py:12-28, GPR_meta_vi.py:216-224.
The two device paths sum in different orders: they agree to rounding, not bit for bit."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import pacoh_oracle as O


@pytest.fixture(scope='module')
def M():
    if not torch.cuda.is_available():
        pytest.skip('needs a HIP device')
    import meta_learning_pacoh_amd as m
    return m


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-300))


def make_tasks(seed, T, n, d, ragged):
    rs = np.random.RandomState(seed)
    tasks = []
    for t in range(T):
        m = n - (t % 3) if (ragged and n > 4) else n
        x = rs.uniform(-3, 3, size=(m, d))
        tasks.append((x, np.sin(x[:, :1]) + 0.3 * x[:, -1:] + 0.05 * rs.randn(m, 1)))
    return tasks


def keep_cols(layout):
    """every column of a parameter row but the kernel network's OUTPUT BIAS: its derivative is exactly zero (a stationary kernel sees
    feature differences only), what the paths return there is rounding noise of different sums"""
    keep = torch.ones(layout.D, dtype=torch.bool)
    sl = layout.slices.get('kernel_nn.out.bias')
    if sl is not None:
        keep[sl[0]:sl[1]] = False
    return keep


CFGS = [
    dict(),                                                                         # two 2 x 32 networks: the compile-time chains
    dict(mean_nn_layers=(32, 32, 32, 32), kernel_nn_layers=(32, 32, 32, 32)),       # the reference launchers' networks
    dict(covar_module='SE', mean_module='NN'),                                      # one network, kernel on the raw inputs
    dict(covar_module='NN', mean_module='constant'),                                # constant mean: d_const through the per-problem output
    dict(covar_module='NN', mean_module='constant', kernel_nn_layers=(16, 16)),     # generic chains
    dict(mean_nn_layers=(20, 12), kernel_nn_layers=(24,)),                          # widths that are no multiple of 16, different depths
]
SHAPES = [(6, 20, 1, 2, 10), (9, 12, 2, 5, 4), (7, 5, 1, 7, 3), (5, 32, 3, 3, 6), (4, 17, 4, 4, 1),     # (T, n, d, tasks per step, rows)
          (6, 8, 2, 3, 2), (5, 1, 1, 2, 2), (6, 7, 4, 4, 3)]      # n <= 8: the one-entry-per-lane GP body (csrc/gp8_body.h), n = 1, f = d = 4


@pytest.mark.parametrize('cfg', CFGS)
@pytest.mark.parametrize('shape', SHAPES)
@pytest.mark.parametrize('ragged', [False, True])
def test_task_fused_score_equals_the_general_sequence(M, cfg, shape, ragged):
    """one likelihood evaluation on the same particles and task batch through both paths: score [P, D], lik [P], the failure flag"""
    from meta_learning_pacoh_amd import _lib as L
    T, n, d, tb, P = shape
    if d > 1 and cfg.get('covar_module') == 'SE' and d > 4:
        pytest.skip('f = d <= 4')
    m = M.GPRegressionMetaLearnedSVGD(make_tasks(11 * T + n, T, n, d, ragged), num_particles=P, task_batch_size=tb, random_seed=5, **cfg)
    idx = torch.from_numpy(np.random.RandomState(3).randint(0, T, size=tb)).to(m.device)
    batch = m.tasks.select(idx)
    D = m.layout.D
    score0, lik0 = torch.zeros(P, D, device=m.device), torch.zeros(P, device=m.device)
    fail0 = torch.zeros(1, dtype=torch.int32, device=m.device)
    m.engine.lml_and_grad(m.particles, batch, weight=1.0, lik_out=lik0, lik_scale=1.0, grad_out=score0, fail_flag=fail0)
    ws = m._setup_task_fused(P, tb)
    assert ws is not None, 'inside the plan'
    score1, lik1 = torch.full((P, D), float('nan'), device=m.device), torch.full((P,), float('nan'), device=m.device)
    fail1 = torch.zeros(1, dtype=torch.int32, device=m.device)
    L.svgd_task_step(m._task_plan, m.particles, batch, m.engine._hypers(m.particles), score1, lik1, 1.0, fail1, ws)
    torch.cuda.synchronize()
    keep = keep_cols(m.layout).to(m.device)
    assert int(fail0) == 0 and int(fail1) == 0
    assert bool(torch.isfinite(score1).all()) and bool(torch.isfinite(lik1).all())
    assert rel(lik1, lik0) < 2e-5
    assert rel(score1[:, keep], score0[:, keep]) < 5e-5
    for p in range(P):                                    # every row on its own (a swapped row would hide in the norm of all)
        assert rel(score1[p, keep], score0[p, keep]) < 2e-4, p


@pytest.mark.parametrize('n', [2, 5, 8])
def test_small_context_body_takes_the_jitter_ladder(M, n):
    """gp8_body (contexts of <= 8 points: Gauss-Jordan sweeps, one matrix entry per lane): two identical points under a noise of 1e-9
    make the second pivot vanish in fp32 -- rung 1 of gpytorch's ladder (jitter 1e-6) succeeds; the general sequence's kernels climb the
    same ladder on the same problem, and a noise of -1 fails every rung on both (flag raised, NaN sums).  PACOH_GP8=0 (the 16 x 16-block
    body inside the same kernel) must agree too"""
    import os
    from meta_learning_pacoh_amd import _lib as L
    tasks = make_tasks(3, 4, n, 1, False)
    for t in range(4):
        tasks[t][0][1 % n] = tasks[t][0][0]                 # a duplicated point (n = 1: nothing to duplicate, the ladder is not needed)
    m = M.GPRegressionMetaLearnedSVGD(tasks, num_particles=3, task_batch_size=2, random_seed=5, covar_module='SE', mean_module='NN')
    batch = m.tasks.select(torch.tensor([0, 2], device=m.device))
    D, P = m.layout.D, 3
    ls = torch.ones(P, 1, device=m.device)
    res = {}
    for tag, nz in (('ladder', 1e-9), ('fail', -1.0)):
        noise = torch.full((P,), nz, device=m.device)
        score0, lik0 = torch.zeros(P, D, device=m.device), torch.zeros(P, device=m.device)
        fail0 = torch.zeros(1, dtype=torch.int32, device=m.device)
        m.engine.lml_and_grad(m.particles, batch, weight=1.0, lik_out=lik0, lik_scale=1.0, grad_out=score0, fail_flag=fail0, hypers=(ls, None, noise))
        out = []
        for gp8 in ('1', '0'):
            os.environ['PACOH_GP8'] = gp8
            L.reload_env()
            try:
                ws = m._setup_task_fused(P, 2)
                score1, lik1 = torch.zeros(P, D, device=m.device), torch.zeros(P, device=m.device)
                fail1 = torch.zeros(1, dtype=torch.int32, device=m.device)
                L.svgd_task_step(m._task_plan, m.particles, batch, (ls, None, noise), score1, lik1, 1.0, fail1, ws)
                torch.cuda.synchronize()
            finally:
                os.environ.pop('PACOH_GP8', None)
                L.reload_env()
            out.append((score1, lik1, int(fail1)))
        res[tag] = (lik0, int(fail0), out)
    lik0, fail0, out = res['ladder']
    assert fail0 == 0 and bool(torch.isfinite(lik0).all())
    for score1, lik1, fail1 in out:
        assert fail1 == 0 and bool(torch.isfinite(lik1).all()) and bool(torch.isfinite(score1).all())
        assert rel(lik1, lik0) < (1e-2 if n > 1 else 1e-5)     # (condition number ~1e6 after the jitter: the factorisation is the error)
    lik0, fail0, out = res['fail']
    assert fail0 == 1
    for score1, lik1, fail1 in out:
        assert fail1 == 1 and bool(torch.isnan(lik1).all())


def test_task_fused_score_against_the_oracle_at_the_launcher_shape(M):
    """experiments/meta_GPR_SVGD_base_exp.py's defaults: 20 sinusoid tasks x 20 points, 2 per step, 10 particles, 4 x 32 networks:
    score and likelihood of one step against the oracle's fp64 autograd (oracle/pacoh_oracle.py: meta_log_prob without the prior)"""
    from meta_learning_pacoh_amd import _lib as L
    import bench
    tasks = bench.sinusoid_tasks(29, 20, 20)
    layers = (32, 32, 32, 32)
    m = M.GPRegressionMetaLearnedSVGD(tasks, num_particles=10, task_batch_size=2, random_seed=28, mean_nn_layers=layers,
                                      kernel_nn_layers=layers, prior_factor=0.1, bandwidth=0.1)
    idx = [3, 11]
    batch = m.tasks.select(torch.tensor(idx, device=m.device))
    ws = m._setup_task_fused(10, 2)
    assert ws is not None
    D = m.layout.D
    score, lik = torch.empty(10, D, device=m.device), torch.empty(10, device=m.device)
    fail = torch.zeros(1, dtype=torch.int32, device=m.device)
    L.svgd_task_step(m._task_plan, m.particles, batch, m.engine._hypers(m.particles), score, lik, 1.0, fail, ws)
    gcfg = O.GPConfig(1, 'NN', 'NN', layers, layers)
    assert gcfg.D == D
    stats = O.compute_normalization_stats(tasks)
    th = m.particles.double().cpu().requires_grad_(True)
    mll = 0.0
    for t in idx:
        x, y = O.prepare_task(tasks[t][0], tasks[t][1], stats, torch.float64)
        mll = mll + O.vectorized_gp_mll(th, x, y, gcfg)
    (ref_score,) = torch.autograd.grad(mll.sum(), th)
    keep = keep_cols(m.layout)
    assert int(fail) == 0
    assert rel(lik, mll.detach()) < 1e-5
    assert rel(score.cpu()[:, keep], ref_score[:, keep]) < 1e-3           # fp32 bar of the north star: 1e-2


@pytest.mark.parametrize('cfg', [dict(), dict(mean_nn_layers=(32, 32, 32, 32), kernel_nn_layers=(32, 32, 32, 32), bandwidth=0.1),
                                 dict(covar_module='SE'), dict(mean_module='constant', optimizer='SGD'), dict(kernel='IMQ')])
@pytest.mark.parametrize('graph', ['0', '1'])
def test_svgd_learner_on_the_task_fused_step(M, cfg, graph, monkeypatch):
    """meta_fit with the task-fused likelihood launch against the general sequence on the same draws: particles and optimizer state
    after 12 steps, eagerly and as replayed graphs (median and fixed bandwidth, Adam and SGD, the IMQ particle kernel's launch order)"""
    monkeypatch.setenv('PACOH_GRAPH', graph)
    tasks = make_tasks(4, 8, 20, 1, ragged=True)
    out = []
    for fused in ('0', '1'):
        monkeypatch.setenv('PACOH_SVGD_TASK_FUSED', fused)
        m = M.GPRegressionMetaLearnedSVGD(tasks, num_particles=5, task_batch_size=2, lr=5e-3, lr_decay=0.9, random_seed=9, **cfg)
        m.meta_fit(verbose=False, n_iter=12, log_period=5)
        assert (m._task_ws is not None) == (fused == '1') and m.opt_step == 12
        out.append(m)
    m0, m1 = out
    keep = keep_cols(m1.layout).to(m1.device)
    assert bool(torch.isfinite(m1.particles).all())
    assert rel(m1.particles[:, keep], m0.particles[:, keep]) < 5e-5
    if cfg.get('optimizer') != 'SGD':
        assert rel(m1.exp_avg[:, keep], m0.exp_avg[:, keep]) < 2e-3 and rel(m1.exp_avg_sq[:, keep], m0.exp_avg_sq[:, keep]) < 2e-3
    mu0, sd0 = m0.predict(tasks[0][0], tasks[0][1], tasks[1][0])
    mu1, sd1 = m1.predict(tasks[0][0], tasks[0][1], tasks[1][0])
    assert np.allclose(mu1, mu0, rtol=1e-3, atol=1e-3) and np.allclose(sd1, sd0, rtol=1e-3, atol=1e-3)


def test_svgd_task_fused_replay_equals_eager_bit_for_bit(M, monkeypatch):
    """the same launches issued one by one and replayed from the captured graphs give the same bits"""
    monkeypatch.setenv('PACOH_SVGD_TASK_FUSED', '1')
    tasks = make_tasks(6, 10, 20, 1, ragged=False)
    res = []
    for graph in ('0', '1'):
        monkeypatch.setenv('PACOH_GRAPH', graph)
        m = M.GPRegressionMetaLearnedSVGD(tasks, num_particles=10, task_batch_size=2, random_seed=2, mean_nn_layers=(32,) * 4,
                                          kernel_nn_layers=(32,) * 4, bandwidth=0.1, prior_factor=0.1)
        m.meta_fit(verbose=False, n_iter=21, log_period=10)
        assert m._task_ws is not None
        res.append(m.particles.clone())
    assert torch.equal(res[0], res[1])


@pytest.mark.parametrize('layers', [(32, 32), (32, 32, 32, 32)])
def test_workgroup_size_of_the_task_kernel_does_not_change_the_bits(M, layers, monkeypatch):
    """8 waves per workgroup or 16 (the launch picks 16 where the weight tiles would take 8 waves more than two rounds: two 4 x 32
    networks; PACOH_MT_NT forces one): the tiles land on other waves, every tile's arithmetic and every slab entry stay what they were"""
    from meta_learning_pacoh_amd import _lib as L
    monkeypatch.setenv('PACOH_SVGD_TASK_FUSED', '1')
    monkeypatch.setenv('PACOH_GRAPH', '0')
    tasks = make_tasks(8, 10, 20, 1, ragged=True)
    res = []
    try:
        for nt in ('512', '1024', '0'):
            monkeypatch.setenv('PACOH_MT_NT', nt)
            L.reload_env()
            m = M.GPRegressionMetaLearnedSVGD(tasks, num_particles=6, task_batch_size=2, random_seed=4, mean_nn_layers=layers,
                                              kernel_nn_layers=layers, bandwidth=0.1, prior_factor=0.1)
            m.meta_fit(verbose=False, n_iter=9, log_period=10)
            assert m._task_ws is not None
            res.append(m.particles.clone())
    finally:
        monkeypatch.delenv('PACOH_MT_NT')
        L.reload_env()
    assert torch.equal(res[0], res[1]) and torch.equal(res[0], res[2])


@pytest.mark.parametrize('cfg', [dict(), dict(mean_nn_layers=(32, 32, 32, 32), kernel_nn_layers=(32, 32, 32, 32)),
                                 dict(covar_module='SE', mean_module='NN')])
@pytest.mark.parametrize('graph', ['0', '1'])
def test_vi_learner_on_the_task_fused_step(M, cfg, graph, monkeypatch):
    """PACOH-VI (diagonal posterior): S posterior samples are the kernel's parameter rows"""
    monkeypatch.setenv('PACOH_GRAPH', graph)
    tasks = make_tasks(8, 8, 20, 1, ragged=True)
    out = []
    for fused in ('0', '1'):
        monkeypatch.setenv('PACOH_SVGD_TASK_FUSED', fused)
        m = M.GPRegressionMetaLearnedVI(tasks, svi_batch_size=6, task_batch_size=2, lr=5e-3, random_seed=13, **cfg)
        m.meta_fit(verbose=False, n_iter=10, log_period=4)
        assert (m._task_ws is not None) == (fused == '1')
        out.append(m)
    m0, m1 = out
    keep = keep_cols(m1.layout).to(m1.device)
    assert bool(torch.isfinite(m1.posterior).all())
    assert rel(m1.posterior[:, keep], m0.posterior[:, keep]) < 5e-5
    assert abs(float(m1._loss) - float(m0._loss)) < 1e-4 * max(1.0, abs(float(m0._loss)))


def test_task_fused_step_limits_and_failure_flag(M, monkeypatch):
    from meta_learning_pacoh_amd.engine import NotPSDError
    from meta_learning_pacoh_amd import _lib as L
    h = L._hidden_arr([32, 32])
    lib = L.load_library()
    assert lib.pacoh_svgd_task_workspace_bytes(2534, 10, 20, 1, 2, L.MEAN_VECTOR, h, 2, 1, h, 2, 2, 0, L.F32) > 0
    assert lib.pacoh_svgd_task_workspace_bytes(2534, 10, 33, 1, 2, L.MEAN_VECTOR, h, 2, 1, h, 2, 2, 0, L.F32) == 0        # n > 32
    assert lib.pacoh_svgd_task_workspace_bytes(2534, 10, 20, 1, 2, L.MEAN_ZERO, h, 0, 0, h, 0, 1, 0, L.F32) == 0          # no network: nothing to fuse
    assert lib.pacoh_svgd_task_workspace_bytes(2534, 10, 20, 1, 2, L.MEAN_VECTOR, h, 2, 1, h, 2, 2, 0, L.F64) == 0        # fp32 only
    assert lib.pacoh_svgd_task_workspace_bytes(2534, 10, 20, 1, 2, L.MEAN_VECTOR, h, 2, 1, h, 2, L._kf(2, L.KERNEL_COSINE), 0, L.F32) == 0
    # more workgroups than are resident at once (a second round of ~20 us latency chains): the throughput kernels -- unless any_size
    assert lib.pacoh_svgd_task_workspace_bytes(2534, 10, 20, 1, 400, L.MEAN_VECTOR, h, 2, 1, h, 2, 2, 0, L.F32) == 0
    assert lib.pacoh_svgd_task_workspace_bytes(2534, 10, 20, 1, 400, L.MEAN_VECTOR, h, 2, 1, h, 2, 2, 1, L.F32) > 0
    m = M.GPRegressionMetaLearnedSVGD(make_tasks(1, 300, 12, 1, False), num_particles=10, task_batch_size=200, random_seed=1)
    m.meta_fit(verbose=False, n_iter=2)
    assert m._task_ws is None
    # n = 64 is outside the plan: the general sequence
    monkeypatch.setenv('PACOH_SVGD_TASK_FUSED', '1')
    m = M.GPRegressionMetaLearnedSVGD(make_tasks(1, 6, 64, 1, False), num_particles=4, task_batch_size=2, random_seed=1)
    m.meta_fit(verbose=False, n_iter=2)
    assert m._task_ws is None and bool(torch.isfinite(m.particles).all())
    # a Cholesky that fails even with the jitter ladder raises at the next synchronisation, as on the general path
    m = M.GPRegressionMetaLearnedSVGD(make_tasks(2, 6, 20, 1, False), num_particles=4, task_batch_size=2, random_seed=1)
    m.particles[2, m.layout.slices['noise_raw'][0]] = float('nan')
    with pytest.raises(NotPSDError):
        m.meta_fit(verbose=False, n_iter=2)
    assert m._task_ws is not None
