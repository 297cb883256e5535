"""pacoh_map_persist -- K whole PACOH-MAP iterations per launch in one workgroup (csrc/map_persist.hip; reference loop body:
meta_learn/GPR_meta_mll.py:104-117, models.py:505-519) -- against the four-launch iteration of the same learner on the same task draws,
against the CPU oracle, and at its limits.  The two device paths sum in different orders, so they agree to rounding, not bit for
bit; one parameter is excluded from the comparison on purpose: the kernel network's OUTPUT BIAS has an exactly-zero derivative (a
stationary kernel sees feature differences only), its gradient is rounding noise, and AdamW turns the sign of that noise into
steps of +-lr -- on either path, and on the reference."""
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


def make_tasks(seed, T, ragged, d=2):
    rs = np.random.RandomState(seed)
    tasks = []
    for t in range(T):
        n = 8 + 2 * (t % 3) if ragged else 12
        x = rs.uniform(-3, 3, size=(n, d))
        tasks.append((x, np.sin(x[:, :1]) + 0.3 * x[:, -1:] + 0.05 * rs.randn(n, 1)))
    return tasks


def fit_both(M, monkeypatch, tasks, n_iter, log_period, **kw):
    out = []
    monkeypatch.setenv('PACOH_MAP_TASK_FUSED', '0')             # (the reference run: the four-launch iteration)
    for persist in ('0', '1'):
        monkeypatch.setenv('PACOH_MAP_PERSIST', persist)
        m = M.GPRegressionMetaLearned(tasks, **kw)
        loss = m.meta_fit(verbose=False, n_iter=n_iter, log_period=log_period)
        out.append((m, float(loss)))
    return out


def keep_mask(m):
    keep = torch.ones_like(m.theta, dtype=torch.bool)
    sl = m.layout.slices.get('kernel_nn.out.bias')
    if sl is not None:
        keep[0, sl[0]:sl[1]] = False
    return keep


CFGS = [
    dict(),                                                                         # two 32-wide networks: the compile-time chains
    dict(covar_module='SE', mean_module='NN'),                                      # one network (BASELINE config #2's modules)
    dict(covar_module='NN', mean_module='constant', feature_dim=3),                 # three features: the f <= 4 GP instantiation
    dict(covar_module='SE', mean_module='constant'),                                # no network: GP + hyper-parameters only
    dict(covar_module='SE', mean_module='zero', learning_mode='learn_kernel'),
    dict(mean_nn_layers=(16, 16), kernel_nn_layers=(16, 16)),                       # generic chains: one quad group + remainder
    dict(mean_nn_layers=(20, 12), kernel_nn_layers=(24,)),                          # widths that are no multiple of 16, different depths
    dict(mean_nn_layers=(32, 32, 32), kernel_nn_layers=(32, 32, 32), weight_decay=0.0, task_batch_size=2),   # (three hidden layers: 2 x 12 points fit the LDS plan)
    dict(learning_mode='learn_mean', covar_module='SE'), dict(learning_mode='learn_kernel', mean_module='constant'),   # frozen column ranges
    dict(mean_nn_layers=(32, 32), kernel_nn_layers=(16,)),
]


@pytest.mark.parametrize('cfg', CFGS)
@pytest.mark.parametrize('ragged', [True, False])
def test_persistent_iterations_equal_the_launch_sequence(M, cfg, ragged, monkeypatch):
    """14 iterations as chunks of 1 + 3 + 4 + 4 + 2 (meta_fit's log periods): parameters, both Adam moments, the last loss and the
    logged running loss; weight decay on every group, decaying learning rate, ragged tasks (zero-padded rows) and full ones"""
    tasks = make_tasks(13, 7, ragged)
    kw = dict(task_batch_size=4, lr_params=1e-2, weight_decay=0.05, lr_decay=0.9, random_seed=3)
    kw.update(cfg)
    (m0, l0), (m1, l1) = fit_both(M, monkeypatch, tasks, 14, 4, **kw)
    assert m0._persist is None and m1._persist is not None and m1.opt_step == 14 and m1.lr_scheduler.epoch == 14
    keep = keep_mask(m1)
    assert bool(torch.isfinite(m1.theta).all())
    assert rel(m1.theta[keep], m0.theta[keep]) < 2e-5
    assert rel(m1.exp_avg[keep], m0.exp_avg[keep]) < 2e-5 and rel(m1.exp_avg_sq[keep], m0.exp_avg_sq[keep]) < 2e-5
    assert abs(l1 - l0) < 1e-5 * max(1.0, abs(l0)) and abs(float(m1._g_cum) - float(m0._g_cum)) < 1e-4
    frozen = torch.ones_like(keep)
    for lo, hi in m1.train_segments:
        frozen[0, lo:hi] = False
    assert torch.equal(m1.theta[frozen], m0.theta[frozen]) and float(m1.exp_avg[frozen].abs().sum()) == 0.0     # untrained entries untouched
    mean0, std0 = m0.predict(*tasks[0], tasks[1][0])
    mean1, std1 = m1.predict(*tasks[0], tasks[1][0])
    assert np.allclose(mean0, mean1, rtol=2e-4, atol=2e-5) and np.allclose(std0, std1, rtol=2e-4, atol=2e-5)


def test_persistent_path_matches_the_oracle_on_the_demo(M, monkeypatch):
    """BASELINE config #1 (demo.py:14-26): 50 AdamW iterations against the CPU oracle's MapOracle on the same draws"""
    env = O.SinusoidDataset(np.random.RandomState(26))
    train, test = env.generate_meta_train_data(20, 5), env.generate_meta_test_data(20, 5, 50)
    monkeypatch.setenv('PACOH_MAP_PERSIST', '1')
    model = M.GPRegressionMetaLearned(train, weight_decay=0.2, num_iter_fit=50, random_seed=30)
    orc = O.MapOracle(train, weight_decay=0.2, num_iter_fit=50, random_seed=30)
    log_o = orc.meta_fit(test, log_period=50, n_iter=50)
    model.meta_fit(test, log_period=50, n_iter=50, verbose=False)
    assert model._persist is not None
    ll, rmse, calib = model.eval_datasets(test)
    assert abs(ll - log_o[-1][2]) < 2e-3 and abs(rmse - log_o[-1][3]) < 2e-3 and abs(calib - log_o[-1][4]) < 5e-3
    lay = model.layout
    lo, hi = lay.slices['kernel_nn.fc_2.weight']
    assert rel(model.theta[0, lo:hi], orc.kernel_net[1].weight.reshape(-1)) < 1e-3
    lo, hi = lay.slices['mean_nn.fc_1.bias']
    assert rel(model.theta[0, lo:hi], orc.mean_net[0].bias.reshape(-1)) < 1e-3


@pytest.mark.parametrize('n,d,tb,T', [(1, 1, 1, 3), (5, 1, 5, 20), (16, 3, 4, 9), (17, 4, 3, 7), (32, 1, 2, 40), (32, 4, 2, 8), (3, 2, 16, 16), (4, 1, 16, 20)])
def test_persistent_kernel_at_the_edges_of_its_shape_range(M, n, d, tb, T, monkeypatch):
    """one point per task, one task per iteration, 16 tasks (every wave a GP), 32 points (two 16-row blocks), four input dimensions
    (the f <= 4 instantiation when the kernel has no network), point counts that leave a tile nearly empty.  (What bounds tasks x
    points is the LDS plan: ~68 floats per point, hidden layer and network beside 4 x the parameter image.)"""
    rs = np.random.RandomState(100 * n + tb)
    tasks = []
    for t in range(T):
        x = rs.uniform(-2, 2, size=(n, d))
        tasks.append((x, np.sin(x.sum(1, keepdims=True)) + 0.05 * rs.randn(n, 1)))
    for cfg in (dict(), dict(covar_module='SE', mean_module='NN')):
        kw = dict(task_batch_size=tb, lr_params=5e-3, weight_decay=0.01, random_seed=5)
        kw.update(cfg)
        (m0, l0), (m1, l1) = fit_both(M, monkeypatch, tasks, 6, 3, **kw)
        assert m1._persist is not None, (n, d, tb, cfg)
        keep = keep_mask(m1)
        assert bool(torch.isfinite(m1.theta).all())
        assert rel(m1.theta[keep], m0.theta[keep]) < 5e-5 and abs(l1 - l0) < 2e-5 * max(1.0, abs(l0))


def test_shapes_outside_the_plan_take_the_launch_sequence(M, monkeypatch):
    monkeypatch.setenv('PACOH_MAP_PERSIST', '1')
    cases = [dict(tasks=make_tasks(1, 20, False), task_batch_size=17),                           # more tasks per iteration than waves
             dict(tasks=[(np.random.RandomState(2).randn(40, 1), np.random.RandomState(3).randn(40, 1))] * 4, task_batch_size=2),   # n > 32
             dict(tasks=make_tasks(4, 6, False), task_batch_size=3, mean_nn_layers=(64, 64), kernel_nn_layers=(64, 64)),
             dict(tasks=make_tasks(5, 6, False), task_batch_size=3, optimizer='SGD'),
             dict(tasks=make_tasks(6, 6, False, d=5), task_batch_size=3)]
    for c in cases:
        tasks = c.pop('tasks')
        m = M.GPRegressionMetaLearned(tasks, random_seed=1, **c)
        m.meta_fit(verbose=False, n_iter=3, log_period=2)
        assert m._persist is None and bool(torch.isfinite(m.theta).all())
    from meta_learning_pacoh_amd import _lib as L
    lib = L.load_library()
    h = L._hidden_arr([32, 32])
    assert lib.pacoh_map_persist_supported(5, 1, 5, L.MEAN_VECTOR, h, 2, 1, h, 2, 2, L.F32) == 1
    assert lib.pacoh_map_persist_supported(5, 1, 5, L.MEAN_VECTOR, h, 2, 1, h, 2, 2, L.F64) == 0          # fp32 only
    assert lib.pacoh_map_persist_supported(33, 1, 5, L.MEAN_VECTOR, h, 2, 1, h, 2, 2, L.F32) == 0
    # the entry point itself refuses what the plan refuses (no launch, no partial work)
    t = torch.zeros(8, device='cuda')
    idx = torch.zeros(1, 5, dtype=torch.int64, device='cuda')
    seg = (L.ctypes.c_int32 * 4)(0, 0, 0, 0)
    rc = lib.pacoh_map_persist(t.data_ptr(), t.data_ptr(), t.data_ptr(), 8, t.data_ptr(), t.data_ptr(), None, 33, 1, idx.data_ptr(), 5,
                               t.data_ptr(), 8, 1, L.MEAN_ZERO, -1, h, 0, 0, -1, h, 0, 1, 0, 1, 2, 1e-3, seg, seg, 1, 0.9, 0.999,
                               None, None, None, L.F32, None)
    assert rc == -2                                                                                   # PACOH_ELIMIT


def test_a_failed_cholesky_raises_like_the_launch_sequence(M, monkeypatch):
    """a NaN noise parameter makes every jittered Cholesky fail (info = -1): gpytorch raises NotPSDError inside the loss evaluation,
    meta_fit raises at its next synchronisation -- on both paths, the persistent one through the failure flag of its last wave"""
    from meta_learning_pacoh_amd.engine import NotPSDError
    tasks = O.sinusoid_tasks_nd(5, 8, 1, seed0=50)
    for persist in ('0', '1'):
        monkeypatch.setenv('PACOH_MAP_PERSIST', persist)
        m = M.GPRegressionMetaLearned(tasks, task_batch_size=3, random_seed=1)
        m.theta[0, m.layout.slices['noise_raw'][0]] = float('nan')
        with pytest.raises(NotPSDError):
            m.meta_fit(verbose=False, n_iter=2)
        assert (m._persist is not None) == (persist == '1')


# ---- the task-fused iteration (pacoh_map_task_step: forward + GP + backward of every task in one launch, then the slab reduction) ----
TASK_CFGS = [
    dict(covar_module='SE', mean_module='NN'),                                      # BASELINE config #2's modules
    dict(),                                                                         # two networks
    dict(mean_nn_layers=(16, 16), kernel_nn_layers=(16, 16)),                       # generic chains
    dict(covar_module='NN', mean_module='constant'),                                # constant mean: d_const through the GP's per-task output
    dict(learning_mode='learn_mean', covar_module='SE'),
    dict(mean_nn_layers=(32, 32, 32), kernel_nn_layers=(32, 32, 32)),
]


@pytest.mark.parametrize('cfg', TASK_CFGS)
@pytest.mark.parametrize('shape', [(24, 32, 1, 24), (40, 12, 2, 21), (9, 5, 1, 40), (6, 17, 3, 5)])
@pytest.mark.parametrize('graph', ['0', '1'])
def test_task_fused_iteration_equals_the_four_launch_iteration(M, cfg, shape, graph, monkeypatch):
    """T tasks of n points, a batch of tb per iteration (more than one workgroup's worth: the persistent kernel does not take it):
    one workgroup per task (several tasks per workgroup at n = 5), ragged batches, 14 iterations eagerly and as replayed graphs"""
    T, n, d, tb = shape
    rs = np.random.RandomState(7 * T + n)
    tasks = []
    for t in range(T):
        m = n - (t % 3) if n > 4 else n
        x = rs.uniform(-3, 3, size=(m, d))
        tasks.append((x, np.sin(x[:, :1]) + 0.3 * x[:, -1:] + 0.05 * rs.randn(m, 1)))
    kw = dict(task_batch_size=tb, lr_params=1e-2, weight_decay=0.05, lr_decay=0.9, random_seed=3)
    kw.update(cfg)
    monkeypatch.setenv('PACOH_MAP_PERSIST', '0')
    monkeypatch.setenv('PACOH_GRAPH', graph)
    out = []
    for fused in ('0', '1'):
        monkeypatch.setenv('PACOH_MAP_TASK_FUSED', fused)
        m = M.GPRegressionMetaLearned(tasks, **kw)
        loss = m.meta_fit(verbose=False, n_iter=14, log_period=4)
        assert m._pipelined and (m._task_ws is not None) == (fused == '1') and m.opt_step == 14
        out.append((m, float(loss)))
    (m0, l0), (m1, l1) = out
    keep = keep_mask(m1)
    assert bool(torch.isfinite(m1.theta).all())
    assert rel(m1.theta[keep], m0.theta[keep]) < 2e-5 and rel(m1.exp_avg[keep], m0.exp_avg[keep]) < 2e-5
    assert rel(m1.exp_avg_sq[keep], m0.exp_avg_sq[keep]) < 2e-5
    assert abs(l1 - l0) < 1e-5 * max(1.0, abs(l0)) and abs(float(m1._g_cum) - float(m0._g_cum)) < 1e-4 * max(1.0, abs(float(m0._g_cum)))


def test_task_fused_step_failure_flag_and_limits(M, monkeypatch):
    from meta_learning_pacoh_amd.engine import NotPSDError
    from meta_learning_pacoh_amd import _lib as L
    monkeypatch.setenv('PACOH_MAP_PERSIST', '0')
    tasks = O.sinusoid_tasks_nd(30, 8, 1, seed0=50)
    m = M.GPRegressionMetaLearned(tasks, task_batch_size=20, random_seed=1)
    m.theta[0, m.layout.slices['noise_raw'][0]] = float('nan')
    with pytest.raises(NotPSDError):
        m.meta_fit(verbose=False, n_iter=2)
    assert m._task_ws is not None
    h = L._hidden_arr([32, 32])
    lib = L.load_library()
    assert lib.pacoh_map_task_workspace_bytes(2000, 32, 1, 256, L.MEAN_VECTOR, h, 2, 0, h, 0, 1, 0, L.F32) > 0
    assert lib.pacoh_map_task_workspace_bytes(2000, 33, 1, 256, L.MEAN_VECTOR, h, 2, 0, h, 0, 1, 0, L.F32) == 0       # n > 32
    assert lib.pacoh_map_task_workspace_bytes(2000, 32, 1, 256, L.MEAN_ZERO, h, 0, 0, h, 0, 1, 0, L.F32) == 0         # no network: nothing to fuse
    assert lib.pacoh_map_task_workspace_bytes(2000, 32, 1, 256, L.MEAN_VECTOR, h, 2, 0, h, 0, 1, 0, L.F64) == 0
    # (round 6) more workgroups than are resident at once: the four-launch iteration, unless any_size
    assert lib.pacoh_map_task_workspace_bytes(2000, 32, 1, 4096, L.MEAN_VECTOR, h, 2, 0, h, 0, 1, 0, L.F32) == 0
    assert lib.pacoh_map_task_workspace_bytes(2000, 32, 1, 4096, L.MEAN_VECTOR, h, 2, 0, h, 0, 1, 1, L.F32) > 0


# ---- wide networks at a tiny batch (round 6, csrc/map_wide.hip): the reference's PACOH-MAP launcher runs 2 tasks x 5 points per iteration
#      through two 4 x 128 networks (experiments/meta_GPR_mll_base_exp.py:29-47) ------------------------------------------------------------
WIDE_CFGS = [
    dict(mean_nn_layers=(128,) * 4, kernel_nn_layers=(128,) * 4),                   # the launcher's networks
    dict(mean_nn_layers=(64, 64), kernel_nn_layers=(128, 64, 32)),                  # different depths and widths per network
    dict(covar_module='SE', mean_module='NN', mean_nn_layers=(128, 128)),           # one network, kernel on the raw inputs
    dict(covar_module='NN', mean_module='constant', kernel_nn_layers=(48, 80)),     # constant mean; widths that are multiples of 16 only
    dict(learning_mode='learn_kernel', mean_module='constant', kernel_nn_layers=(64,)),   # one hidden layer; the frozen mean column untouched
]


@pytest.mark.parametrize('cfg', WIDE_CFGS)
@pytest.mark.parametrize('shape', [(20, 5, 1, 2), (8, 8, 2, 2), (6, 16, 1, 1), (7, 3, 4, 5), (9, 10, 3, 1),
                                   (8, 8, 2, 4), (20, 5, 1, 5), (7, 12, 1, 2), (6, 16, 3, 2)])      # ... and two point tiles: 17 .. 32 points
@pytest.mark.parametrize('graph', ['0', '1'])
def test_wide_network_iteration_equals_the_general_sequence(M, cfg, shape, graph, monkeypatch):
    """T tasks of n <= 16 points (ragged), tb per iteration with tb x n <= 32: forward, GP and backward of the whole batch in ONE workgroup that
    streams the weights from theta (map_wide_kernel) + the slab reduction, against the layer-by-layer general sequence (~60 launches)
    on the same draws: parameters, both Adam moments, losses after 12 iterations, eagerly and as replayed graphs"""
    T, n, d, tb = shape
    if cfg.get('covar_module') == 'SE' and d > 4:
        pytest.skip('f = d <= 4')
    rs = np.random.RandomState(5 * T + n)
    tasks = []
    for t in range(T):
        m = n - (t % 3) if n > 4 else n
        x = rs.uniform(-3, 3, size=(m, d))
        tasks.append((x, np.sin(x[:, :1]) + 0.3 * x[:, -1:] + 0.05 * rs.randn(m, 1)))
    kw = dict(task_batch_size=tb, lr_params=2e-3, weight_decay=0.02, lr_decay=0.9, random_seed=3)
    kw.update(cfg)
    monkeypatch.setenv('PACOH_MAP_PERSIST', '0')
    monkeypatch.setenv('PACOH_GRAPH', graph)
    out = []
    for fused in ('0', '1'):
        monkeypatch.setenv('PACOH_MAP_TASK_FUSED', fused)
        m = M.GPRegressionMetaLearned(tasks, **kw)
        loss = m.meta_fit(verbose=False, n_iter=12, log_period=5)
        assert (m._task_ws is not None) == (fused == '1') and m.opt_step == 12
        out.append((m, float(loss)))
    (m0, l0), (m1, l1) = out
    keep = keep_mask(m1)
    assert bool(torch.isfinite(m1.theta).all())
    assert rel(m1.theta[keep], m0.theta[keep]) < 2e-5 and rel(m1.exp_avg[keep], m0.exp_avg[keep]) < 2e-4
    assert rel(m1.exp_avg_sq[keep], m0.exp_avg_sq[keep]) < 2e-4
    assert abs(l1 - l0) < 1e-5 * max(1.0, abs(l0))
    if cfg.get('learning_mode') == 'learn_kernel':       # frozen column ranges: bit for bit the initial values on both paths
        lo, hi = m1.layout.slices['constant_mean']
        assert torch.equal(m1.theta[:, lo:hi], m0.theta[:, lo:hi])


def test_wide_network_iteration_against_the_oracle_at_the_launcher_shape(M, monkeypatch):
    """experiments/meta_GPR_mll_base_exp.py's defaults (20 sinusoid tasks x 5 points, 2 per iteration, 4 x 128 networks): the first 25
    iterations of the oracle's PACOH-MAP on the same task draws (oracle/pacoh_oracle.py: MapOracle, pinned to the recorded demo run)"""
    import bench
    monkeypatch.setenv('PACOH_MAP_PERSIST', '0')
    tasks = bench.sinusoid_tasks(29, 20, 5)
    layers = (128,) * 4
    kw = dict(mean_nn_layers=layers, kernel_nn_layers=layers, task_batch_size=2, weight_decay=0.0, lr_decay=0.98, random_seed=28, lr_params=1e-3)
    m = M.GPRegressionMetaLearned(tasks, **kw)
    m.meta_fit(verbose=False, n_iter=25, log_period=100)
    assert m._task_ws is not None
    orc = O.MapOracle(tasks, num_iter_fit=25, **kw)
    orc.meta_fit(None, log_period=100, n_iter=25)
    lay = m.layout
    for name, ref in (('kernel_nn.fc_2.weight', orc.kernel_net[1].weight), ('kernel_nn.fc_4.weight', orc.kernel_net[3].weight),
                      ('mean_nn.fc_1.bias', orc.mean_net[0].bias), ('mean_nn.fc_3.weight', orc.mean_net[2].weight),
                      ('mean_nn.out.weight', orc.mean_net[4].weight)):
        lo, hi = lay.slices[name]
        assert rel(m.theta[0, lo:hi], ref.detach().reshape(-1)) < 1e-3, name


def test_wide_network_limits(M):
    from meta_learning_pacoh_amd import _lib as L
    lib = L.load_library()
    h128, h32, h40 = L._hidden_arr([128] * 4), L._hidden_arr([32, 32]), L._hidden_arr([40, 40])
    D = 200000
    assert lib.pacoh_map_task_workspace_bytes(D, 5, 1, 2, L.MEAN_VECTOR, h128, 4, 1, h128, 4, 2, 0, L.F32) > 0
    assert lib.pacoh_map_task_workspace_bytes(D, 5, 1, 6, L.MEAN_VECTOR, h128, 4, 1, h128, 4, 2, 0, L.F32) > 0         # 30 points: two tiles
    assert lib.pacoh_map_task_workspace_bytes(D, 5, 1, 7, L.MEAN_VECTOR, h128, 4, 1, h128, 4, 2, 0, L.F32) == 0        # 35 points > two tiles
    assert lib.pacoh_map_task_workspace_bytes(D, 17, 1, 1, L.MEAN_VECTOR, h128, 4, 1, h128, 4, 2, 0, L.F32) == 0       # tasks of more than 16 points
    assert lib.pacoh_map_task_workspace_bytes(D, 5, 1, 2, L.MEAN_VECTOR, h40, 2, 1, h40, 2, 2, 0, L.F32) == 0          # width no multiple of 16
    assert lib.pacoh_map_task_workspace_bytes(D, 5, 1, 2, L.MEAN_VECTOR, h32, 2, 1, h32, 2, 2, 0, L.F32) > 0           # narrow: the LDS-image kernel
    assert lib.pacoh_map_task_workspace_bytes(D, 5, 1, 2, L.MEAN_VECTOR, h128, 4, 1, h128, 4, 2, 0, L.F64) == 0
