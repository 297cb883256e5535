"""N>1 path on CPU: world_size-2 gloo.  Each rank evaluates ITS shard of the step's task batch (here with
the CPU oracle standing in for the HIP engine -- tests may use the oracle) and the partial log-likelihoods
and scores are combined by the package's single packed all-reduce; the result must equal the full batch."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import pacoh_oracle as O


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(1)
    from meta_learning_pacoh_amd import parallel
    T, n, d, P = 7, 12, 2, 3
    tasks = O.sinusoid_tasks_nd(T, n, d)
    stats = O.compute_normalization_stats(tasks)
    otasks = [O.prepare_task(x, y, stats, torch.float64) for x, y in tasks]
    cfg = O.GPConfig(d, 'NN', 'NN', mean_nn_layers=(8, 8), kernel_nn_layers=(8, 8))
    pm, ps = O.hyperprior_mean_std(cfg.layout)
    torch.manual_seed(0)
    theta = O.hyperprior_sample(cfg.layout, pm, ps, P).double()
    rds = np.random.RandomState(5)                       # shared seed -> identical global draw on every rank
    idx = rds.randint(0, T, size=T)
    pre = O.meta_pre_factor([n] * T)
    local = parallel.shard(idx)
    assert parallel.world() == (rank, world)
    th = theta.clone().requires_grad_(True)
    lik = torch.zeros(P, dtype=torch.float64)
    if len(local):
        mll = torch.stack([O.vectorized_gp_mll(th, *otasks[i], cfg) for i in local], -1).sum(-1)
        lik = pre * mll
        (score,) = torch.autograd.grad(lik.sum(), th)
    else:
        score = torch.zeros_like(theta)
    lik, score = parallel.all_reduce_sum_(lik.detach(), score)
    # reference: the whole batch on one process
    th2 = theta.clone().requires_grad_(True)
    full = pre * torch.stack([O.vectorized_gp_mll(th2, *otasks[i], cfg) for i in idx], -1).sum(-1)
    (score_full,) = torch.autograd.grad(full.sum(), th2)
    ok = bool(torch.allclose(lik, full.detach(), rtol=1e-12, atol=1e-12)) and \
        bool(torch.allclose(score, score_full, rtol=1e-10, atol=1e-12))
    # PACOH-MAP's operand: grad[1, D] | loss in one buffer, summed in place
    buf = torch.arange(5, dtype=torch.float64) * (rank + 1)
    parallel.all_reduce_buffer_(buf)
    ok = ok and bool(torch.equal(buf, torch.arange(5, dtype=torch.float64) * 3))
    # unseeded learners agree on rank 0's seed; a given seed is kept as rank 0 gave it
    s_none = parallel.broadcast_seed(None)
    s_given = parallel.broadcast_seed(100 + rank)
    box = [None, None]
    dist.all_gather_object(box, (s_none, s_given))
    ok = ok and box[0] == box[1] and s_given == 100 and isinstance(s_none, int)
    # the debug check of the host RNG streams: equal draws pass, different ones are caught
    os.environ['PACOH_CHECK_RANKS'] = '1'
    parallel.check_same_draws([np.arange(3)], [[1.0, 2.0]])
    try:
        parallel.check_same_draws([np.arange(3)], [[1.0, 2.0 + rank]])
        ok = False
    except AssertionError:
        pass
    # one rank cannot use RCCL directly (library / symbol / id): ALL ranks agree on that before anybody enters the blocking
    # communicator creation, and torch.distributed carries the exchange
    import warnings
    from meta_learning_pacoh_amd import _lib as L
    made = []

    def fake_uid():
        if rank == 1:
            raise RuntimeError('librccl not loadable on this rank')
        return b'x' * L.COMM_ID_BYTES
    real_uid, real_comm = L.comm_unique_id, parallel.RcclComm
    L.comm_unique_id, parallel.RcclComm = fake_uid, (lambda **kw: made.append(1))
    os.environ['PACOH_COMM'] = 'rccl'
    parallel._direct = None
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        comm = parallel._direct_comm()
    ok = ok and comm is None and not made and parallel._direct is False
    os.environ.pop('PACOH_COMM')
    L.comm_unique_id, parallel.RcclComm, parallel._direct = real_uid, real_comm, None
    with open(os.path.join(out_dir, 'rank%d.txt' % rank), 'w') as f:
        f.write('ok' if ok else 'mismatch')
    dist.destroy_process_group()


def test_task_sharding_plus_allreduce_equals_full_batch(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert open(os.path.join(str(tmp_path), 'rank%d.txt' % r)).read() == 'ok'
