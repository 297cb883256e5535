"""N > 1 path on REAL kernels: two processes share the one MI355X of the test box (gloo transport with HIP tensors -- RCCL refuses
two ranks on one device), each runs the SVGD learner on its shard of every step's task batch, and after three steps both ranks
must hold the particles of the single-process run.  Only the transport differs from the multi-GPU job (`nccl` = RCCL there)."""
import os
import socket
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

from oracle import pacoh_oracle as O


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _run(world_rank, world, port, out_dir, kind):
    if world > 1:
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(world_rank), WORLD_SIZE=str(world))
        dist.init_process_group('gloo', rank=world_rank, world_size=world)
    torch.cuda.set_device(0)
    import meta_learning_pacoh_amd as M
    tasks = O.sinusoid_tasks_nd(9, 16, 2, seed0=300)
    if kind == 'svgd_imq':
        # the IMQ particle kernel's update (round 4: on the step feed / in the step graphs) behind the same exchange
        model = M.GPRegressionMetaLearnedSVGD(tasks, num_particles=4, task_batch_size=6, lr=1e-2, random_seed=11, kernel='IMQ', optimizer='SGD')
        model.meta_fit(verbose=False, n_iter=3)
        state = model.particles
    elif kind in ('svgd', 'svgd_unseeded'):
        # unseeded: rank 0's seed is broadcast (parallel.broadcast_seed), so the ranks still draw the same task batches
        model = M.GPRegressionMetaLearnedSVGD(tasks, num_particles=4, task_batch_size=6, lr=1e-2,
                                              random_seed=11 if kind == 'svgd' else None)
        model.meta_fit(verbose=False, n_iter=3)
        state = model.particles
    elif kind in ('map', 'map_sgd'):
        # PACOH-MAP shards its task batch too: grad[1,D] + loss all-reduced between the two graphs of a step.
        # 'map_sgd': the update is linear in the gradient, so the sharded sum can be compared tightly with the single process;
        # 'map': AdamW (the graphed path) normalises near-zero gradient entries to +-lr steps, which amplifies re-association noise
        model = M.GPRegressionMetaLearned(tasks, task_batch_size=5, lr_params=1e-2, weight_decay=0.1, random_seed=11,
                                          optimizer='Adam' if kind == 'map' else 'SGD')
        model.meta_fit(verbose=False, n_iter=4)
        state = model.theta.clone()
        # one parameter is left out of the comparison: the kernel network's OUTPUT BIAS has an exactly-zero derivative (a stationary
        # kernel sees feature differences only), its gradient is rounding noise and AdamW turns the sign of that noise into steps of
        # +-lr on every path, the reference's included (tests/test_gpu_map_persist.py: keep_mask)
        lo, hi = model.layout.slices['kernel_nn.out.bias']
        state[0, lo:hi] = 0.0
    else:
        model = M.GPRegressionMetaLearnedVI(tasks, svi_batch_size=3, task_batch_size=6, lr=1e-2, random_seed=11,
                                            mean_module='constant', covar_module='SE')
        model.meta_fit(verbose=False, n_iter=3)
        state = model.posterior
    torch.cuda.synchronize()
    np.save(os.path.join(out_dir, '%s_w%d_r%d.npy' % (kind, world, world_rank)), state.cpu().numpy())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_unseeded_two_rank_run_stays_in_step():
    """random_seed=None with two ranks: without the seed broadcast each rank would sample its own task batches (and VI its own
    noise) and the all-reduce would mix gradients of different objectives; with it both ranks end bit-identical"""
    if not torch.cuda.is_available():
        pytest.skip('needs a HIP device')
    with tempfile.TemporaryDirectory() as out:
        ctx = mp.get_context('spawn')
        port = _free_port()
        procs = [ctx.Process(target=_run, args=(r, 2, port, out, 'svgd_unseeded')) for r in range(2)]
        for q in procs:
            q.start()
        for q in procs:
            q.join(600)
            assert q.exitcode == 0
        r0 = np.load(os.path.join(out, 'svgd_unseeded_w2_r0.npy'))
        r1 = np.load(os.path.join(out, 'svgd_unseeded_w2_r1.npy'))
        assert np.isfinite(r0).all() and np.array_equal(r0, r1)


@pytest.mark.parametrize('kind', ['svgd', 'svgd_imq', 'vi', 'map', 'map_sgd'])
def test_two_ranks_on_one_gpu_match_single_process(kind):
    if not torch.cuda.is_available():
        pytest.skip('needs a HIP device')
    with tempfile.TemporaryDirectory() as out:
        ctx = mp.get_context('spawn')
        p = ctx.Process(target=_run, args=(0, 1, 0, out, kind))
        p.start(); p.join(600)
        assert p.exitcode == 0
        port = _free_port()
        procs = [ctx.Process(target=_run, args=(r, 2, port, out, kind)) for r in range(2)]
        for q in procs:
            q.start()
        for q in procs:
            q.join(600)
            assert q.exitcode == 0
        ref = np.load(os.path.join(out, '%s_w1_r0.npy' % kind))
        r0 = np.load(os.path.join(out, '%s_w2_r0.npy' % kind))
        r1 = np.load(os.path.join(out, '%s_w2_r1.npy' % kind))
        assert np.array_equal(r0, r1)                                  # replicas stay bit-identical
        assert np.isfinite(ref).all()
        # sharding only changes the order of the sum over tasks (fp32 re-association) -- for AdamW too, once the one parameter whose
        # gradient is pure rounding noise is left out (see _run)
        assert np.abs(r0 - ref).max() < 2e-3 * max(1.0, np.abs(ref).max())
        assert np.linalg.norm(r0 - ref) < 1e-3 * np.linalg.norm(ref)


def test_bench_two_ranks_prints_both_scaling_legs():
    """`torch.distributed.run --nproc-per-node 2 bench.py --gpus 2` (gloo transport: two ranks on the box's one GPU) prints ONE JSON
    line whose `legs` holds the weak and the strong scaling mode, each timed with the exchange in place"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PACOH_BENCH_BACKEND='gloo')
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                        '--master-port', str(_free_port()), os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '4', '--warmup', '2'],
                       cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['world_size_seen'] == 2 and out['scaling'] == 'weak'
    assert set(out['legs']) == {'weak', 'strong'}
    assert out['legs']['weak']['tasks_total'] == 2048 and out['legs']['strong']['tasks_total'] == 1024
    for leg in out['legs'].values():
        assert leg['finite'] and leg['ms_per_step'] > 0 and leg['world_size_seen'] == 2
        assert 'torch.distributed' in leg['exchange'] and leg['all_reduce_us'] is not None and leg['all_reduce_us'] > 0
    assert out['value'] == out['legs']['weak']['value']
    assert out['legs']['strong']['ideal_ms_per_step'] == round(out['legs']['weak']['ms_per_step'] / 2, 4) and out['steady']['steps'] == 200
