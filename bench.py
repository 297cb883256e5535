#!/usr/bin/env python
"""Benchmark of the PACOH task-GP hot path on MI355X.

Workload (BASELINE.json configs[2], the one the metric is quoted on): PACOH-SVGD meta-training,
1024 tasks per GPU, n_ctx = 64, d = 4, 20 particles, NN(32,32) mean + NN(32,32) kernel features
(D = 2534 prior parameters per particle), fp32.  One "step" = one full svgd_step over every
(task, particle) pair held by the job: per-particle MLP features -> fused Gram/Cholesky/solve/log-det
LML and its gradient -> MLP backward -> hyper-prior gradient -> SVGD phi -> Adam.  One "eval" = one
(task, particle) LML + gradient.  Weak scaling: every rank processes 1024 tasks x 20 particles, the
score [20 x 2534] is summed with one RCCL all-reduce per step.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus 8 --steps 20 --warmup 5

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

TASKS_PER_GPU, N_CTX, DIM, PARTICLES = 1024, 64, 4, 20
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy peak)
FP32_PEAK_TFLOPS = 157.3       # fp32 vector peak == fp32-input MFMA peak


def make_tasks(n_tasks, n, d, seed0=1000):
    """synthetic d-dimensional sinusoid-of-mean tasks (parameters as
    experiments/data_sim.py:242-248), per task RandomState(seed0 + t)"""
    tasks = []
    for t in range(n_tasks):
        rs = np.random.RandomState(seed0 + t)
        X = rs.uniform(-5, 5, size=(n, d))
        amp, x_shift = rs.uniform(0.7, 1.3), rs.normal(0.0, 0.1)
        y_shift, slope = rs.normal(5.0, 0.1), rs.normal(0.5, 0.2)
        xm = X.mean(axis=1, keepdims=True)
        Y = slope * xm + amp * np.sin(1.5 * (xm - x_shift)) + y_shift + 0.1 * rs.normal(size=(n, 1))
        tasks.append((X, Y))
    return tasks


def gp_flops_per_eval(n, f, w_nn):
    """SURVEY.md 8(d) flop model: F = 6 n W_nn + n^2(3f+2) + n^3/3 + 2n^2 + [bwd] 2n^3/3 + 2n^2 + 4 n^2 f"""
    gp = n * n * (3 * f + 2) + n ** 3 / 3 + 2 * n * n + 2 * n ** 3 / 3 + 2 * n * n + 4 * n * n * f
    return 6 * n * w_nn + gp, gp


def cpu_baseline(budget_s=12.0):
    """The CPU oracle (plain torch restatement of the reference's arithmetic, oracle/pacoh_oracle.py) on
    the host cores, on a bounded sample of the same workload.  Reported baseline only."""
    from oracle import pacoh_oracle as O
    cores = os.cpu_count() or 1
    T_s = 32
    tasks = make_tasks(T_s, N_CTX, DIM)
    stats = O.compute_normalization_stats(tasks)
    otasks = [O.prepare_task(x, y, stats, torch.float32) for x, y in tasks]
    cfg = O.GPConfig(DIM, 'NN', 'NN')
    pm, ps = O.hyperprior_mean_std(cfg.layout, 0.5, 3.0)
    torch.manual_seed(0)
    theta = O.hyperprior_sample(cfg.layout, pm, ps, PARTICLES)
    def rate(loop, threads, budget):
        torch.set_num_threads(threads)
        O.meta_score(theta, otasks, cfg, pm, ps, 0.01, loop=loop)          # warm-up
        t0, reps = time.time(), 0
        while time.time() - t0 < budget:
            O.meta_score(theta, otasks, cfg, pm, ps, 0.01, loop=loop)
            reps += 1
        return T_s * PARTICLES * reps / (time.time() - t0)

    # small batched LAPACK/BLAS calls do not scale to every core of a big host: probe a few thread counts
    cands = sorted({1, min(8, cores), min(32, cores), cores})
    probe = {th: rate(False, th, budget_s / 8) for th in cands}
    best = max(probe, key=probe.get)
    batched = rate(False, best, budget_s / 4)
    looped = rate(True, best, budget_s / 4)
    torch.set_num_threads(cores)
    return {'value': round(batched, 1), 'unit': 'evals/s', 'cores': best, 'kind': 'port',
            'sample': '%d tasks x %d particles (n=%d, d=%d, NN/NN, fp32) LML+autograd score with the CPU oracle, fully '
                      'batched over tasks x particles, best of torch threads %s on a %d-core host (%s evals/s); '
                      'reference-style python loop over tasks at %d threads: %.1f evals/s'
                      % (T_s, PARTICLES, N_CTX, DIM, cands, cores, {k: round(v) for k, v in probe.items()}, best, looped)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)     # 0.15 s of GPU time; 20 steps still carry ~4 % of start-up per step
    ap.add_argument('--warmup', type=int, default=10)     # (the first step after a synchronise waits for the host to issue it)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    assert world == args.gpus, '--gpus must equal WORLD_SIZE (launch N>1 with torch.distributed.run)'
    # PACOH_BENCH_BACKEND=gloo lets the N > 1 code path be exercised on a box with fewer GPUs than ranks (ranks then share
    # devices; RCCL itself refuses two ranks per device).  The measured job always uses 'nccl' = RCCL over xGMI.
    backend = os.environ.get('PACOH_BENCH_BACKEND', 'nccl')
    dev_index = local_rank if backend == 'nccl' else local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_index)
    if world > 1:
        import torch.distributed as dist
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', dev_index))
        else:
            dist.init_process_group(backend)

    import meta_learning_pacoh_amd as M
    from meta_learning_pacoh_amd import _lib as L

    T_global = TASKS_PER_GPU * world
    tasks = make_tasks(T_global, N_CTX, DIM)
    model = M.GPRegressionMetaLearnedSVGD(tasks, num_particles=PARTICLES, covar_module='NN', mean_module='NN',
                                          task_batch_size=-1, lr=1e-3, random_seed=0)
    D = model.layout.D

    def step():
        idx_local, pre = model._sample_task_batch()
        model.svgd_step(idx_local, pre)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device='cuda')
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    finite = bool(torch.isfinite(model.particles).all())

    # ---- per-kernel breakdown with HIP events on the launch stream (separate, instrumented steps) ----
    L.PROFILE = {}
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    prof = L.profile_summary()
    L.PROFILE = None
    kernel_ms = {k: round(v[1] / args.steps, 4) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])}
    dom = max(prof.items(), key=lambda kv: kv[1][1])[0]
    evals_per_gpu = TASKS_PER_GPU * PARTICLES
    w_nn = 2 * (DIM * 32 + 32 * 32) + 32 * 1 + 32 * 2
    f_total, f_gp = gp_flops_per_eval(N_CTX, 2, w_nn)               # W_nn = 2400 MAC/point over both nets
    pmc_kernels = {}                    # HBM bytes per launch of each kernel, from the committed PMC profile
    try:
        with open(os.path.join(ROOT, 'profiles', 'r01_pmc_hbm_traffic.json')) as fh:
            pmc_kernels = json.load(fh)['kernels']
    except Exception:
        pass

    def kernel_roofline(name):
        """fp32-peak roofline of one fused kernel from its HIP-event time in this run (flop models: SURVEY 8d)"""
        launches, tot_ms = prof[name]
        per_launch_s = tot_ms / launches * 1e-3
        key = {'mlp_bwd': 'mlp_mfma_bwd', 'gp_lml_fwdbwd': 'gp_mfma_kernel', 'mlp_fwd': 'mlp_mfma_fwd'}.get(name, name)
        cands = [v['hbm_bytes_per_launch'] for k, v in pmc_kernels.items() if key in k]
        traffic = cands[0] if cands else None
        if name in ('gp_lml_fwdbwd', 'meta_lml_grad'):
            flops = (f_gp if name == 'gp_lml_fwdbwd' else f_total) * evals_per_gpu
            note = ('fp32 VALU/LDS kernel priced against the fp32 peak (vector == f32 MFMA rate); '
                    'algorithmic flops per eval = %.0f (SURVEY 8d model)' % (flops / evals_per_gpu))
        else:
            # MLP kernels: algorithmic flops 2*n*W per eval forward, 4*n*W backward (+ recompute)
            mult = 2 if name == 'mlp_fwd' else 6
            flops = mult * N_CTX * (w_nn / 2) * evals_per_gpu
            note = ('fp32 MFMA + VALU kernel (registers/LDS only between HBM in/out), priced against the fp32 peak with '
                    'the 2*n*W (fwd) / 6*n*W (bwd, incl. recompute) flop model, W = %d MAC per point' % (w_nn // 2))
        return {'kernel': name, 'bound': 'mfma', 'achieved': round(flops / per_launch_s / 1e12, 4), 'peak': FP32_PEAK_TFLOPS,
                'unit': 'TFLOP/s', 'frac': round(flops / per_launch_s / 1e12 / FP32_PEAK_TFLOPS, 5), 'traffic': traffic, 'note': note}

    roofline = kernel_roofline(dom)
    # the two heaviest kernels trade places from run to run (0.30 ms each): report every fused kernel's fraction as well
    kernel_rooflines = {k: {'achieved': r['achieved'], 'frac': r['frac'], 'unit': 'TFLOP/s'}
                        for k, r in ((k, kernel_roofline(k)) for k in ('gp_lml_fwdbwd', 'mlp_bwd', 'mlp_fwd') if k in prof)}

    # ---- standalone Gram build (the HBM-write-bound kernel): same problem count, materialised K ----
    gram = None
    if rank == 0:
        B = evals_per_gpu
        z = torch.randn(B, N_CTX, 2, device='cuda')
        ls = torch.rand(PARTICLES, 2, device='cuda') + 0.5
        for _ in range(10):
            L.gram_rbf_ard(z, 1, z, 1, ls, None, None, False, B, PARTICLES)
        torch.cuda.synchronize()
        reps = 100
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        K = torch.empty(B, N_CTX, N_CTX, device='cuda')
        lib = L.load_library()
        s.record()
        for _ in range(reps):
            lib.pacoh_gram_rbf_ard(z.data_ptr(), 1, z.data_ptr(), 1, ls.data_ptr(), None, None, 0, K.data_ptr(),
                                   B, PARTICLES, N_CTX, N_CTX, 2, 0, torch.cuda.current_stream().cuda_stream)
        e.record()
        torch.cuda.synchronize()
        t_k = s.elapsed_time(e) / reps * 1e-3
        alg_bytes = B * (N_CTX * 2 * 4 + N_CTX * N_CTX * 4)
        traffic = None                      # HBM bytes per launch from the committed PMC profile of this same launch shape
        try:
            with open(os.path.join(ROOT, 'profiles', 'r01_pmc_hbm_traffic.json')) as fh:
                kern = json.load(fh)['kernels']
                key = [k for k in kern if k.startswith('gram_kernel<float, 2')][0]
                traffic = kern[key]['hbm_bytes_per_launch']
        except Exception:
            pass
        gram = {'kernel': 'gram_rbf_ard', 'bound': 'hbm', 'achieved': round(alg_bytes / t_k / 1e9, 1),
                'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(alg_bytes / t_k / 1e9 / HBM_PEAK_GBS, 4),
                'traffic': traffic, 'algorithmic_bytes': alg_bytes, 'bytes_per_gram': N_CTX * 2 * 4 + N_CTX * N_CTX * 4, 'grams': B,
                'us_per_launch': round(t_k * 1e6, 2)}

    if rank == 0:
        cpu = None if args.no_cpu_baseline or world > 1 else cpu_baseline()
        value = T_global * PARTICLES * args.steps / elapsed
        out = {
            'metric': 'task-GP LML+grad evals/sec (n_ctx=64, d=4, 20 particles)',
            'value': round(value, 1), 'unit': 'evals/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(elapsed / args.steps * 1e3, 4), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'PACOH-SVGD svgd_step, cfg#3: %d tasks/GPU x %d particles, n_ctx=%d, d=%d, '
                                   'NN(32,32) mean + NN(32,32) kernel (D=%d), task sharding + 1 RCCL all-reduce/step'
                                   % (TASKS_PER_GPU, PARTICLES, N_CTX, DIM, D),
                       'tasks_per_gpu': TASKS_PER_GPU, 'particles': PARTICLES, 'n_ctx': N_CTX, 'd': DIM,
                       'evals_per_step': T_global * PARTICLES, 'parallelism': 'task-shard x%d' % world,
                       'finite': finite},
            'roofline': roofline, 'kernel_rooflines': kernel_rooflines, 'gram_roofline': gram, 'kernel_ms_per_step': kernel_ms, 'cpu_baseline': cpu,
        }
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
