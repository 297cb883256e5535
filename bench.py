#!/usr/bin/env python
"""Benchmark of the PACOH task-GP hot path on MI355X.

Default workload (BASELINE.json configs[2], the one the metric is quoted on): PACOH-SVGD meta-training, 1024 tasks,
n_ctx = 64, d = 4, 20 particles, NN(32,32) mean + NN(32,32) kernel features (D = 2534 prior parameters per particle), fp32.
One "step" = one full SVGD step over every (task, particle) pair held by the job, exactly as meta_fit runs it (task draw ->
task gather -> per-particle MLP features -> fused Gram/Cholesky/solve/log-det LML and its gradient -> MLP backward ->
hyper-parameter reductions -> [all-reduce] -> prior score + SVGD phi + Adam), replayed from the captured step graph.
One "eval" = one (task, particle) LML + gradient.

    python bench.py --gpus 1 --steps 200 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus 8 --steps 200 --warmup 10 [--scaling strong]

--scaling weak (default): every rank holds 1024 tasks (global batch 1024 x N); --scaling strong: 1024 tasks in total, sharded
over the N ranks (cfg #3 as BASELINE.json words it).  Either way the score [20 x 2534] (+ 20 likelihood sums) is summed with
ONE all-reduce per step.  --config 2 | 4 | 5 times the other BASELINE configurations the same way (parity-test cases; the
headline metric is config 3).  Prints ONE JSON line on rank 0.
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

TASKS, N_CTX, DIM, PARTICLES = 1024, 64, 4, 20
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy peak)
FP32_PEAK_TFLOPS = 157.3       # fp32 vector peak == fp32-input MFMA peak (MI355X_MICROARCH.md)
FP64_PEAK_TFLOPS = 78.6        # fp64 matrix peak: AMD's MI355X figure; tools/mfma_peak.hip measures what v_mfma_f64_16x16x4 sustains
STEADY_STEPS = 200            # the extra region behind the timed one (the line's `steady` object)
PMC_PROFILE = os.path.join('profiles', 'r06_pmc_hbm_traffic.json')
DENSE_PMC_PROFILE = os.path.join('profiles', 'r06_dense_pmc_hbm_traffic.json')


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


def gp_flops(n, f):
    """n^2(3f+2) + n^3/3 + 2n^2 + [bwd] 2n^3/3 + 2n^2 + 4 n^2 f per LML+grad evaluation"""
    return n * n * (3 * f + 2) + n ** 3 / 3 + 2 * n * n + 2 * n ** 3 / 3 + 2 * n * n + 4 * n * n * f


def net_macs(d_in, layers, d_out):
    prev, w = d_in, 0
    for h in layers:
        w += prev * h
        prev = h
    return w + prev * d_out


def cpu_baseline(budget_s=12.0):
    """The CPU oracle (plain torch restatement of the reference's arithmetic, oracle/pacoh_oracle.py) on
    the host cores, on a bounded sample of the config-3 workload.  Reported baseline only."""
    from oracle import pacoh_oracle as O
    from meta_learning_pacoh_amd.util import host_cpu_budget
    cores = host_cpu_budget()                              # (what the cgroup lets this process use, not the cores it can see)
    threads_before = torch.get_num_threads()
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
    cands = sorted({1, min(4, cores), min(8, cores), cores})
    probe = {th: rate(False, th, budget_s / 8) for th in cands}
    best = max(probe, key=probe.get)
    batched = rate(False, best, budget_s / 4)
    looped = rate(True, best, budget_s / 4)
    torch.set_num_threads(threads_before)
    return {'value': round(batched, 1), 'unit': 'evals/s', 'cores': best, 'kind': 'port',
            'sample': '%d tasks x %d particles (n=%d, d=%d, NN/NN, fp32) LML+autograd score with the CPU oracle, fully '
                      'batched over tasks x particles, best of torch threads %s with %d usable CPUs (cgroup quota; %s evals/s); '
                      'reference-style python loop over tasks at %d threads: %.1f evals/s'
                      % (T_s, PARTICLES, N_CTX, DIM, cands, cores, {k: round(v) for k, v in probe.items()}, best, looped)}


def pmc_profile_state(path):
    """{'profile': path, 'profile_source_hash': ..., 'current_source_hash': ..., 'stale': bool}: is the committed PMC profile the
    `traffic` figures are read from still a profile of THESE kernel sources?  (a kernel change without
    tools/profile_round.sh used to leave stale traffic in the line silently)"""
    from meta_learning_pacoh_amd._build import source_hash
    cur, rec = source_hash(), None
    try:
        with open(os.path.join(ROOT, path)) as fh:
            rec = json.load(fh).get('source_hash')
    except Exception:
        pass
    return {'profile': path, 'profile_source_hash': rec, 'current_source_hash': cur, 'stale': rec != cur}


def pmc_traffic(substr):
    """HBM bytes per launch of the kernel whose name contains `substr`, from the COMMITTED rocprofv3 PMC profile of this
    workload (separate FETCH_SIZE / WRITE_SIZE passes, gfx950 x2 FETCH correction; tools/pmc_summary.py) -- not measured in
    this run"""
    try:
        with open(os.path.join(ROOT, PMC_PROFILE)) as fh:
            for k, v in json.load(fh)['kernels'].items():
                if substr in k:
                    return v['hbm_bytes_per_launch']
    except Exception:
        pass
    return None


def assemble_line(metric, value, world, steps, warmup, ms_per_step, scaling, dtype, backend, world_size_seen, leg, legs, host_ms,
                  step_mode, config, roofline, kernel_rooflines, step_flops, gram, pp, others, cpu):
    """the ONE JSON line of the driver's contract (tests/test_host_logic.py asserts its shape without a GPU).  At N > 1 on the
    headline configuration `legs` holds both scaling modes -- {'weak': {...}, 'strong': {...}}, each with ms_per_step, value,
    exchange, world_size_seen, all_reduce_us -- and the top-level value / ms_per_step are those of the `scaling` leg"""
    kernel_sum, prof_ms = pp['kernel_sum'], pp['ms_per_step']
    return {
        'metric': metric,
        'value': round(value, 1), 'unit': 'evals/s', 'n_gpus': world, 'steps': steps, 'warmup': warmup,
        'ms_per_step': round(ms_per_step, 4), 'higher_is_better': True, 'scaling': scaling,
        'vs_baseline': None, 'dtype': dtype, 'data': 'synthetic',
        'backend': backend, 'world_size_seen': world_size_seen,
        'exchange': leg.get('exchange'), 'all_reduce_us': leg.get('all_reduce_us'), 'legs': legs,
        'steady': leg.get('steady'),
        'host_ms_per_step': round(host_ms, 4), 'step_mode': step_mode,
        'config': config,
        'roofline': roofline, 'kernel_rooflines': kernel_rooflines,
        'step_algorithmic_tflops': round(step_flops / (ms_per_step * 1e-3) / 1e12, 3),
        'gram_roofline': gram,
        # per-kernel HIP-event times come from a SEPARATE pass that issues the same launch sequence eagerly (events cannot be
        # read out of a graph replay); that pass's own wall time per step is printed next to them: kernel_sum <= profile pass.
        # kernel_ms_per_step_events_raw: the intervals as measured; kernel_ms_per_step: less the event pair's own overhead
        # (schema 2, round 3 on: round-2 lines carry the raw values under kernel_ms_per_step)
        'schema': 2,
        'kernel_ms_per_step': pp['kernel_ms'],
        'kernel_ms_per_step_events_raw': pp['kernel_ms_raw'], 'event_overhead_us': round(pp['event_overhead_ms'] * 1e3, 2),
        'kernel_sum_ms_per_step': round(kernel_sum, 4),
        'profile_pass_ms_per_step': round(prof_ms, 4),
        'launch_gaps_ms_per_step': round(prof_ms - kernel_sum, 4),
        'other_configs': others,
        'cpu_baseline': cpu,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)     # 0.12 s of GPU time at config 3
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--scaling', choices=['weak', 'strong'], default='weak')
    ap.add_argument('--config', type=lambda v: int(v) if v.isdigit() else v, choices=[1, 2, 3, 4, 5, 'ref_svgd', 'ref_vi', 'shard128', 'ref_map'], default=3)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-other-configs', action='store_true')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    assert world == args.gpus, '--gpus must equal WORLD_SIZE (launch N>1 with torch.distributed.run)'
    # PACOH_BENCH_BACKEND=gloo lets the N > 1 code path be exercised on a box with fewer GPUs than ranks (ranks then share
    # devices; RCCL itself refuses two ranks per device).  The measured job always uses 'nccl' = RCCL over xGMI.
    backend = os.environ.get('PACOH_BENCH_BACKEND', 'nccl')
    dev_index = local_rank if backend == 'nccl' else local_rank % max(1, torch.cuda.device_count())
    if backend != 'nccl':
        os.environ['PACOH_SHARE_DEVICE'] = '1'
    torch.cuda.set_device(dev_index)
    # Host side of the GPU job: a handful of intra-op threads, never more than this process's share of the CPUs it may really use.  The
    # test hosts show 256 cores behind a cgroup quota of 16; torch's default pool (128 spinning workers) burnt that quota on the
    # per-chunk staging copies and the kernel froze the process for tens of milliseconds at a time -- 0.43 ms steps measured as 0.50-0.57
    # whenever such a freeze left the GPU idle in front of a short timed region (util.host_cpu_budget, tools/short_region_probe.py)
    from meta_learning_pacoh_amd.util import host_cpu_budget
    budget = max(1, host_cpu_budget() // max(1, world if backend == 'nccl' else 1))
    torch.set_num_threads(max(1, min(4, budget, torch.get_num_threads())))
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', dev_index))
        else:
            dist.init_process_group(backend)

    import meta_learning_pacoh_amd as M
    from meta_learning_pacoh_amd import _lib as L

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def exchange_name():
        if world == 1:
            return None
        from meta_learning_pacoh_amd import parallel
        comm = parallel._direct_comm()
        return ('pacoh_allreduce_sum (RCCL) on the compute stream, captured in the step graph' if comm is not None
                else 'torch.distributed.all_reduce between two graphs per step')

    def timed_leg(scaling, primary):
        """build the workload, warm it up, time EXACTLY args.steps steps between barrier + synchronize, MAX over ranks"""
        wl = WORKLOADS[args.config](world, scaling, M, L)
        # (enough untimed steps for the learners' first look at graph replay vs plain launches -- engine.StepMode -- and, in the
        #  chunks below, their second: neither decision falls into the timed region)
        wl['run'](max(40, args.warmup))
        barrier()
        # A fresh box needs a second or two of the real launch path before it issues steps at its steady rate (first process after
        # boot: hipGraphLaunch measured at 0.18 ms per step against 0.03 ms a minute later, enough to starve a 0.5 ms step).  More
        # untimed steps, in chunks, until two consecutive chunks take the same time (or 5 s have passed); every rank runs the same
        # number of chunks.
        prev, t_warm = None, time.perf_counter()
        for _ in range(40):
            t_c = time.perf_counter()
            wl['run'](128)                                # (16 + 112 steps: engine.first_chunk; the second look needs a chunk of >= 60)
            barrier()
            cur = time.perf_counter() - t_c
            stable = prev is not None and abs(cur - prev) <= 0.03 * prev and time.perf_counter() - t_warm >= 1.0
            flag = torch.tensor([1.0 if (stable or time.perf_counter() - t_warm > 5.0) else 0.0], device='cuda')
            if world > 1:
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if float(flag.item()) > 0:
                break
            prev = cur
        t0 = time.perf_counter()
        wl['run'](args.steps)
        t_issued = time.perf_counter()                   # the host has issued every step (nothing synchronises inside)
        barrier()
        elapsed = time.perf_counter() - t0
        host_ms = (t_issued - t0) / args.steps * 1e3
        if world > 1:
            tt = torch.tensor([elapsed], dtype=torch.float64, device='cuda')
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
        # the steady state, driver-visible: one more region of STEADY_STEPS steps right behind the --steps region (same workload, same
        # bracketing).  The top-level value stays the --steps region's; short regions also pay the clock ramp after the idle barrier
        # (tools/README.md), which this one dilutes.
        t1 = time.perf_counter()
        wl['run'](STEADY_STEPS)
        barrier()
        steady_s = time.perf_counter() - t1
        if world > 1:
            tt = torch.tensor([steady_s], dtype=torch.float64, device='cuda')
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            steady_s = float(tt.item())
        leg = {'scaling': scaling, 'ms_per_step': round(elapsed / args.steps * 1e3, 4),
               'steady': {'steps': STEADY_STEPS, 'ms_per_step': round(steady_s / STEADY_STEPS * 1e3, 4),
                          'value': round(wl['evals_per_step'] * STEADY_STEPS / steady_s, 1)},
               'value': round(wl['evals_per_step'] * args.steps / elapsed, 1), 'unit': 'evals/s',
               'evals_per_step': wl['evals_per_step'], 'host_ms_per_step': round(host_ms, 4), 'finite': wl['finite'](),
               'exchange': exchange_name(), 'world_size_seen': (dist.get_world_size() if world > 1 else 1),
               'step_mode': wl.get('mode', lambda: None)(), 'tasks_total': wl.get('extra', {}).get('tasks_total')}
        return leg, wl, elapsed

    leg, wl, elapsed = timed_leg(args.scaling, True)
    gpu_only = wl['gpu_only']() if ('gpu_only' in wl and world == 1) else None
    device_noise = wl['device_noise']() if ('device_noise' in wl and world == 1) else None
    finite = leg['finite']
    ms_per_step = elapsed / args.steps * 1e3
    host_ms = leg['host_ms_per_step']
    evals_per_step, metric, dtype, describe, extra = wl['evals_per_step'], wl.get('metric'), wl['dtype'], wl['describe'], wl.get('extra', {})
    step_mode = leg['step_mode']
    exchange = leg['exchange']

    pp = profile_pass(wl, L, 50)                         # (its own step count: --steps 20 would make the per-kernel averages noisy)
    kernel_ms, kernel_sum, prof_ms = pp['kernel_ms'], pp['kernel_sum'], pp['ms_per_step']
    roofline, kernel_rooflines, step_flops = rooflines(wl, pp)

    def all_reduce_us(pp_):
        """HIP-event time of the step's one exchange in the eager per-kernel pass (the in-graph collective cannot be timed from
        outside the graph): RCCL's kernel on the compute stream, or torch.distributed's call"""
        for key in ('allreduce_sum', 'allreduce_torch'):
            if key in pp_['kernel_ms_raw']:
                return round(pp_['kernel_ms_raw'][key] * 1e3, 2)
        return None
    leg['all_reduce_us'] = all_reduce_us(pp) if world > 1 else None

    # N > 1, headline configuration: BOTH scaling modes from the one invocation the driver makes -- weak (1024 tasks per GPU) and
    # strong (1024 tasks in total: cfg #3 as BASELINE.json words it).  The line's own value / ms_per_step are the --scaling leg's.
    legs = None
    if world > 1 and args.config == 3:
        legs = {args.scaling: leg}
        other = 'strong' if args.scaling == 'weak' else 'weak'
        del wl
        torch.cuda.empty_cache()
        leg2, wl2, _ = timed_leg(other, False)
        leg2['all_reduce_us'] = all_reduce_us(profile_pass(wl2, L, 20))
        legs[other] = leg2
        # what the first real record needs to explain its own efficiency: a rank of the weak leg does the N = 1 job's work, so the
        # strong leg's ideal step is that time / N; what is above it is the one-round-of-workgroups floor of a small shard plus the exchange
        legs['strong']['ideal_ms_per_step'] = round(legs['weak']['ms_per_step'] / world, 4)
        legs['strong']['ideal_source'] = 'legs.weak.ms_per_step (1024 tasks per GPU = the N = 1 job) / %d' % world
        wl = wl2

    gram = gram_leg(L) if (rank == 0 and args.config == 3) else None

    # the other BASELINE configurations (parity-test cases, not the headline): bounded legs on the same box so that their numbers
    # are driver-visible too -- N = 1, default config only
    others = predictive = None
    if rank == 0 and world == 1 and args.config == 3 and not args.no_other_configs:
        del wl
        torch.cuda.empty_cache()
        others = {}
        for c in (1, 2, 4, 5, 'ref_svgd', 'ref_vi', 'ref_map', 'shard128'):
            key = 'cfg%d' % c if isinstance(c, int) else c
            try:
                others[key] = other_config_leg(c, M, L)
            except Exception as exc:                     # (a side leg must never take the headline line down with it)
                others[key] = {'error': repr(exc)}
                torch.cuda.empty_cache()
        try:
            predictive = predictive_leg(L)
        except Exception as exc:
            predictive = {'error': repr(exc)}

    if rank == 0:
        cpu = None if (args.no_cpu_baseline or world > 1 or args.config != 3) else cpu_baseline()
        value = evals_per_step * args.steps / elapsed
        out = assemble_line(
            metric='task-GP LML+grad evals/sec (n_ctx=64, d=4, 20 particles)' if args.config == 3 else metric, value=value,
            world=world, steps=args.steps, warmup=args.warmup, ms_per_step=ms_per_step, scaling=args.scaling, dtype=dtype,
            backend=(backend if world > 1 else None), world_size_seen=(dist.get_world_size() if world > 1 else 1), leg=leg, legs=legs,
            host_ms=host_ms, step_mode=step_mode,
            config=dict({'workload': describe, 'evals_per_step': evals_per_step, 'parallelism': 'task-shard x%d' % world,
                         'finite': finite}, **extra),
            roofline=roofline, kernel_rooflines=kernel_rooflines, step_flops=step_flops, gram=gram, pp=pp, others=others, cpu=cpu)
        if gpu_only is not None:
            out['gpu_ms_per_step_noise_resident'] = gpu_only      # (PACOH-VI: the step without the host's noise draw, _vi_gpu_only)
        if device_noise is not None:
            out['ms_per_step_device_noise'] = device_noise        # (PACOH-VI with noise='device': wl_ref_vi)
        out['predictive'] = predictive                    # (row A11 beside the LML: predictive_leg; None outside the default N = 1 run)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def predictive_leg(L, reps=20):
    """SURVEY 8 row A11 beside the LML: the marginal posterior predictive (mean + variance at m test points per problem) at the
    context sizes of cfg #3 and cfg #4, fp32, NN-feature dimension 2 -- pacoh_gp_predict, inputs resident, HIP events"""
    out = {}
    for tag, B, n, m in (('cfg3_shape', 20480, 64, 64), ('cfg4_shape', 5120, 128, 128)):
        f = 2
        g = torch.Generator().manual_seed(n)
        X = torch.randn(B, n, f, generator=g).cuda(); Y = torch.randn(B, n, generator=g).cuda(); Xs = torch.randn(B, m, f, generator=g).cuda()
        ls = torch.full((1, f), 0.6931).cuda(); nz = torch.tensor([0.313]).cuda()
        run = lambda: L.gp_predict(X, 1, None, L.MEAN_ZERO, Y, 1, Xs, 1, None, ls, None, nz, B, 1)
        for _ in range(3):
            res = run()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            res = run()
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / reps
        out[tag] = {'problems': B, 'n_ctx': n, 'm_test': m, 'ms_per_call': round(ms, 4), 'value': round(B / (ms * 1e-3), 1),
                    'unit': 'predictive evals/s', 'finite': bool(torch.isfinite(res[0]).all() and torch.isfinite(res[1]).all()),
                    'info_max': int(res[3].abs().max())}
        del X, Y, Xs
    return out


def event_overhead_ms(L, reps=64):
    """what a HIP-event pair adds to the kernel between them: the median interval around an (almost) empty launch.  An event is a
    packet of its own in the queue: the kernel behind the start event cannot be dispatched before that packet has retired, and
    the end event's timestamp is taken when ITS packet is processed -- 8-12 us per timed scope on this stack, 5-10 % of the
    0.1-0.18 ms kernels of the step, which rocprofv3 (it reads the kernels' own dispatch timestamps) does not see.  The empty
    kernel itself runs about 2 us, so the corrected times err on the short side by that much."""
    buf = torch.ones(1, device='cuda')
    one = torch.ones(1, device='cuda')
    for _ in range(8):
        L.scale_dev(buf, one)
    torch.cuda.synchronize()
    # the pairs are queued BEHIND a few milliseconds of other work, as the kernels of the profile pass are (the host runs ahead of
    # the GPU there): measured on an idle queue the interval would be the host's time between two calls, not the GPU's
    big = torch.empty(64 << 20, device='cuda')
    for _ in range(40):
        big.fill_(1.0)
    pairs = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        L.scale_dev(buf, one)
        e.record()
        pairs.append((s, e))
    torch.cuda.synchronize()
    iv = sorted(a.elapsed_time(b) for a, b in pairs)
    return iv[len(iv) // 2]


def profile_pass(wl, L, prof_steps):
    """per-kernel times with HIP events on the launch stream: the SAME launch sequence issued eagerly (PACOH_NO_GRAPH=1), and the
    wall time per step of that very pass.  kernel_ms: event intervals minus the event pair's own overhead (event_overhead_ms) per
    timed scope; kernel_ms_raw: the intervals as measured"""
    os.environ['PACOH_NO_GRAPH'] = '1'
    wl['run'](6)
    torch.cuda.synchronize()
    L.PROFILE = {}
    t0 = time.perf_counter()
    wl['run'](prof_steps)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    raw = L.profile_summary()
    L.PROFILE = None
    os.environ.pop('PACOH_NO_GRAPH')
    ov = event_overhead_ms(L)
    prof = {k: (n, max(t - n * ov, 0.0)) for k, (n, t) in raw.items()}
    return {'prof': prof, 'steps': prof_steps, 'ms_per_step': wall / prof_steps * 1e3, 'event_overhead_ms': ov,
            'kernel_ms': {k: round(v[1] / prof_steps, 4) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])},
            'kernel_ms_raw': {k: round(v[1] / prof_steps, 4) for k, v in sorted(raw.items(), key=lambda kv: -kv[1][1])},
            'kernel_sum': sum(v[1] for v in prof.values()) / prof_steps}


def rooflines(wl, pp):
    prof, prof_steps = pp['prof'], pp['steps']
    raw_ms = {k: v * prof_steps for k, v in pp['kernel_ms_raw'].items()}
    peak = FP64_PEAK_TFLOPS if wl['dtype'] == 'f64' else FP32_PEAK_TFLOPS

    def kernel_roofline(name):
        launches, tot_ms = prof[name]
        per_step_s = tot_ms / prof_steps * 1e-3
        alg, exe = wl['flops'][name]
        key = wl.get('pmc_keys', {}).get(name)
        return {'kernel': name, 'bound': 'mfma', 'achieved': round(alg / per_step_s / 1e12, 3), 'peak': peak, 'unit': 'TFLOP/s',
                'frac': round(alg / per_step_s / 1e12 / peak, 5), 'algorithmic_frac': round(alg / per_step_s / 1e12 / peak, 5),
                'executed_frac': round(exe / per_step_s / 1e12 / peak, 5),
                'ms_per_step': round(per_step_s * 1e3, 4), 'launches_per_step': launches / prof_steps,
                'traffic': pmc_traffic(key) if key else None,
                'traffic_source': 'committed PMC profile %s (not measured in this run)' % PMC_PROFILE,
                'traffic_profile': pmc_profile_state(PMC_PROFILE),
                'ms_per_step_events_raw': round(raw_ms[name] / prof_steps, 4), 'event_overhead_us': round(pp['event_overhead_ms'] * 1e3, 2),
                'note': 'algorithmic flops per step and GPU (SURVEY 8d model; MLP backward = 4 n W, the part of the forward it still '
                        'recomputes -- the first layer, the rest comes from the activation stash -- is counted only in executed_frac) '
                        '/ HIP-event time of the kernel in this run, less the event pair\'s own overhead per timed scope '
                        '(event_overhead_us: the interval around an empty launch; rocprofv3 reads the dispatch timestamps and does '
                        'not see it) / %s peak %.1f TFLOP/s'
                        % ('fp64 matrix' if wl['dtype'] == 'f64' else 'fp32', peak)}

    modelled = [k for k in wl['flops'] if k in prof]
    dom = max(modelled, key=lambda k: prof[k][1]) if modelled else None
    roofline = kernel_roofline(dom) if dom else None
    kernel_rooflines = {k: {kk: r[kk] for kk in ('achieved', 'algorithmic_frac', 'executed_frac', 'ms_per_step', 'unit')}
                        for k, r in ((k, kernel_roofline(k)) for k in modelled)}
    return roofline, kernel_rooflines, sum(wl['flops'][k][0] for k in modelled)


def cpu_baseline_other(cfg, budget_s=2.5):
    """bounded CPU baseline (<= ~3 s) beside configurations #2 / #4 / #5, with the oracle (kind "port"): the same LML + gradient arithmetic
    in plain torch on the host cores.  #2: PACOH-MAP loss + autograd over the 256 sinusoid tasks, batched over tasks and as the
    reference's python loop; #4: the [S, D] score of a 16-task sample at n = 128; #5: fp64 batched LAPACK Cholesky LML + autograd
    gradients on a 4-problem sample (n = 512, d = 8)."""
    from oracle import pacoh_oracle as O
    from meta_learning_pacoh_amd.util import host_cpu_budget
    cores = host_cpu_budget()
    threads = max(1, min(8, cores))
    before = torch.get_num_threads()
    torch.set_num_threads(threads)

    def rate(fn, units):
        fn()                                               # warm-up
        t0, reps = time.time(), 0
        while time.time() - t0 < budget_s / 2:
            fn()
            reps += 1
        return units * reps / (time.time() - t0)
    try:
        if cfg == 2:
            tasks = sinusoid_tasks(27, 256, 32)
            orc = O.MapOracle(tasks, covar_module='SE', mean_module='NN', task_batch_size=256, random_seed=1)
            xs = torch.stack([x for x, _ in orc.tasks]); ys = torch.stack([y for _, y in orc.tasks])

            def batched():
                orc.optimizer.zero_grad()
                (-orc.task_mll(xs, ys).sum()).backward()

            def looped():
                orc.optimizer.zero_grad()
                loss = 0.0
                for x, y in orc.tasks:
                    loss = loss - orc.task_mll(x, y)
                loss.backward()
            vb, vl = rate(batched, 256), rate(looped, 256)
            sample = '256 sinusoid tasks x n=32 (SinusoidDataset(RandomState(27))), SE kernel + NN(32,32) mean, fp32 loss + autograd: batched over tasks %.0f evals/s, reference-style python loop %.0f evals/s' % (vb, vl)
            value = vb
        elif cfg == 'ref_map':
            # the reference's PACOH-MAP launcher defaults restated: 2 tasks x 5 points per iteration through two 4 x 128 networks, AdamW
            layers = (128, 128, 128, 128)
            orc = O.MapOracle(sinusoid_tasks(29, 20, 5), mean_nn_layers=layers, kernel_nn_layers=layers, task_batch_size=2, weight_decay=0.0,
                              lr_decay=0.98, lr_params=1e-3, random_seed=28, num_iter_fit=10)
            value = rate(lambda: orc.meta_fit(None, log_period=1000, n_iter=10), 2 * 10)
            sample = ('10 PACOH-MAP iterations at the launcher shape per sample (2 tasks x 5 points, NN(128,128,128,128) mean + kernel, fp32): '
                      'loss + autograd + AdamW with the oracle')
        elif cfg in ('ref_svgd', 'ref_vi'):
            # one step's [P, D] score at the launcher shape: 2 tasks x 10 particles / samples, n = 20, 4 x 32 networks
            tasks = sinusoid_tasks(29, 20, 20)[:2]
            stats = O.compute_normalization_stats(tasks)
            otasks = [O.prepare_task(x, y, stats, torch.float32) for x, y in tasks]
            gcfg = O.GPConfig(1, 'NN', 'NN', REF_LAYERS, REF_LAYERS)
            pm, ps = O.hyperprior_mean_std(gcfg.layout, 0.5, 3.0)
            torch.manual_seed(0)
            theta = O.hyperprior_sample(gcfg.layout, pm, ps, 10)
            vb = rate(lambda: O.meta_score(theta, otasks, gcfg, pm, ps, 0.1, loop=False), 20)
            vl = rate(lambda: O.meta_score(theta, otasks, gcfg, pm, ps, 0.1, loop=True), 20)
            sample = ('one step at the launcher shape: 2 tasks x 10 particles (n=20, d=1, NN(32,32,32,32) mean + kernel, fp32) LML + autograd score: '
                      'batched %.0f evals/s, reference-style python loop over tasks %.0f evals/s' % (vb, vl))
            value = max(vb, vl)
        elif cfg == 4:
            T_s, S = 16, 10
            tasks = _rand_tasks(T_s, 128, 1, 28)
            stats = O.compute_normalization_stats(tasks)
            otasks = [O.prepare_task(x, y, stats, torch.float32) for x, y in tasks]
            gcfg = O.GPConfig(1, 'NN', 'NN')
            pm, ps = O.hyperprior_mean_std(gcfg.layout, 0.5, 3.0)
            torch.manual_seed(0)
            theta = O.hyperprior_sample(gcfg.layout, pm, ps, S)
            vb = rate(lambda: O.meta_score(theta, otasks, gcfg, pm, ps, 0.01, loop=False), T_s * S)
            vl = rate(lambda: O.meta_score(theta, otasks, gcfg, pm, ps, 0.01, loop=True), T_s * S)
            sample = '%d tasks x %d samples (n=128, d=1, NN/NN, fp32) LML + autograd score: batched %.0f evals/s, python loop over tasks %.0f evals/s' % (T_s, S, vb, vl)
            value = vb
        else:
            Bs, n, d = 4, 512, 8
            g = torch.Generator().manual_seed(5)
            X = torch.randn(Bs, n, d, dtype=torch.float64, generator=g).requires_grad_(True)
            Y = torch.randn(Bs, n, dtype=torch.float64, generator=g)
            ls = torch.full((1, 1, d), 0.6931, dtype=torch.float64, requires_grad=True)
            nz = torch.tensor(0.313, dtype=torch.float64, requires_grad=True)

            def lml():
                for v in (X, ls, nz):
                    v.grad = None
                O.gp_mll(X, torch.zeros(Bs, n, dtype=torch.float64), Y, ls, 1.0, nz).sum().backward()
            value = rate(lml, Bs)
            sample = '%d problems (n=512, d=8, fp64): Gram + batched LAPACK Cholesky LML + autograd gradients' % Bs
        return {'value': round(value, 1), 'unit': 'evals/s', 'cores': threads, 'kind': 'port', 'sample': sample}
    finally:
        torch.set_num_threads(before)


def cfg5_hbm_report(L):
    """BASELINE config #5's HBM-roofline report: the fp64 Gram leg at n = 512, d = 8 measured here (HIP events, the public entry point on
    one point set: full symmetric matrices, 2.13 MB per Gram; the path itself launches the lower block triangle only), and the
    large-context kernels' HBM traffic against one pass over the matrices from the COMMITTED PMC profile (not measured in this run)."""
    B, n, d = 256, 512, 8
    z = torch.randn(B, n, d, dtype=torch.float64, device='cuda')
    ls = torch.full((1, d), 0.6931, dtype=torch.float64, device='cuda')
    nz = torch.tensor([0.313], dtype=torch.float64, device='cuda')
    K = torch.empty(B, n, n, dtype=torch.float64, device='cuda')
    lib = L.load_library()
    st = torch.cuda.current_stream().cuda_stream

    def run():
        lib.pacoh_gram_rbf_ard(z.data_ptr(), 1, z.data_ptr(), 1, ls.data_ptr(), None, nz.data_ptr(), 1, K.data_ptr(), B, 1, n, n, d, 1, st)
    for _ in range(5):
        run()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        run()
    e.record()
    torch.cuda.synchronize()
    t_k = s.elapsed_time(e) / 20 * 1e-3
    alg = B * (n * d * 8 + n * n * 8)
    out = {'gram': {'kernel': 'gram_rbf_ard fp64 n=512 d=8 (full matrices, one point set)', 'bound': 'hbm', 'algorithmic_bytes': alg, 'bytes_per_gram': n * d * 8 + n * n * 8,
                    'us_per_launch': round(t_k * 1e6, 1), 'achieved': round(alg / t_k / 1e9, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                    'frac': round(alg / t_k / 1e9 / HBM_PEAK_GBS, 4)}}
    try:
        with open(os.path.join(ROOT, DENSE_PMC_PROFILE)) as fh:
            prof = json.load(fh)
        one_pass = B * n * n * 8
        out['traffic'] = {'source': 'committed PMC profile %s (not measured in this run)' % DENSE_PMC_PROFILE, 'profile_state': pmc_profile_state(DENSE_PMC_PROFILE),
                          'one_pass_over_the_matrices_bytes': one_pass,
                          'kernels': {k: {'hbm_bytes_per_pass': v, 'ratio_to_one_pass': round(v / one_pass, 2)} for k, v in prof['per_pass'].items()
                                      if not k.startswith(('at::', '__amd'))}}
    except Exception:
        out['traffic'] = None
    return out


def other_config_leg(cfg, M, L, steps=256):
    """one of BASELINE.json's other configurations, timed like the headline one but bounded (about 2-3 s): three untimed (64 / 128 / 64-step)
    calls (both looks of the learners at graph replay vs plain launches -- engine.StepMode -- happen there, not in the timed
    region), `steps` timed ones between synchronisations (two 128-step noise chunks of PACOH-VI), then the eager per-kernel pass"""
    wl = WORKLOADS[cfg](1, 'weak', M, L)
    for n in (64, 128, 64):
        wl['run'](n)
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    wl['run'](steps)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    gpu_only = wl['gpu_only']() if 'gpu_only' in wl else None
    pp = profile_pass(wl, L, min(steps, 30))
    roofline, _, step_flops = rooflines(wl, pp)
    out = {'metric': wl['metric'], 'workload': wl['describe'], 'dtype': wl['dtype'], 'steps': steps, 'ms_per_step': round(ms, 5),
           'value': round(wl['evals_per_step'] / (ms * 1e-3), 1), 'unit': 'evals/s', 'finite': wl['finite'](),
           'step_algorithmic_tflops': round(step_flops / (ms * 1e-3) / 1e12, 3),
           'dominant_kernel': None if roofline is None else {k: roofline[k] for k in ('kernel', 'achieved', 'peak', 'unit', 'algorithmic_frac', 'ms_per_step')},
           'kernel_ms_per_step': pp['kernel_ms'], 'profile_pass_ms_per_step': round(pp['ms_per_step'], 4)}
    if gpu_only is not None:
        # (ms_per_step above includes the host's per-step noise draw from torch's CPU generator -- the reference's stream; this is the
        #  same step replayed with the noise already resident: _vi_gpu_only)
        out['gpu_ms_per_step_noise_resident'] = gpu_only
    if 'device_noise' in wl:
        out['ms_per_step_device_noise'] = wl['device_noise']()
    del wl
    torch.cuda.empty_cache()
    if cfg == 5:
        out['hbm'] = cfg5_hbm_report(L)
    if cfg in (2, 4, 5, 'ref_svgd', 'ref_vi', 'ref_map'):
        try:
            out['cpu_baseline'] = cpu_baseline_other(cfg)
        except Exception as exc:
            out['cpu_baseline'] = {'error': repr(exc)}
    return out


def gram_leg(L):
    """standalone Gram build (the HBM-write-bound kernel): the config-3 problem count, materialised K"""
    B = TASKS * PARTICLES
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
    return {'kernel': 'gram_rbf_ard', 'bound': 'hbm', 'achieved': round(alg_bytes / t_k / 1e9, 1),
            'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(alg_bytes / t_k / 1e9 / HBM_PEAK_GBS, 4),
            'traffic': pmc_traffic('gram_kernel<float, 2'),
            'traffic_source': 'committed PMC profile %s (not measured in this run)' % PMC_PROFILE,
            'traffic_profile': pmc_profile_state(PMC_PROFILE),
            'algorithmic_bytes': alg_bytes, 'bytes_per_gram': N_CTX * 2 * 4 + N_CTX * N_CTX * 4, 'grams': B,
            'us_per_launch': round(t_k * 1e6, 2)}


# ---- workloads: each returns run(n_steps), evals per step (whole job), flop models per HIP-event name (per step and GPU) ----
def _rand_tasks(T, n, d, seed):
    rs = np.random.RandomState(seed)
    return [(rs.uniform(-5, 5, (n, d)), rs.normal(size=(n, 1))) for _ in range(T)]


def sinusoid_tasks(seed, n_tasks, n_samples):
    """the reference's SinusoidDataset(RandomState(seed)).generate_meta_train_data(n_tasks, n_samples) (experiments/data_sim.py:203-248,
    same RNG call order: amplitude, x shift, y shift, slope, period, then X, then the noise): the tasks SURVEY.md 8d names for
    configurations #1 (seed 26, 20 x 5 = demo.py:17) and #2 (seed 27, 256 x 32)"""
    rs = np.random.RandomState(seed)
    out = []
    for _ in range(n_tasks):
        amp, x_shift = rs.uniform(0.7, 1.3), rs.normal(loc=0.0, scale=0.1)
        y_shift, slope = rs.normal(loc=5.0, scale=0.1), rs.normal(loc=0.5, scale=0.2)
        period = rs.uniform(1.5, 1.5)
        X = rs.uniform(-5, 5, size=(n_samples, 1))
        f = slope * X + amp * np.sin(period * (X - x_shift)) + y_shift
        out.append((X, f + 0.1 * rs.normal(size=f.shape)))
    return out


def _mode(model):
    """how the learner ended up issuing its steps: {'graph': bool, 'eager_ms': .., 'graph_ms': ..} (engine.StepMode)"""
    sm = model._step_mode
    return dict({'graph': bool(sm.use_graph) if sm.use_graph is not None else True}, **(sm.timings or {}))


def wl_cfg3(world, scaling, M, L):
    T_global = TASKS * world if scaling == 'weak' else TASKS
    model = M.GPRegressionMetaLearnedSVGD(make_tasks(T_global, N_CTX, DIM), num_particles=PARTICLES, covar_module='NN',
                                          mean_module='NN', task_batch_size=-1, lr=1e-3, random_seed=0)
    ev = T_global * PARTICLES / world
    w = net_macs(DIM, (32, 32), 1) + net_macs(DIM, (32, 32), 2)                 # 2400 MAC per point over both networks
    w1 = 2 * DIM * 32                                                           # ... of which the first layers (recomputed by the backward)
    return dict(run=model._train_steps, evals_per_step=T_global * PARTICLES, dtype='f32', mode=lambda: _mode(model),
                finite=lambda: bool(torch.isfinite(model.particles).all()),
                flops={'gp_lml_fwdbwd': (gp_flops(N_CTX, 2) * ev,) * 2, 'mlp_fwd': (2 * N_CTX * w * ev,) * 2,
                       'mlp_bwd': (4 * N_CTX * w * ev, (4 * w + 2 * w1) * N_CTX * ev)},
                pmc_keys={'gp_lml_fwdbwd': 'gp_reg_kernel', 'mlp_fwd': 'mlp_fused_fwd', 'mlp_bwd': 'mlp_fused_bwd'},
                describe='PACOH-SVGD step as meta_fit runs it (hipGraph replay), cfg#3: %d tasks %s x %d particles, n_ctx=%d, d=%d, '
                         'NN(32,32) mean + NN(32,32) kernel (D=%d), task sharding + 1 all-reduce/step'
                         % (TASKS, 'per GPU' if scaling == 'weak' else 'in total', PARTICLES, N_CTX, DIM, model.layout.D),
                extra={'tasks_total': T_global, 'particles': PARTICLES, 'n_ctx': N_CTX, 'd': DIM})


def wl_cfg1(world, scaling, M, L):
    """BASELINE.json configs[0], the reference's demo (demo.py:14-26): PACOH-MAP on 20 sinusoid tasks x 5 points, 5 tasks per iteration,
    NN(32,32) mean + kernel features, AdamW with weight decay 0.2 -- a latency-sized iteration (four launches)"""
    model = M.GPRegressionMetaLearned(sinusoid_tasks(26, 20, 5), task_batch_size=5, weight_decay=0.2, random_seed=30)
    w = net_macs(1, (32, 32), 1) + net_macs(1, (32, 32), 2)
    ev = 5 / world
    return dict(run=model._train_steps, evals_per_step=5, dtype='f32', finite=lambda: bool(torch.isfinite(model.theta).all()), mode=lambda: _mode(model),
                metric='task-GP LML+grad evals/sec (PACOH-MAP demo: 20 tasks x 5 points, 5 per iteration)',
                flops={'gp_lml_fwdbwd': (gp_flops(5, 2) * ev,) * 2, 'mlp_fwd': (2 * 5 * w * ev,) * 2, 'mlp_bwd': (4 * 5 * w * ev, 6 * 5 * w * ev),
                       'map_persist': ((gp_flops(5, 2) + 6 * 5 * w) * ev,) * 2},        # (the whole iteration is one kernel: pacoh_map_persist)
                describe='PACOH-MAP iteration (K iterations per launch of the persistent kernel where the shape fits, else hipGraph replay), cfg#1 = the reference\'s demo: SinusoidDataset(RandomState(26)) 20 tasks x 5 points, 5 tasks per iteration, NN(32,32) mean + kernel, AdamW',
                extra={'tasks_total': 20, 'n_ctx': 5, 'd': 1})


def wl_cfg2(world, scaling, M, L):
    """PACOH-MAP, 256 sinusoid tasks (SinusoidDataset(RandomState(27))), n_ctx = 32, d = 1, SE kernel + NN(32,32) mean, the full
    task batch every iteration"""
    T = 256 * world if scaling == 'weak' else 256
    model = M.GPRegressionMetaLearned(sinusoid_tasks(27, T, 32), covar_module='SE', mean_module='NN', task_batch_size=T, random_seed=1)
    ev = T / world
    w = net_macs(1, (32, 32), 1)
    return dict(run=model._train_steps, evals_per_step=T, dtype='f32', finite=lambda: bool(torch.isfinite(model.theta).all()), mode=lambda: _mode(model),
                metric='task-GP LML+grad evals/sec (PACOH-MAP, 256 tasks, n_ctx=32, d=1, SE kernel)',
                flops={'gp_lml_fwdbwd': (gp_flops(32, 1) * ev,) * 2, 'mlp_fwd': (2 * 32 * w * ev,) * 2,
                       'mlp_bwd': (4 * 32 * w * ev, 6 * 32 * w * ev),
                       'map_task_step': ((gp_flops(32, 1) + 6 * 32 * w) * ev,) * 2},    # (forward + GP + backward in one launch + the slab reduction)
                describe='PACOH-MAP iteration (hipGraph replay; two launches per iteration: pacoh_map_task_step), cfg#2: %d tasks x n_ctx=32, d=1, SE kernel + NN(32,32) mean, AdamW' % T,
                extra={'tasks_total': T, 'n_ctx': 32, 'd': 1})


def wl_cfg4(world, scaling, M, L):
    """PACOH-VI, 512 tasks, n_ctx = 128, NN(32,32) mean + kernel features, 10 posterior samples per step"""
    T, S = (512 * world if scaling == 'weak' else 512), 10
    model = M.GPRegressionMetaLearnedVI(_rand_tasks(T, 128, 1, 28), svi_batch_size=S, random_seed=1)
    ev = T * S / world
    w = net_macs(1, (32, 32), 1) + net_macs(1, (32, 32), 2)
    return dict(run=model._train_steps, evals_per_step=T * S, dtype='f32', finite=lambda: bool(torch.isfinite(model.posterior).all()), mode=lambda: _mode(model),
                gpu_only=_vi_gpu_only(model),
                metric='task-GP LML+grad evals/sec (PACOH-VI, 512 tasks, n_ctx=128, 10 posterior samples)',
                flops={'gp_lml_fwdbwd': (gp_flops(128, 2) * ev,) * 2, 'mlp_fwd': (2 * 128 * w * ev,) * 2,
                       'mlp_bwd': (4 * 128 * w * ev, (4 * w + 2 * 2 * 32) * 128 * ev)},
                describe='PACOH-VI step (hipGraph replay), cfg#4: %d tasks x %d samples, n_ctx=128, d=1, NN(32,32) mean + kernel' % (T, S),
                extra={'tasks_total': T, 'samples': S, 'n_ctx': 128, 'd': 1})


def wl_cfg5(world, scaling, M, L):
    """Large-context stress: 256 tasks, n_ctx = 512, d = 8, fp64, SE kernel: LML + gradient through the HBM-resident path.
    A single GP does not shard: N ranks run N replicas (no collective)."""
    B, n, d = 256, 512, 8
    g = torch.Generator().manual_seed(5)
    X = torch.randn(B, n, d, dtype=torch.float64, generator=g).cuda()
    Y = torch.randn(B, n, dtype=torch.float64, generator=g).cuda()
    ls = torch.full((1, d), 0.6931, dtype=torch.float64, device='cuda')
    nz = torch.tensor([0.313], dtype=torch.float64, device='cuda')
    os1 = torch.ones(1, dtype=torch.float64, device='cuda')
    last = {}

    def run(k):
        for _ in range(k):
            last['out'] = L.gp_lml_fwdbwd(X, 1, None, L.MEAN_ZERO, Y, 1, ls, os1, nz, B, 1)
    fl = gp_flops(n, d) * B
    return dict(run=run, evals_per_step=B * world, dtype='f64', finite=lambda: bool(torch.isfinite(last['out'][0]).all()),
                metric='task-GP LML+grad evals/sec (n_ctx=512, d=8, fp64)',
                flops={'gp_lml_dense': (fl, fl)},
                describe='cfg#5 large-context stress: %d GPs x n_ctx=512, d=8, fp64, SE kernel, LML + gradients (Gram -> MFMA-panel '
                         'Cholesky -> triangular inverse -> Z^T Z -> contractions); replicas only across ranks' % B,
                extra={'problems_per_gpu': B, 'n_ctx': n, 'd': d})


REF_LAYERS = (32, 32, 32, 32)     # num_layers=4, layer_size=32 (experiments/meta_GPR_SVGD_base_exp.py:29-30, meta_GPR_vi_base_exp.py:29-30)


def _ref_launcher_flops(S):
    w = net_macs(1, REF_LAYERS, 1) + net_macs(1, REF_LAYERS, 2)
    ev = 2 * S
    return {'gp_lml_fwdbwd': (gp_flops(20, 2) * ev,) * 2, 'mlp_fwd': (2 * 20 * w * ev,) * 2, 'mlp_bwd': (4 * 20 * w * ev, 6 * 20 * w * ev),
            'svgd_task_step': ((gp_flops(20, 2) + 6 * 20 * w) * ev,) * 2}


def _vi_gpu_only(model):
    """PACOH-VI draws S x D standard normals per step from torch's CPU generator (the reference's stream, GPR_meta_vi.py:216-224 through
    Normal.rsample): ~3 ns per number on one host thread, which at the launchers' shape (10 x 6566) is 0.2 ms per step -- several times what
    the GPU needs.  This times the GPU's share alone: 16 rows of the chunk uploaded last (noise resident) replayed 8 times from the
    captured step graphs, nothing drawn in between.  The optimizer state moves on: a timing, not training."""
    def go():
        from meta_learning_pacoh_amd.engine import replay_steps
        graphs = getattr(model, '_graphs', None)
        if not graphs or len(graphs) != 1:
            return None
        def burst():
            model._feed.ctr.zero_()
            replay_steps(16, graphs[0], model._graph_many)
        burst()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(8):
            burst()
        torch.cuda.synchronize()
        return round((time.perf_counter() - t0) / 128 * 1e3, 5)
    return go


def wl_ref_svgd(world, scaling, M, L):
    """the reference's own PACOH-SVGD launcher at its defaults (experiments/meta_GPR_SVGD_base_exp.py:23-49, 79-103): seed 28,
    SinusoidDataset(RandomState(29)) 20 tasks x 20 points, task_batch_size = 2, 10 particles, 4 x 32 mean and kernel networks, Adam
    1e-3 with decay 0.98, prior_factor 0.1, RBF particle kernel with bandwidth 0.1 -- 20 GP problems per step (VERDICT r5 missing #1)"""
    model = M.GPRegressionMetaLearnedSVGD(sinusoid_tasks(29, 20, 20), weight_prior_std=0.5, prior_factor=0.1, covar_module='NN',
                                          mean_module='NN', kernel_nn_layers=REF_LAYERS, mean_nn_layers=REF_LAYERS, random_seed=28,
                                          optimizer='Adam', lr=1e-3, lr_decay=0.98, kernel='RBF', bandwidth=0.1, num_particles=10,
                                          task_batch_size=2)
    return dict(run=model._train_steps, evals_per_step=20, dtype='f32', mode=lambda: _mode(model),
                finite=lambda: bool(torch.isfinite(model.particles).all()),
                metric='task-GP LML+grad evals/sec (PACOH-SVGD at the reference launcher\'s defaults: 2 tasks x 10 particles per step, n_ctx=20)',
                flops=_ref_launcher_flops(10),
                describe='PACOH-SVGD step at the defaults of experiments/meta_GPR_SVGD_base_exp.py: 20 sinusoid tasks x 20 points, '
                         'task_batch_size=2, 10 particles, NN(32,32,32,32) mean + kernel (D=%d), bandwidth 0.1' % model.layout.D,
                extra={'tasks_total': 20, 'particles': 10, 'n_ctx': 20, 'd': 1})


def wl_ref_vi(world, scaling, M, L):
    """the reference's PACOH-VI launcher at its defaults (experiments/meta_GPR_vi_base_exp.py:23-52, 86-103): as wl_ref_svgd with a
    diagonal Gaussian posterior and svi_batch_size = 10 samples per step"""
    model = M.GPRegressionMetaLearnedVI(sinusoid_tasks(29, 20, 20), weight_prior_std=0.5, prior_factor=0.1, covar_module='NN',
                                        mean_module='NN', kernel_nn_layers=REF_LAYERS, mean_nn_layers=REF_LAYERS, random_seed=28,
                                        optimizer='Adam', lr=1e-3, lr_decay=0.98, svi_batch_size=10, cov_type='diag', task_batch_size=2)
    def device_noise():
        # the same learner with noise='device' (an option the reference does not have: the reparameterisation noise from the device
        # generator, nothing drawn on the host) as meta_fit runs it: what the step costs once the host's stream is out of the way
        m2 = M.GPRegressionMetaLearnedVI(sinusoid_tasks(29, 20, 20), weight_prior_std=0.5, prior_factor=0.1, covar_module='NN',
                                         mean_module='NN', kernel_nn_layers=REF_LAYERS, mean_nn_layers=REF_LAYERS, random_seed=28,
                                         optimizer='Adam', lr=1e-3, lr_decay=0.98, svi_batch_size=10, cov_type='diag', task_batch_size=2,
                                         noise='device')
        for n in (64, 128, 64):
            m2._train_steps(n)
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        m2._train_steps(256)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 256 * 1e3
        return round(ms, 5) if bool(torch.isfinite(m2.posterior).all()) else None
    return dict(run=model._train_steps, evals_per_step=20, dtype='f32', mode=lambda: _mode(model), gpu_only=_vi_gpu_only(model),
                device_noise=device_noise,
                finite=lambda: bool(torch.isfinite(model.posterior).all()),
                metric='task-GP LML+grad evals/sec (PACOH-VI at the reference launcher\'s defaults: 2 tasks x 10 samples per step, n_ctx=20)',
                flops=_ref_launcher_flops(10),
                describe='PACOH-VI step at the defaults of experiments/meta_GPR_vi_base_exp.py: 20 sinusoid tasks x 20 points, '
                         'task_batch_size=2, 10 posterior samples, NN(32,32,32,32) mean + kernel, diagonal posterior',
                extra={'tasks_total': 20, 'samples': 10, 'n_ctx': 20, 'd': 1})


def wl_ref_map(world, scaling, M, L):
    """the reference's PACOH-MAP launcher at its defaults (experiments/meta_GPR_mll_base_exp.py:22-47, 79-99): seed 28,
    SinusoidDataset(RandomState(29)) 20 tasks x 5 points, 2 tasks per iteration, 4 x 128 mean and kernel networks, AdamW 1e-3 with decay
    0.98 and weight_decay 0 -- ten points through 128-wide layers per iteration: launch latency and nothing else"""
    layers = (128, 128, 128, 128)
    model = M.GPRegressionMetaLearned(sinusoid_tasks(29, 20, 5), learning_mode='both', covar_module='NN', mean_module='NN', mean_nn_layers=layers,
                                      kernel_nn_layers=layers, weight_decay=0.0, lr_params=1e-3, lr_decay=0.98, task_batch_size=2, random_seed=28,
                                      optimizer='Adam')
    w = net_macs(1, layers, 1) + net_macs(1, layers, 2)
    return dict(run=model._train_steps, evals_per_step=2, dtype='f32', finite=lambda: bool(torch.isfinite(model.theta).all()), mode=lambda: _mode(model),
                metric='task-GP LML+grad evals/sec (PACOH-MAP at the reference launcher\'s defaults: 2 tasks x 5 points per iteration, 4 x 128 networks)',
                flops={'gp_lml_fwdbwd': (gp_flops(5, 2) * 2,) * 2, 'mlp_fwd': (2 * 5 * w * 2,) * 2, 'mlp_bwd': (4 * 5 * w * 2, 6 * 5 * w * 2),
                       'map_task_step': ((gp_flops(5, 2) + 6 * 5 * w) * 2,) * 2},      # (forward + GP + backward in one workgroup: map_wide_kernel)
                describe='PACOH-MAP iteration at the defaults of experiments/meta_GPR_mll_base_exp.py: 20 sinusoid tasks x 5 points, batch_size=2, '
                         'NN(128,128,128,128) mean + kernel (D=%d), AdamW' % model.layout.D,
                extra={'tasks_total': 20, 'n_ctx': 5, 'd': 1})


def wl_shard128(world, scaling, M, L):
    """one rank's share of BASELINE config #3 strong-scaled over 8 GPUs: 128 of the 1024 tasks x 20 particles (no exchange: N = 1)"""
    global TASKS
    keep, TASKS = TASKS, 128
    try:
        wl = wl_cfg3(1, 'weak', M, L)
    finally:
        TASKS = keep
    wl['metric'] = 'task-GP LML+grad evals/sec (cfg#3\'s 1/8 strong-scaling shard: 128 tasks x 20 particles, n_ctx=64, d=4)'
    return wl


WORKLOADS = {1: wl_cfg1, 2: wl_cfg2, 3: wl_cfg3, 4: wl_cfg4, 5: wl_cfg5, 'ref_svgd': wl_ref_svgd, 'ref_vi': wl_ref_vi,
             'shard128': wl_shard128, 'ref_map': wl_ref_map}

if __name__ == '__main__':
    main()
