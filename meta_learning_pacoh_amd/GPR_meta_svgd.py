"""PACOH-SVGD on MI355X: API of GPRegressionMetaLearnedSVGD (meta_learn/GPR_meta_svgd.py:14-230).
One svgd_step = [MLP fwd x2] -> fused GP LML fwd+bwd over tasks x particles -> [MLP bwd x2] ->
hyper-prior grad -> SVGD phi -> Adam, all HIP kernels on one stream; with torch.distributed
initialised, tasks are sharded over ranks and the score is all-reduced once per step (RCCL)."""
import os
import time

import numpy as np
import torch

from . import _lib as L
from . import parallel
from .abstract import RegressionModelMetaLearned
from .distributions import GaussianPredictive
from .engine import (AsyncUploader, GPEngine, NotPSDError, ParamLayout, StepFeed, StepMode, TaskBatch, build_step_graphs,
                     first_chunk, replay_steps, run_step)
from .util import StepLR


def consume_vectorized_gp_init_rng(layout):
    """Advance the torch CPU generator exactly as building the reference's VectorizedGP does (every
    LinearVectorized draws normal(in*out), weight.uniform_, bias.uniform_; models.py:283-293; mean_nn
    before kernel_nn, random_gp.py:33-46) so that seeded runs draw the same initial particles."""
    def net(in_dim, out_dim, layers):
        prev = in_dim
        for size in list(layers) + [out_dim]:
            w = torch.normal(0, 1, size=(prev * size,))
            w.uniform_(-1.0, 1.0)
            torch.empty(size).uniform_(-1.0, 1.0)
            prev = size
    if layout.mean_module == 'NN':
        net(layout.input_dim, 1, layout.mean_nn_layers)
    if layout.covar_module == 'NN':
        net(layout.input_dim, layout.feature_dim, layout.kernel_nn_layers)


def sample_hyper_prior(layout, prior_mean, prior_std, n):
    """CatDist.sample: block after block torch.normal(loc.expand, scale.expand) (models.py:183-184)"""
    blocks = []
    for name, (lo, hi) in layout.slices.items():
        loc = prior_mean[lo:hi].expand(n, hi - lo)
        scale = prior_std[lo:hi].expand(n, hi - lo)
        blocks.append(torch.normal(loc, scale))
    return torch.cat(blocks, dim=-1)


def harmonic_pre_factor(sizes):
    """m~/(m~+T), m~ = harmonic mean of the batch's dataset sizes, T = batch length (random_gp.py:209-212)"""
    sizes = np.asarray(sizes, dtype=np.float32)
    hm = np.float32(1.0) / np.mean(np.float32(1.0) / sizes, dtype=np.float32)
    return float(hm / (hm + np.float32(len(sizes))))


class _RandomGPLearner(RegressionModelMetaLearned):
    """what SVGD and VI share: RandomGPMeta semantics (random_gp.py:116-222) on the device engine"""

    def _setup_random_gp(self, meta_train_data, mean_module, covar_module, mean_nn_layers, kernel_nn_layers,
                         task_batch_size):
        assert mean_module in ['NN', 'constant'] and covar_module in ['NN', 'SE']
        meta_train_data = list(meta_train_data)
        if task_batch_size < 1:
            self.task_batch_size = len(meta_train_data)
        else:
            self.task_batch_size = min(task_batch_size, len(meta_train_data))
        self._check_meta_data_shapes(meta_train_data)
        self._compute_normalization_stats(meta_train_data)
        # NB: the reference never forwards feature_dim to VectorizedGP -> always 2 (GPR_meta_svgd.py:167-170)
        self.layout = ParamLayout(self.input_dim, mean_module, covar_module, mean_nn_layers, kernel_nn_layers,
                                  feature_dim=2, with_outputscale=False)
        self.engine = GPEngine(self.layout, noise_floor=0.0)         # plain softplus (random_gp.py:73)
        pm, ps = self.layout.hyper_prior_mean_std(self.weight_prior_std, self.bias_prior_std)
        self._prior_mean_cpu, self._prior_std_cpu = pm, ps
        self.prior_mean = pm.to(self.dtype).to(self.device)
        self.prior_std = ps.to(self.dtype).to(self.device)
        consume_vectorized_gp_init_rng(self.layout)
        return meta_train_data

    def _setup_tasks(self, meta_train_data):
        tasks = [self._prepare_data_per_task(x, y) for x, y in meta_train_data]
        self.tasks = TaskBatch(tasks, self.device, self.dtype)

    def _take_idx(self, k):
        """the next k global task draws, int64 [k, B]: one randint call of shape [k, B] consumes the numpy stream exactly like k calls
        of size B (GPR_meta_svgd.py:102 draws one batch per iteration)"""
        return self.rds_numpy.randint(0, self.tasks.T, size=(k, self.task_batch_size))

    def _sample_task_batch(self):
        """global with-replacement draw from the shared seed (GPR_meta_svgd.py:102), then this rank's shard"""
        idx = self._take_idx(1)[0]
        pre = harmonic_pre_factor(self.tasks.sizes[idx])
        local = parallel.shard(idx)
        return local, pre

    # ---- the training loop's steps, captured in hipGraphs ---------------------------------------------------------------------
    # One step is ~12 launches; issued one by one from Python that is ~0.35 ms of host time, more than the GPU needs for a small
    # configuration or a 1/8 shard of a large one.  The step's kernels read everything that changes between steps (task draws,
    # pre-factor, learning rate, Adam's bias corrections) from device buffers filled for up to GRAPH_CHUNK steps at once
    # (engine.StepFeed), so the launch sequence is captured once and replayed.  With several ranks the step is two graphs around
    # the eager all-reduce of the packed score buffer.  PACOH_NO_GRAPH=1 runs the very same launch sequence eagerly.
    GRAPH_CHUNK = 1024

    def _graphs_allowed(self):
        return (os.environ.get('PACOH_NO_GRAPH', '0') != '1' and self.tasks.n <= L.gp_small_max_n(self.dtype, True)
                and not L.FORCE_DENSE)

    def _local_batch_size(self):
        return len(parallel.shard(np.arange(self.task_batch_size)))

    def _draw_steps(self, k, lr_scheduler, first_step, weight_decay=0.0):
        """task draws (this rank's shard) and step scalars of the next k steps, vectorised: one randint call of shape [k, B] consumes
        the numpy stream exactly like k calls of size B (GPR_meta_svgd.py:102 draws one batch per iteration)"""
        return self._rows_to_feed(self._take_idx(k), lr_scheduler, first_step, weight_decay)

    def _rows_to_feed(self, idx, lr_scheduler, first_step, weight_decay=0.0):
        """task draws idx [k, B] -> (this rank's shard of them, the steps' scalar rows)"""
        k = idx.shape[0]
        # harmonic pre-factor per step (random_gp.py:209-212); tasks of one size: every row is the same computation on the same
        # numbers -- done once
        rows = idx if self.tasks.ragged else idx[:1]
        sizes = self.tasks.sizes[rows].astype(np.float32)
        hm = np.float32(1.0) / np.mean(np.float32(1.0) / sizes, axis=1, dtype=np.float32)
        pre = (hm / (hm + np.float32(self.task_batch_size))).astype(np.float64)
        if not self.tasks.ragged:
            pre = np.repeat(pre, k)
        sc_rows = L.step_scalar_rows(pre, lr_scheduler.lrs(k), first_step, weight_decay=weight_decay)
        rank, world = parallel.world()
        local = np.ascontiguousarray(idx[:, rank::world])
        parallel.check_same_draws(local, sc_rows)
        return (local if local.shape[1] > 0 else None), sc_rows

    # Under-filled grids (round 6): the reference's own launchers run 2 tasks x 10 particles / samples of 20 points per step -- 20 GP
    # problems, for which networks forward -> GP -> networks backward -> slab reduction are four kernel latencies.  There the
    # likelihood half of a step is pacoh_svgd_task_step: one workgroup per (task, parameter row) does all three stages with the
    # activations in LDS, then the slab reduction -- two launches.  It wins while all its workgroups are resident at once (one round of
    # ~20 us latency chains: up to 256-768 problems by network size, profiles/r06_task_fused_crossover.txt); the library's workspace
    # query answers that question for the device at hand, larger grids stay on the throughput kernels.

    def _setup_task_fused(self, rows, tb_local):
        """-> workspace of L.svgd_task_step for `rows` parameter rows x tb_local tasks per step, or None (general launch sequence):
        fp32, RBF GP kernel, a shape inside the task-fused kernel's plan whose workgroups are all resident at once.
        PACOH_SVGD_TASK_FUSED=0 / 1: never / wherever the plan allows (tests, A/B)"""
        self._task_plan = self._task_ws = None
        force = os.environ.get('PACOH_SVGD_TASK_FUSED')
        if force == '0' or tb_local < 1 or self.dtype != torch.float32 or L.FORCE_DENSE:
            return None
        plan = L.MapPersistPlan(self.layout, self.tasks, tb_local, 0.0, [(0, self.layout.D)], self.dtype)
        ws = L.svgd_task_workspace(plan, rows, tb_local, self.device, any_size=force == '1')
        if ws is not None:
            self._task_plan, self._task_ws = plan, ws
        return ws

    def _check_numerics(self):
        """raise where the reference raises: gpytorch's psd_safe_cholesky -> NotPSDError (read at synchronisation points only)"""
        flag = getattr(self, '_fail', None)
        bad = flag is not None and int(flag.item()) != 0
        if not bad and parallel.world()[1] > 1 and getattr(self, '_lik', None) is not None:
            # another rank's shard failed: its NaN likelihood sums reach every rank through the all-reduce, so that all ranks raise
            # at the same synchronisation point (a rank raising alone would leave the others waiting in the next collective)
            bad = not bool(torch.isfinite(self._lik).all())
        if bad:
            if flag is not None:
                flag.zero_()
            raise NotPSDError('a task kernel matrix was not positive definite even after adding jitter (1e-6 .. 1e-4)')

    def _idx_uploader(self):
        up = getattr(self, '_idx_up', None)
        if up is None:
            up = self._idx_up = AsyncUploader(self.device, torch.int64)
        return up

    def _log_prob_and_score(self, theta, idx_local, pre_factor, with_prior=True):
        """RandomGPMeta.log_prob and its gradient (random_gp.py:204-222; svgd.py:15-16):
        log_prob[p] = prior_factor*log p(theta_p) + pre_factor * sum_t mll[t,p]
        (with_prior=False: the likelihood term and its score only -- the fused SVGD update adds the prior's score itself)"""
        P, D = theta.shape
        packed, score, lik = parallel.packed_score_buffer(P, D, theta.dtype, theta.device)   # lik = pre_factor * sum_t mll[t,p]
        if len(idx_local) > 0:
            batch = self.tasks.select(self._idx_uploader().upload(idx_local))
            self.engine.lml_and_grad(theta, batch, weight=pre_factor, lik_out=lik, lik_scale=pre_factor, grad_out=score)
        else:
            packed.zero_()
        lik, score = parallel.all_reduce_sum_(lik, score, packed)     # ONE exchange per step, no packing copies
        if not with_prior:
            return lik, score
        logprior = L.prior_logprob_grad(theta, self.prior_mean, self.prior_std, score, self.prior_factor)
        L.axpy(lik, logprior, self.prior_factor)                      # lik += prior_factor * log p(theta)
        return lik, score

    def _mixture_predict(self, theta, context_x, context_y, test_x, return_density, mixture=True):
        cx, cy, tx = self._prepare_predict(context_x, context_y, test_x)
        mu, var, cov, _ = self.engine.predict(theta, cx, cy, tx, want_cov=return_density)
        dist = GaussianPredictive(mu, var, cov, self.y_mean.reshape(-1)[0], self.y_std.reshape(-1)[0], mixture=mixture)
        if return_density:
            return dist
        return dist.mean.cpu().numpy(), dist.stddev.cpu().numpy()


class GPRegressionMetaLearnedSVGD(_RandomGPLearner):

    def __init__(self, meta_train_data, num_iter_fit=10000, feature_dim=1,
                 prior_factor=0.01, weight_prior_std=0.5, bias_prior_std=3.0,
                 covar_module='NN', mean_module='NN', mean_nn_layers=(32, 32), kernel_nn_layers=(32, 32),
                 optimizer='Adam', lr=1e-3, lr_decay=1.0, kernel='RBF', bandwidth=None, num_particles=10,
                 task_batch_size=-1, normalize_data=True, random_seed=None):
        """Arguments as in the reference (GPR_meta_svgd.py:16-44)."""
        super().__init__(normalize_data, random_seed)
        assert mean_module in ['NN', 'constant', 'zero'] and covar_module in ['NN', 'SE']
        assert optimizer in ['Adam', 'SGD']
        if kernel not in ('RBF', 'IMQ'):                       # GPR_meta_svgd.py:173-179
            raise NotImplementedError
        self.kernel = kernel
        assert num_particles <= L.SVGD_MAX_PARTICLES, 'too many particles for the SVGD kernels'
        self.num_iter_fit, self.prior_factor, self.feature_dim = num_iter_fit, prior_factor, feature_dim
        self.weight_prior_std, self.bias_prior_std = weight_prior_std, bias_prior_std
        self.num_particles, self.bandwidth, self.optimizer_name = num_particles, bandwidth, optimizer
        meta_train_data = self._setup_random_gp(meta_train_data, mean_module, covar_module, mean_nn_layers,
                                                kernel_nn_layers, task_batch_size)
        # initial particles = one draw from the hyper-prior (GPR_meta_svgd.py:182)
        particles = sample_hyper_prior(self.layout, self._prior_mean_cpu, self._prior_std_cpu, num_particles)
        self.particles = particles.to(self.dtype).to(self.device).contiguous()
        self.exp_avg = torch.zeros_like(self.particles)
        self.exp_avg_sq = torch.zeros_like(self.particles)
        self.opt_step = 0
        self.lr_scheduler = StepLR(lr, 1000, lr_decay)
        self._svgd_ws = None
        self._feed = self._graphs = None
        self._step_mode = StepMode()
        self._setup_tasks(meta_train_data)
        self.fitted = False

    # ---- one SVGD step = likelihood body -> [all-reduce] -> update body ----------------------------------------------------------
    def _setup_step(self, tb_local):
        if getattr(self, '_feed', None) is not None and self._feed.tb == tb_local:
            return
        P, D = self.particles.shape
        self._packed, self._score, self._lik = parallel.packed_score_buffer(P, D, self.dtype, self.device)
        self._fail = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._bw_out = torch.zeros(1, dtype=self.dtype, device=self.device)
        self._feed = StepFeed(self.device, self.dtype, tb_local, chunk=self.GRAPH_CHUNK)
        self._svgd_ws = L.svgd_update_workspace(self.particles, self._svgd_ws)
        self._graphs = None
        # Five launches per step instead of six (csrc/step_tail.h): the distance matrix rides in the forward launch, the next step's
        # scalars and task batch are fetched by the update launch.  A rank without tasks of its own has no forward launch: it keeps
        # the step_begin launch.  PACOH_SVGD_PIPELINE=0: the round-2 launch sequence (A/B measurements, bit-identity tests)
        # (the IMQ particle kernel has its own launch sequence: _body_update_imq)
        self._pipelined = tb_local > 0 and self.kernel == 'RBF' and os.environ.get('PACOH_SVGD_PIPELINE', '1') != '0'
        if self.kernel == 'IMQ':
            self._imq_phi = torch.empty_like(self.particles)
            self._imq_h = torch.empty(D, dtype=self.dtype, device=self.device) if self.bandwidth is None else None
            self._imq_ws = L.svgd_imq_workspace(self.particles)
        self._setup_task_fused(P, tb_local)
        if self._pipelined:
            self._feed.pipeline(self.tasks, self.engine, self.particles)
        # median bandwidth computed beside the hyper-parameter reduction instead of inside the update (P <= 64: one wavefront's sort)
        self._bw_ahead = self._pipelined and self.bandwidth is None and P <= 64

    def _body_likelihood(self):
        """score[P,D] = d/d theta_p sum_t mll[t,p] over this rank's tasks (unscaled), lik[P] the sums themselves"""
        if self._pipelined and self._task_ws is not None:
            L.svgd_task_step(self._task_plan, self.particles, self._feed.batch, self._feed.hyp, self._score, self._lik, 1.0, self._fail,
                             self._task_ws, svgd=(self.particles, self._svgd_ws, self._feed.ctr, self._bw_ahead))
            return
        if self._pipelined:
            self.engine.lml_and_grad(self.particles, self._feed.batch, weight=1.0, lik_out=self._lik, lik_scale=1.0,
                                     grad_out=self._score, fail_flag=self._fail, hypers=self._feed.hyp,
                                     svgd_tail=(self.particles, self._svgd_ws, self._feed.ctr, self._bw_ahead))
            return
        # select + gather + hyper transforms + the particles' distance matrix: one launch; the counter is advanced by the update
        # (IMQ: no distance matrix -- its bandwidths are per-dimension medians; the counter is advanced by the Adam launch, or
        # behind this one for SGD)
        if self.kernel == 'IMQ':
            batch, hyp = self._feed.begin(self.tasks, self.engine, self.particles, advance=self.optimizer_name != 'Adam')
        else:
            batch, hyp = self._feed.begin(self.tasks, self.engine, self.particles, advance=False, svgd=(self.particles, self._svgd_ws))
        if batch is None:
            self._packed.zero_()
            return
        if self._task_ws is not None:
            L.svgd_task_step(self._task_plan, self.particles, batch, hyp, self._score, self._lik, 1.0, self._fail, self._task_ws)
            return
        self.engine.lml_and_grad(self.particles, batch, weight=1.0, lik_out=self._lik, lik_scale=1.0, grad_out=self._score,
                                 fail_flag=self._fail, hypers=hyp)

    def _body_update(self):
        """prior score + pre-factor + bandwidth + phi + optimizer in one launch (distances: _body_likelihood), particles updated
        in place, step counter advanced"""
        if self.kernel == 'IMQ':
            return self._body_update_imq()
        self.last_bandwidth = self._bw_out
        if self._pipelined:
            L.svgd_update_next(self.particles, self._score, self.prior_mean, self.prior_std, self.prior_factor, self.bandwidth,
                               self.optimizer_name, self.exp_avg, self.exp_avg_sq, self._svgd_ws, self._bw_out, self._feed, self.tasks,
                               self._feed.hyper, bandwidth_ready=self._bw_ahead)
            return
        _, self._svgd_ws = L.svgd_update_dev(self.particles, self._score, self.prior_mean, self.prior_std, self.prior_factor,
                                             self.bandwidth, self.optimizer_name, self._feed.sc, self.exp_avg, self.exp_avg_sq,
                                             workspace=self._svgd_ws, bw_out=self._bw_out, dist_done=True,
                                             step_counter=self._feed.ctr)
        self.last_bandwidth = self._bw_out

    def _body_update_imq(self):
        """SVGD.step with the IMQ particle kernel (svgd.py:12-28, 58-77) on the step feed: every step-dependent scalar (pre-factor,
        learning rate, Adam's bias corrections) is read from the feed's selected row in device memory, so that the launch
        sequence is captured and replayed like the RBF one.  score <- pre * score + prior_factor * d log prior (random_gp.py:
        204-222), -phi from pacoh_svgd_phi_imq (per-dimension median bandwidths incl. the gradient through them), optimizer step."""
        sc = self._feed.sc
        L.prior_score_dev(self.particles, self.prior_mean, self.prior_std, self._score, self.prior_factor,
                          sc[L.SC_SCORE_SCALE:L.SC_SCORE_SCALE + 1])
        neg_phi, self.last_bandwidth, self._imq_ws = L.svgd_phi_imq(self.particles, self._score, bandwidth=self.bandwidth, neg=True,
                                                                    workspace=self._imq_ws, phi_out=self._imq_phi, h_out=self._imq_h)
        if self.optimizer_name == 'Adam':
            L.adam_step_dev(self.particles, neg_phi, self.exp_avg, self.exp_avg_sq, sc[L.SC_ADAM:L.SC_ADAM + 4],
                            step_counter=self._feed.ctr)
        else:
            L.scale_dev(neg_phi, sc[L.SC_LR:L.SC_LR + 1])          # particles -= lr * (-phi)
            L.axpy(self.particles, neg_phi, -1.0)

    def _exchange(self):
        parallel.all_reduce_buffer_(self._packed)         # ONE exchange per step: score [P, D] | lik [P], in place

    def _build_graphs(self):
        state = (self.particles, self.exp_avg, self.exp_avg_sq, self._feed.ctr, self._fail)
        saved = [t.clone() for t in state]
        # (the large-context path allocates O(tasks x n^2) scratch per step inside the graph's pool: one step per graph there)
        self._graphs, self._graph_many = build_step_graphs(self._body_likelihood, self._exchange, self._body_update, self._feed,
                                                           many_ok=self.tasks.n <= 128)
        for t, sv in zip(state, saved):
            t.copy_(sv)                                   # undo what the warm-up runs did
        if self._pipelined:
            self._feed.prologue()                         # (batch buffers, scalars and hyper-parameters of the restored particles)

    def _run_step(self, graphed):
        run_step(self._graphs, graphed, self._body_likelihood, self._exchange, self._body_update)

    def _train_steps(self, n_steps):
        """the next n_steps SVGD steps of the training loop (task draws from rds_numpy, lr from the scheduler)"""
        self._setup_step(self._local_batch_size())
        graphed = self._graphs_allowed()
        ramp = True
        while n_steps > 0:
            k = first_chunk(n_steps, self.GRAPH_CHUNK) if ramp else min(n_steps, self.GRAPH_CHUNK)
            ramp = False
            idx_rows, sc_rows = self._draw_steps(k, self.lr_scheduler, self.opt_step + 1)
            self._feed.upload(idx_rows, sc_rows)
            if self._pipelined:
                self._feed.prologue()
            if graphed and self._graphs is None:
                self._build_graphs()                      # (captured with real operands in the feed; state and counter are restored)
            if graphed:
                # replay or eager launches, whichever is faster here (engine.StepMode); several steps per replay where possible
                many = (lambda n: replay_steps(n, self._graphs[0], self._graph_many)) if len(self._graphs) == 1 else None
                self._step_mode.run(k, self._run_step, many)
            else:
                for _ in range(k):
                    self._run_step(False)
            self.opt_step += k
            for _ in range(k):
                self.lr_scheduler.step()
            n_steps -= k

    def svgd_step(self, idx_local, pre_factor):
        """SVGD.step (meta_learn/svgd.py:25-28) on an explicit task draw: particles.grad = -phi; optimizer.step()"""
        self._setup_step(len(idx_local))
        self.opt_step += 1
        self._feed.upload(np.asarray(idx_local).reshape(1, -1) if len(idx_local) > 0 else None,
                          [L.step_scalars(pre_factor, self.lr_scheduler.lr, self.opt_step)])
        if self._pipelined:
            self._feed.prologue()
        self._run_step(False)

    def meta_fit(self, valid_tuples=None, verbose=True, log_period=500, n_iter=None):
        """GPR_meta_svgd.py:82-121"""
        assert (valid_tuples is None) or (all([len(valid_tuple) == 4 for valid_tuple in valid_tuples]))
        t = time.time()
        if n_iter is None:
            n_iter = self.num_iter_fit
        itr = 0
        while itr < n_iter:
            nxt = 1 if itr == 0 else min(n_iter, (itr // log_period + 1) * log_period)      # up to the next log line
            self._train_steps(nxt - itr)                  # (both particle kernels: steps replayed from captured graphs)
            itr = nxt
            if itr == 1 or itr % log_period == 0:
                torch.cuda.synchronize()
                self._check_numerics()
                duration = time.time() - t
                t = time.time()
                message = 'Iter %d/%d - Time %.2f sec' % (itr, self.num_iter_fit, duration)
                if valid_tuples is not None:
                    valid_ll, valid_rmse, calibr_err = self.eval_datasets(valid_tuples)
                    message += ' - Valid-LL: %.3f - Valid-RMSE: %.3f - Calib-Err %.3f' % (valid_ll, valid_rmse, calibr_err)
                if verbose:
                    self.logger.info(message)
        self._check_numerics()
        self.fitted = True

    def predict(self, context_x, context_y, test_x, return_density=False):
        """GPR_meta_svgd.py:123-159: equal-weighted mixture over the particles' GP posteriors"""
        return self._mixture_predict(self.particles, context_x, context_y, test_x, return_density)

    def _eval_params(self, **kwargs):
        return (self.particles, True, False) if not kwargs else None

    def state_dict(self):
        return {'particles': self.particles.cpu().clone(), 'exp_avg': self.exp_avg.cpu().clone(),
                'exp_avg_sq': self.exp_avg_sq.cpu().clone(), 'step': self.opt_step, 'epoch': self.lr_scheduler.epoch}

    def load_state_dict(self, state_dict):
        self.particles.copy_(state_dict['particles']); self.exp_avg.copy_(state_dict['exp_avg']); self.exp_avg_sq.copy_(state_dict['exp_avg_sq'])
        self.opt_step, self.lr_scheduler.epoch = int(state_dict['step']), int(state_dict['epoch'])
