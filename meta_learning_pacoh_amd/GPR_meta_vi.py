"""PACOH-VI on MI355X: API of GPRegressionMetaLearnedVI (meta_learn/GPR_meta_vi.py:14-262) with the
diagonal or full-covariance Gaussian variational hyper-posterior (random_gp.py:224-251).  The reparameterised
ELBO gradient needs only the per-sample score from the device engine:
  diag:  theta_s = loc + exp(scale) * eps_s          full:  theta_s = loc + tril(tril_cov) eps_s
  elbo_s  = log p(theta_s) - prior_factor * log q(theta_s),   loss = -mean_s elbo_s
  dloss/dloc = -mean_s score_s;   dloss/dscale = -mean_s (score_s * exp(scale) * eps_s + prior_factor)
  dloss/dL_ij = -mean_s score_si eps_sj - [i == j] prior_factor / L_ii   (j <= i; zero above the diagonal)."""
import math
import os
import time

import torch

from . import _lib as L
from . import parallel
from .engine import AsyncUploader, StepFeed, StepMode, build_step_graphs, first_chunk, replay_steps, run_step
from .GPR_meta_svgd import _RandomGPLearner
from .util import StepLR

LOG_2PI = math.log(2 * math.pi)


def init_vi_posterior(D, init_std=0.1):
    """RandomGPPosterior.__init__ for cov_type='diag' (meta_learn/random_gp.py:244-247): loc ~ N(0, 0.1),
    scale (= log std) ~ N(log 0.1, 0.1), drawn in this order from the torch CPU generator -> [2, D]"""
    loc = torch.normal(0.0, init_std, size=(D,))
    scale = torch.normal(math.log(0.1), init_std, size=(D,))
    return torch.stack([loc, scale])


def init_vi_posterior_full(D, init_std=0.1):
    """RandomGPPosterior.__init__ for cov_type='full' (random_gp.py:244,249-250): loc ~ N(0, 0.1), then
    tril_cov = diag(U(0.05, 0.1)) -> [D+1, D] (row 0 = loc, rows 1.. = tril_cov)"""
    loc = torch.normal(0.0, init_std, size=(D,))
    tril = torch.diag(torch.ones(D).uniform_(0.05, 0.1))
    return torch.cat([loc.reshape(1, D), tril], dim=0)


def standard_normal(n, D, out=None):
    """the eps of Normal(loc, scale).rsample((n,)) (torch.distributions: _standard_normal = torch.normal(zeros, ones)): the same use of
    the CPU generator, hence the same values, as the reference's.  ATen evaluates normal(mean tensor, std tensor) as
    `out.normal_(0, 1); out.mul_(std).add_(mean)`: filling with normal_() directly gives the same bits from the same generator state
    (tests/test_host_logic.py pins it for sizes on both sides of the generator's 16-element blocks) without the two extra passes --
    0.29 -> 0.24 ms per draw of 10 x 6566 on the build host, and the draw is what bounds a PACOH-VI step at the reference launcher's
    shape.  out (float32 [n, D], e.g. a row of pinned staging memory): filled in place"""
    if out is not None and out.dtype == torch.float32 and out.is_contiguous():
        return out.normal_()
    eps = torch.empty(n, D).normal_()
    if out is not None:
        out.copy_(eps)
    return eps


class GPRegressionMetaLearnedVI(_RandomGPLearner):

    def __init__(self, meta_train_data, num_iter_fit=10000, feature_dim=1,
                 prior_factor=0.01, weight_prior_std=0.5, bias_prior_std=3.0,
                 covar_module='NN', mean_module='NN', mean_nn_layers=(32, 32), kernel_nn_layers=(32, 32),
                 optimizer='Adam', lr=1e-3, lr_decay=1.0, svi_batch_size=10, cov_type='diag',
                 task_batch_size=-1, normalize_data=True, random_seed=None, noise='host'):
        """Arguments as in the reference (GPR_meta_vi.py:16-44), plus one it does not have:
        noise: 'host' (default) draws the reparameterisation noise of every rsample from the torch CPU generator -- the reference's
               stream, value for value; on one host thread that is ~2.3 ns per number, which BOUNDS the step wherever the GPU needs less
               than S x D x 2.3 ns (the launchers' shape: 0.156 ms per step against 0.051 ms of GPU work).  'device' fills the same
               buffers from the device's generator instead (torch.manual_seed seeds it too; same distribution, another stream):
               meta_fit, predict and eval_datasets then draw nothing on the host."""
        super().__init__(normalize_data, random_seed)
        assert mean_module in ['NN', 'constant', 'zero'] and covar_module in ['NN', 'SE']
        assert optimizer in ['Adam', 'SGD']
        assert cov_type in ['diag', 'full']
        assert noise in ['host', 'device']
        self.cov_type, self.noise = cov_type, noise
        self.num_iter_fit, self.prior_factor, self.feature_dim = num_iter_fit, prior_factor, feature_dim
        self.weight_prior_std, self.bias_prior_std = weight_prior_std, bias_prior_std
        self.svi_batch_size, self.optimizer_name = svi_batch_size, optimizer
        meta_train_data = self._setup_random_gp(meta_train_data, mean_module, covar_module, mean_nn_layers,
                                                kernel_nn_layers, task_batch_size)
        # RandomGPPosterior init (random_gp.py:244-247), torch CPU generator, then moved to the device
        init = init_vi_posterior if cov_type == 'diag' else init_vi_posterior_full
        self.posterior = init(self.layout.D).to(self.dtype).to(self.device).contiguous()   # [2, D] | [D+1, D]
        self.exp_avg = torch.zeros_like(self.posterior)
        self.exp_avg_sq = torch.zeros_like(self.posterior)
        self.opt_step = 0
        self.lr_scheduler = StepLR(lr, 1000, lr_decay)
        self._feed = self._graphs = None
        self._step_mode = StepMode()
        self._setup_tasks(meta_train_data)
        self.fitted = False

    @property
    def loc(self):
        return self.posterior[0]

    @property
    def scale(self):
        """diag: log std [D];  full: the tril_cov parameter [D, D]"""
        return self.posterior[1] if self.cov_type == 'diag' else self.posterior[1:]

    def _rsample(self, n):
        """Normal(loc, exp(scale)).rsample((n,)): eps from the torch CPU generator (reference stream);
        returns (theta[n,D], eps[n,D], log q(theta)[n])"""
        if self.noise == 'device':
            eps = torch.empty(n, self.layout.D, dtype=self.dtype, device=self.device).normal_()
        else:
            up = getattr(self, '_eps_up', None)
            if up is None:
                up = self._eps_up = AsyncUploader(self.device, self.dtype)
            eps = up.upload(standard_normal(n, self.layout.D))
        theta, log_q = L.vi_sample(self.posterior, eps, full=self.cov_type == 'full')
        return theta, eps, log_q

    def get_neg_elbo_and_grad(self, idx_local, pre_factor):
        """GPR_meta_vi.py:216-224 plus its backward, -> (loss, grad[2, D] | grad[D+1, D])"""
        S = self.svi_batch_size
        theta, eps, log_q = self._rsample(S)
        log_prob, score = self._log_prob_and_score(theta, idx_local, pre_factor)
        loss = torch.empty((), dtype=theta.dtype, device=theta.device)      # -mean_s (log p(theta_s) - prior_factor log q(theta_s))
        L.reduce_tasks(log_prob.reshape(S, 1, 1), loss.reshape(1, 1), scale=-1.0 / S)
        L.reduce_tasks(log_q.reshape(S, 1, 1), loss.reshape(1, 1), scale=self.prior_factor / S, accumulate=True)
        return loss, L.vi_grad(self.posterior, eps, score, self.prior_factor, full=self.cov_type == 'full')

    # ---- one VI step as hipGraph(s): sample + likelihood score -> [all-reduce] -> ELBO gradient + Adam ---------------------------
    def _setup_step(self, tb_local):
        if getattr(self, '_feed', None) is not None and self._feed.tb == tb_local:
            return
        S, D = self.svi_batch_size, self.layout.D
        self._packed, self._score, self._lik = parallel.packed_score_buffer(S, D, self.dtype, self.device)
        self._fail = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._loss = torch.zeros((), dtype=self.dtype, device=self.device)
        self._vi_ws = L.vi_update_workspace(self.posterior)
        # The noise of a chunk is drawn on the host, one rsample per step in the reference's stream order (0.1 ms per step at S = 10,
        # D = 2.5 k): chunks of 128 steps, so that drawing chunk k+1 overlaps the GPU running chunk k instead of preceding it
        # (measured at cfg #4, 200-step calls: one chunk 0.577 ms per step, 128-step chunks 0.52-0.55, 64-step chunks 0.62)
        chunk = max(1, min(self.GRAPH_CHUNK, int(os.environ.get('PACOH_VI_CHUNK', '128')), (64 << 20) // (S * D * 4)))
        self._feed = StepFeed(self.device, self.dtype, tb_local, chunk=chunk, aux_shape=(S, D))
        self._graphs = None
        self._setup_task_fused(S, tb_local)              # (under-filled grids: the task-fused likelihood launch, GPR_meta_svgd.py)

    def _body_likelihood(self):
        hyp = None
        if self.cov_type == 'diag' and os.environ.get('PACOH_VI_UNFUSED') != '1':
            # select (incl. the step's noise) + task gather + the step's samples, their log q and transformed hyper-parameters: one
            # launch; the update launch at the end of the step advances the counter
            batch, hyp, self._theta, self._log_q = self._feed.begin_vi(self.tasks, self.engine, self.posterior, self.svi_batch_size)
        else:
            batch, _ = self._feed.begin(self.tasks, advance=False)
            self._theta, self._log_q = L.vi_sample(self.posterior, self._feed.aux, full=self.cov_type == 'full')
        if batch is None:
            self._packed.zero_()
            return
        if hyp is not None and self._task_ws is not None:
            L.svgd_task_step(self._task_plan, self._theta, batch, hyp, self._score, self._lik, 1.0, self._fail, self._task_ws)
            return
        self.engine.lml_and_grad(self._theta, batch, weight=1.0, lik_out=self._lik, lik_scale=1.0, grad_out=self._score,
                                 fail_flag=self._fail, hypers=hyp)

    def _body_update(self):
        S = self.svi_batch_size
        if self.cov_type == 'diag' and os.environ.get('PACOH_VI_UNFUSED') != '1':      # (Adam: _train_steps is only used with it)
            L.vi_update_dev(self.posterior, self._feed.aux, self._theta, self._score, self._lik, self._log_q, self.prior_mean,
                            self.prior_std, self.prior_factor, self._feed.sc, self.exp_avg, self.exp_avg_sq, self._loss,
                            self._vi_ws, step_counter=self._feed.ctr)
            return
        L.scale_dev(self._packed, self._feed.sc[L.SC_SCORE_SCALE:L.SC_SCORE_SCALE + 1])      # pre-factor on score and likelihood
        logprior = L.prior_logprob_grad(self._theta, self.prior_mean, self.prior_std, self._score, self.prior_factor)
        L.axpy(self._lik, logprior, self.prior_factor)
        L.reduce_tasks(self._lik.reshape(S, 1, 1), self._loss.reshape(1, 1), scale=-1.0 / S)
        L.reduce_tasks(self._log_q.reshape(S, 1, 1), self._loss.reshape(1, 1), scale=self.prior_factor / S, accumulate=True)
        grad = L.vi_grad(self.posterior, self._feed.aux, self._score, self.prior_factor, full=self.cov_type == 'full')
        L.adam_step_dev(self.posterior, grad, self.exp_avg, self.exp_avg_sq, self._feed.sc[L.SC_ADAM:L.SC_ADAM + 4],
                        step_counter=self._feed.ctr)

    def _exchange(self):
        parallel.all_reduce_buffer_(self._packed)         # ONE exchange per step: score [S, D] | lik [S], in place

    def _build_graphs(self):
        state = (self.posterior, self.exp_avg, self.exp_avg_sq, self._feed.ctr, self._fail)
        saved = [t.clone() for t in state]
        # (the large-context path allocates O(tasks x n^2) scratch per step inside the graph's pool: one step per graph there)
        self._graphs, self._graph_many = build_step_graphs(self._body_likelihood, self._exchange, self._body_update, self._feed,
                                                           many_ok=self.tasks.n <= 128)
        for t, sv in zip(state, saved):
            t.copy_(sv)

    def _run_step(self, graphed):
        run_step(self._graphs, graphed, self._body_likelihood, self._exchange, self._body_update)

    def _train_steps(self, n_steps):
        self._setup_step(self._local_batch_size())
        graphed = self._graphs_allowed()
        S, D = self.svi_batch_size, self.layout.D
        k = 0
        while n_steps > 0:
            # chunk sizes 16, 24, 36, 52, ...: the host draws 0.15-0.19 ms of noise per step, the GPU needs 0.4-0.5 ms per step -- a
            # chunk must take the host less to prepare than the one in flight takes the GPU, or the GPU runs dry (growth 2 left a
            # margin of 10 % on the slower hosts: cfg #4 read 0.42 or 0.49 ms per step depending on the box)
            k = first_chunk(n_steps, self._feed.chunk) if k == 0 else min(n_steps, self._feed.chunk, (k + k // 2 + 3) // 4 * 4)
            idx_rows, sc_rows = self._draw_steps(k, self.lr_scheduler, self.opt_step + 1)
            # the reference's stream: one rsample per step, drawn straight into the pinned staging rows (noise='device': the chunk's
            # rows filled by one launch of the device generator, nothing drawn or copied on the host)
            self._feed.upload(idx_rows, sc_rows, 'device' if self.noise == 'device' else (lambda j, out: standard_normal(S, D, out=out)))
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
        return self._loss

    def meta_fit(self, valid_tuples=None, verbose=True, log_period=500, n_iter=None):
        """GPR_meta_vi.py:84-128"""
        assert (valid_tuples is None) or (all([len(valid_tuple) == 4 for valid_tuple in valid_tuples]))
        t = time.time()
        if n_iter is None:
            n_iter = self.num_iter_fit
        loss = None
        itr = 0
        while itr < n_iter:
            nxt = 1 if itr == 0 else min(n_iter, (itr // log_period + 1) * log_period)      # up to the next log line
            if self.optimizer_name == 'Adam':
                loss = self._train_steps(nxt - itr)
            else:
                for _ in range(nxt - itr):
                    idx_local, pre = self._sample_task_batch()
                    loss, grad = self.get_neg_elbo_and_grad(idx_local, pre)
                    self.opt_step += 1
                    L.axpy(self.posterior, grad, -self.lr_scheduler.lr)
                    self.lr_scheduler.step()
            itr = nxt
            if itr == 1 or itr % log_period == 0:
                duration = time.time() - t
                t = time.time()
                message = 'Iter %d/%d - Loss: %.6f - Time %.2f sec' % (itr, self.num_iter_fit, loss.item(), duration)
                self._check_numerics()
                if valid_tuples is not None:
                    valid_ll, valid_rmse, calibr_err = self.eval_datasets(valid_tuples)
                    message += ' - Valid-LL: %.3f - Valid-RMSE: %.3f - Calib-Err %.3f' % (valid_ll, valid_rmse, calibr_err)
                if verbose:
                    self.logger.info(message)
        self.fitted = True
        out = loss.item() if loss is not None else float('nan')
        self._check_numerics()
        return out

    def predict(self, context_x, context_y, test_x, n_posterior_samples=100, mode='Bayes', return_density=False):
        """GPR_meta_vi.py:130-174: 'Bayes' averages over posterior samples, 'MAP' uses the posterior mode"""
        assert mode in ['bayes', 'Bayes', 'MAP', 'map']
        if mode in ('Bayes', 'bayes'):
            theta, _, _ = self._rsample(n_posterior_samples)
            return self._mixture_predict(theta, context_x, context_y, test_x, return_density, mixture=True)
        theta = self.loc.reshape(1, -1).contiguous()
        return self._mixture_predict(theta, context_x, context_y, test_x, return_density, mixture=False)

    def _eval_params(self, n_posterior_samples=100, mode='Bayes', **kwargs):
        """eval_datasets in one batched pass: predict() draws n_posterior_samples fresh parameter rows per call, so T tasks need
        T draws from the torch CPU generator in task order (the reference's stream) -- one sampling launch over all T*n rows"""
        if kwargs:
            return None
        assert mode in ['bayes', 'Bayes', 'MAP', 'map']
        if mode in ('MAP', 'map'):
            return self.loc.reshape(1, -1).contiguous(), False, False

        def draw(T):
            if self.noise == 'device':
                eps = torch.empty(T * n_posterior_samples, self.layout.D, dtype=self.dtype, device=self.device).normal_()
            else:
                eps = torch.cat([standard_normal(n_posterior_samples, self.layout.D) for _ in range(T)]).to(self.dtype).to(self.device)
            theta, _ = L.vi_sample(self.posterior, eps, full=self.cov_type == 'full')
            return theta
        return draw, True, True

    def state_dict(self):
        return {'posterior': self.posterior.cpu().clone(), 'exp_avg': self.exp_avg.cpu().clone(),
                'exp_avg_sq': self.exp_avg_sq.cpu().clone(), 'step': self.opt_step, 'epoch': self.lr_scheduler.epoch}

    def load_state_dict(self, state_dict):
        self.posterior.copy_(state_dict['posterior']); self.exp_avg.copy_(state_dict['exp_avg']); self.exp_avg_sq.copy_(state_dict['exp_avg_sq'])
        self.opt_step, self.lr_scheduler.epoch = int(state_dict['step']), int(state_dict['epoch'])
