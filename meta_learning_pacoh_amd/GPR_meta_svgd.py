"""PACOH-SVGD on MI355X: API of GPRegressionMetaLearnedSVGD (meta_learn/GPR_meta_svgd.py:14-230).
One svgd_step = [MLP fwd x2] -> fused GP LML fwd+bwd over tasks x particles -> [MLP bwd x2] ->
hyper-prior grad -> SVGD phi -> Adam, all HIP kernels on one stream; with torch.distributed
initialised, tasks are sharded over ranks and the score is all-reduced once per step (RCCL)."""
import time

import numpy as np
import torch

from . import _lib as L
from . import parallel
from .abstract import RegressionModelMetaLearned
from .distributions import GaussianPredictive
from .engine import AsyncUploader, GPEngine, ParamLayout, TaskBatch
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

    def _sample_task_batch(self):
        """global with-replacement draw from the shared seed (GPR_meta_svgd.py:102), then this rank's shard"""
        idx = self.rds_numpy.randint(0, self.tasks.T, size=self.task_batch_size)
        pre = harmonic_pre_factor(self.tasks.sizes[idx])
        local = parallel.shard(idx)
        return local, pre

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
        assert num_particles <= 64, 'pacoh_svgd_phi supports up to 64 particles'
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
        self._setup_tasks(meta_train_data)
        self.fitted = False

    def svgd_step(self, idx_local, pre_factor):
        """SVGD.step (meta_learn/svgd.py:25-28): particles.grad = -phi; optimizer.step()"""
        if self.kernel == 'RBF':
            # prior score + phi + optimizer step in one kernel (three launches with the distance / bandwidth kernels)
            _, score = self._log_prob_and_score(self.particles, idx_local, pre_factor, with_prior=False)
            self.opt_step += 1
            self.particles, self.last_bandwidth, self._svgd_ws = L.svgd_update(
                self.particles, score, self.prior_mean, self.prior_std, self.prior_factor, self.bandwidth, self.optimizer_name,
                self.lr_scheduler.lr, self.opt_step, self.exp_avg, self.exp_avg_sq, workspace=self._svgd_ws)
            return
        _, score = self._log_prob_and_score(self.particles, idx_local, pre_factor)
        phi_fn = L.svgd_phi_imq                                             # IMQ: alpha=0.5, beta=-0.5 (svgd.py:70)
        neg_phi, self.last_bandwidth, self._svgd_ws = phi_fn(self.particles, score, bandwidth=self.bandwidth, neg=True,
                                                             workspace=self._svgd_ws)
        self.opt_step += 1
        if self.optimizer_name == 'Adam':
            L.adam_step(self.particles, neg_phi, self.exp_avg, self.exp_avg_sq, self.lr_scheduler.lr, self.opt_step)
        else:
            L.axpy(self.particles, neg_phi, -self.lr_scheduler.lr)

    def meta_fit(self, valid_tuples=None, verbose=True, log_period=500, n_iter=None):
        """GPR_meta_svgd.py:82-121"""
        assert (valid_tuples is None) or (all([len(valid_tuple) == 4 for valid_tuple in valid_tuples]))
        t = time.time()
        if n_iter is None:
            n_iter = self.num_iter_fit
        for itr in range(1, n_iter + 1):
            idx_local, pre = self._sample_task_batch()
            self.svgd_step(idx_local, pre)
            self.lr_scheduler.step()
            if itr == 1 or itr % log_period == 0:
                torch.cuda.synchronize()
                duration = time.time() - t
                t = time.time()
                message = 'Iter %d/%d - Time %.2f sec' % (itr, self.num_iter_fit, duration)
                if valid_tuples is not None:
                    valid_ll, valid_rmse, calibr_err = self.eval_datasets(valid_tuples)
                    message += ' - Valid-LL: %.3f - Valid-RMSE: %.3f - Calib-Err %.3f' % (valid_ll, valid_rmse, calibr_err)
                if verbose:
                    self.logger.info(message)
        self.fitted = True

    def predict(self, context_x, context_y, test_x, return_density=False):
        """GPR_meta_svgd.py:123-159: equal-weighted mixture over the particles' GP posteriors"""
        return self._mixture_predict(self.particles, context_x, context_y, test_x, return_density)

    def _eval_params(self, **kwargs):
        return (self.particles, True, False) if not kwargs else None

    def state_dict(self):
        return {'particles': self.particles.cpu().clone(), 'exp_avg': self.exp_avg.cpu().clone(),
                'exp_avg_sq': self.exp_avg_sq.cpu().clone(), 'step': self.opt_step, 'epoch': self.lr_scheduler.epoch}

    def load_state_dict(self, sd):
        self.particles.copy_(sd['particles']); self.exp_avg.copy_(sd['exp_avg']); self.exp_avg_sq.copy_(sd['exp_avg_sq'])
        self.opt_step, self.lr_scheduler.epoch = int(sd['step']), int(sd['epoch'])
