"""PACOH-VI on MI355X: API of GPRegressionMetaLearnedVI (meta_learn/GPR_meta_vi.py:14-262) with the
diagonal or full-covariance Gaussian variational hyper-posterior (random_gp.py:224-251).  The reparameterised
ELBO gradient needs only the per-sample score from the device engine:
  diag:  theta_s = loc + exp(scale) * eps_s          full:  theta_s = loc + tril(tril_cov) eps_s
  elbo_s  = log p(theta_s) - prior_factor * log q(theta_s),   loss = -mean_s elbo_s
  dloss/dloc = -mean_s score_s;   dloss/dscale = -mean_s (score_s * exp(scale) * eps_s + prior_factor)
  dloss/dL_ij = -mean_s score_si eps_sj - [i == j] prior_factor / L_ii   (j <= i; zero above the diagonal)."""
import math
import time

import torch

from . import _lib as L
from .engine import AsyncUploader
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


def standard_normal(n, D):
    """the eps of Normal(loc, scale).rsample((n,)) (torch.distributions: _standard_normal)"""
    return torch.normal(torch.zeros(n, D), torch.ones(n, D))


class GPRegressionMetaLearnedVI(_RandomGPLearner):

    def __init__(self, meta_train_data, num_iter_fit=10000, feature_dim=1,
                 prior_factor=0.01, weight_prior_std=0.5, bias_prior_std=3.0,
                 covar_module='NN', mean_module='NN', mean_nn_layers=(32, 32), kernel_nn_layers=(32, 32),
                 optimizer='Adam', lr=1e-3, lr_decay=1.0, svi_batch_size=10, cov_type='diag',
                 task_batch_size=-1, normalize_data=True, random_seed=None):
        """Arguments as in the reference (GPR_meta_vi.py:16-44)."""
        super().__init__(normalize_data, random_seed)
        assert mean_module in ['NN', 'constant', 'zero'] and covar_module in ['NN', 'SE']
        assert optimizer in ['Adam', 'SGD']
        assert cov_type in ['diag', 'full']
        self.cov_type = cov_type
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

    def meta_fit(self, valid_tuples=None, verbose=True, log_period=500, n_iter=None):
        """GPR_meta_vi.py:84-128"""
        assert (valid_tuples is None) or (all([len(valid_tuple) == 4 for valid_tuple in valid_tuples]))
        t = time.time()
        if n_iter is None:
            n_iter = self.num_iter_fit
        loss = None
        for itr in range(1, n_iter + 1):
            idx_local, pre = self._sample_task_batch()
            loss, grad = self.get_neg_elbo_and_grad(idx_local, pre)
            self.opt_step += 1
            if self.optimizer_name == 'Adam':
                L.adam_step(self.posterior, grad, self.exp_avg, self.exp_avg_sq, self.lr_scheduler.lr, self.opt_step)
            else:
                L.axpy(self.posterior, grad, -self.lr_scheduler.lr)
            self.lr_scheduler.step()
            if itr == 1 or itr % log_period == 0:
                duration = time.time() - t
                t = time.time()
                message = 'Iter %d/%d - Loss: %.6f - Time %.2f sec' % (itr, self.num_iter_fit, loss.item(), duration)
                if valid_tuples is not None:
                    valid_ll, valid_rmse, calibr_err = self.eval_datasets(valid_tuples)
                    message += ' - Valid-LL: %.3f - Valid-RMSE: %.3f - Calib-Err %.3f' % (valid_ll, valid_rmse, calibr_err)
                if verbose:
                    self.logger.info(message)
        self.fitted = True
        return loss.item() if loss is not None else float('nan')

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
            eps = torch.cat([standard_normal(n_posterior_samples, self.layout.D) for _ in range(T)])
            theta, _ = L.vi_sample(self.posterior, eps.to(self.dtype).to(self.device), full=self.cov_type == 'full')
            return theta
        return draw, True, True

    def state_dict(self):
        return {'posterior': self.posterior.cpu().clone(), 'exp_avg': self.exp_avg.cpu().clone(),
                'exp_avg_sq': self.exp_avg_sq.cpu().clone(), 'step': self.opt_step, 'epoch': self.lr_scheduler.epoch}

    def load_state_dict(self, sd):
        self.posterior.copy_(sd['posterior']); self.exp_avg.copy_(sd['exp_avg']); self.exp_avg_sq.copy_(sd['exp_avg_sq'])
        self.opt_step, self.lr_scheduler.epoch = int(sd['step']), int(sd['epoch'])
