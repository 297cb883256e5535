"""Single-task GP regression with a learnable mean / kernel (no meta-learning) on MI355X: API of the reference's
GPRegressionLearned (meta_learn/GPR_mll.py:11-216, base class RegressionModel meta_learn/abstract.py:7-115).
It is the T = 1 special case of the PACOH-MAP path and runs on the same kernels."""
import time

import numpy as np
import torch

from . import _lib as L
from .abstract import _calib_error
from .config import get_device
from .distributions import GaussianPredictive
from .engine import GPEngine, NotPSDError, ParamLayout, TaskBatch
from .modules import apply_initial_values, resolve_covar_module, resolve_mean_module
from .util import _handle_input_dimensionality, get_logger


class _ReduceLROnPlateau:
    """torch.optim.lr_scheduler.ReduceLROnPlateau(mode='max', factor, patience=10, threshold=1e-4 'rel')
    as used at GPR_mll.py:104-107 (stepped with the validation log-likelihood at every log line)"""

    def __init__(self, lr, factor, patience=10, threshold=1e-4):
        self.lr, self.factor, self.patience, self.threshold = lr, factor, patience, threshold
        self.best, self.num_bad = -float('inf'), 0

    def step(self, metric):
        # torch's rule for mode='max', threshold_mode='rel' is `metric > best * (1 + threshold)` whatever the sign of best
        if metric > self.best * (1.0 + self.threshold):
            self.best, self.num_bad = metric, 0
        else:
            self.num_bad += 1
        if self.num_bad > self.patience:
            self.lr *= self.factor
            self.num_bad = 0


class GPRegressionLearned:

    def __init__(self, train_x, train_t, learning_mode='both', lr=1e-3, weight_decay=0.0, feature_dim=2,
                 num_iter_fit=1000, covar_module='NN', mean_module='NN', mean_nn_layers=(32, 32), kernel_nn_layers=(32, 32),
                 optimizer='Adam', normalize_data=True, lr_scheduler=True, random_seed=None):
        """Arguments as in the reference (GPR_mll.py:13-36)."""
        assert learning_mode in ['learn_mean', 'learn_kernel', 'both', 'vanilla']
        # strings as the reference, or ZeroMean / ConstantMean / (Scale)RBFKernel objects (modules.py); other objects cannot run here
        mean_module, mean_init = resolve_mean_module(mean_module)
        covar_module, covar_init, learn_os = resolve_covar_module(covar_module)
        assert mean_module in ['NN', 'constant', 'zero'] and covar_module in ['NN', 'SE', 'COS']
        assert optimizer in ['Adam', 'SGD']
        self.normalize_data, self.logger = normalize_data, get_logger()
        self.device, self.dtype = get_device(), torch.float32
        if random_seed is not None:                           # RegressionModel.__init__, abstract.py:18-20
            torch.manual_seed(random_seed)
            np.random.seed(random_seed + 1)
        self.lr, self.weight_decay, self.num_iter_fit, self.optimizer_name = lr, weight_decay, num_iter_fit, optimizer

        # ---- data handling: normalisation statistics of THIS dataset (abstract.py:96-110) ----
        train_x, train_t = _handle_input_dimensionality(np.asarray(train_x), np.asarray(train_t))
        self.input_dim, self.output_dim = train_x.shape[-1], train_t.shape[-1]
        assert self.output_dim == 1
        self.n_train_samples = train_x.shape[0]
        if normalize_data:
            self.x_mean, self.y_mean = np.mean(train_x, axis=0), np.mean(train_t, axis=0)
            self.x_std, self.y_std = np.std(train_x, axis=0) + 1e-8, np.std(train_t, axis=0) + 1e-8
        else:
            self.x_mean, self.y_mean = np.zeros(train_x.shape[1]), np.zeros(train_t.shape[1])
            self.x_std, self.y_std = np.ones(train_x.shape[1]), np.ones(train_t.shape[1])
        xn = ((train_x - self.x_mean[None, :]) / self.x_std[None, :]).astype(np.float32)
        tn = ((train_t - self.y_mean[None, :]) / self.y_std[None, :]).astype(np.float32).flatten()
        self.task = TaskBatch([(xn, tn)], self.device, self.dtype)
        self._ctx_x, self._ctx_y = self.task.x[0], self.task.y[0]

        # ---- model: shared flat parameter vector, same RNG order as the reference (kernel net, then mean net) ----
        if covar_module == 'NN':
            assert learning_mode in ['learn_kernel', 'both'], 'neural network parameters must be learned'
        if mean_module == 'NN':
            assert learning_mode in ['learn_mean', 'both'], 'neural network parameters must be learned'
        self.layout = lay = ParamLayout(self.input_dim, mean_module, covar_module, mean_nn_layers, kernel_nn_layers,
                                        feature_dim, with_outputscale=True)
        theta = torch.zeros(lay.D)

        def init_net(prefix, out_dim, layers):
            prev = self.input_dim
            names = ['fc_%i' % (i + 1) for i in range(len(layers))] + ['out']
            for name, size in zip(names, list(layers) + [out_dim]):
                lin = torch.nn.Linear(prev, size)
                lo, hi = lay.slices['%s.%s.bias' % (prefix, name)]
                theta[lo:hi] = lin.bias.detach()
                lo, hi = lay.slices['%s.%s.weight' % (prefix, name)]
                theta[lo:hi] = lin.weight.detach().reshape(-1)
                prev = size
        if covar_module == 'NN':
            init_net('kernel_nn', feature_dim, kernel_nn_layers)
        if mean_module == 'NN':
            init_net('mean_nn', 1, mean_nn_layers)
        apply_initial_values(theta, lay, dict(mean_init, **covar_init))
        self.theta = theta.reshape(1, -1).to(self.dtype).to(self.device)
        # GaussianLikelihood() default noise constraint is GreaterThan(1e-4) [gpytorch-upstream]
        self.engine = GPEngine(lay, noise_floor=1e-4)

        # optimiser groups: NN groups carry `weight_decay`, all other groups torch.optim.AdamW's DEFAULT 1e-2
        # (GPR_mll.py:57,69,82,92-98: AdamW(self.parameters) without a global weight_decay)
        segs = []
        if covar_module == 'NN':
            segs.append(lay.block_range('kernel_nn.') + (weight_decay,))
        if mean_module == 'NN':
            segs.append(lay.block_range('mean_nn.') + (weight_decay,))
        segs.append(lay.slices['noise_raw'] + (1e-2,))
        if learning_mode in ('learn_kernel', 'both'):
            segs.append(lay.slices['lengthscale_raw'] + (1e-2,))
            if learn_os:
                segs.append(lay.slices['outputscale_raw'] + (1e-2,))
        if learning_mode in ('learn_mean', 'both') and mean_module == 'constant':
            segs.append(lay.slices['constant_mean'] + (1e-2,))
        self.train_segments = segs
        self.exp_avg, self.exp_avg_sq, self.opt_step = torch.zeros_like(self.theta), torch.zeros_like(self.theta), 0
        self.lr_scheduler = _ReduceLROnPlateau(lr, 0.2 if lr_scheduler else 1.0)
        self.fitted = False

    # ------------------------------------------------------------------------------------------------
    def fit(self, valid_x=None, valid_t=None, verbose=True, log_period=500, n_iter=None):
        """GPR_mll.py:111-168: maximise the per-datapoint marginal log-likelihood of the training set"""
        assert (valid_x is None and valid_t is None) or (isinstance(valid_x, np.ndarray) and isinstance(valid_t, np.ndarray))
        t = time.time()
        n_iter = self.num_iter_fit if n_iter is None else n_iter
        loss = None
        fail = torch.zeros(1, dtype=torch.int32, device=self.device)
        for itr in range(1, n_iter + 1):
            lml, grad, _ = self.engine.lml_and_grad(self.theta, self.task, weight=-1.0, fail_flag=fail)
            loss = torch.empty((), dtype=self.dtype, device=self.device)
            L.reduce_tasks(lml.reshape(-1, 1, 1), loss.reshape(1, 1), scale=-1.0)
            self.opt_step += 1
            for lo, hi, wd in self.train_segments:
                p, g = self.theta[0, lo:hi], grad[0, lo:hi]
                if self.optimizer_name == 'Adam':
                    L.adam_step(p, g, self.exp_avg[0, lo:hi], self.exp_avg_sq[0, lo:hi], self.lr_scheduler.lr, self.opt_step,
                                weight_decay=wd)
                else:
                    L.axpy(p, g, -self.lr_scheduler.lr)
            if itr == 1 or itr % log_period == 0:
                duration = time.time() - t
                t = time.time()
                message = 'Iter %d/%d - Loss: %.3f - Time %.3f sec' % (itr, self.num_iter_fit, loss.item(), duration)
                if int(fail.item()) != 0:                 # gpytorch raises inside the loss evaluation (psd_safe_cholesky)
                    raise NotPSDError('the kernel matrix was not positive definite even after adding jitter (1e-6 .. 1e-4)')
                if valid_x is not None:
                    valid_ll, valid_rmse, calibr_err = self.eval(valid_x, valid_t)
                    self.lr_scheduler.step(valid_ll)
                    message += ' - Valid-LL: %.3f - Valid-RMSE: %.3f - Calib-Err %.3f' % (valid_ll, valid_rmse, calibr_err)
                self._last_log = message
                if verbose:
                    self.logger.info(message)
        self.fitted = True
        return loss.item() if loss is not None else float('nan')

    def predict(self, test_x, return_density=False, **kwargs):
        """GPR_mll.py:170-195: p(t | test_x, train_x, train_t)"""
        test_x = np.asarray(test_x)
        if test_x.ndim == 1:
            test_x = np.expand_dims(test_x, axis=-1)
        tx = ((test_x - self.x_mean[None, :]) / self.x_std[None, :]).astype(np.float32)
        tx = torch.from_numpy(np.ascontiguousarray(tx)).to(self.device)
        mu, var, cov, _ = self.engine.predict(self.theta, self._ctx_x[:self.n_train_samples],
                                              self._ctx_y[:self.n_train_samples], tx, want_cov=return_density)
        dist = GaussianPredictive(mu, var, cov, self.y_mean.reshape(-1)[0], self.y_std.reshape(-1)[0], mixture=False)
        if return_density:
            return dist
        return dist.mean.cpu().numpy(), dist.stddev.cpu().numpy()

    def eval(self, test_x, test_t, **kwargs):
        """abstract.py:25-48 -> (avg joint log-likelihood per test point, rmse, calibration error)"""
        test_x, test_t = _handle_input_dimensionality(np.asarray(test_x), np.asarray(test_t))
        ty = torch.from_numpy(test_t).contiguous().float().flatten().to(self.device)
        pred = self.predict(test_x, return_density=True)
        avg_ll = pred.log_prob(ty) / ty.shape[0]
        rmse = torch.mean(torch.pow(pred.mean - ty, 2)).sqrt()
        return avg_ll.cpu().item(), rmse.cpu().item(), _calib_error(pred, ty).cpu().item()

    def confidence_intervals(self, test_x, confidence=0.9, **kwargs):
        """abstract.py:50-57 -> (ucb, lcb)"""
        pred = self.predict(test_x, return_density=True)
        alpha = (1 - confidence) / 2
        m = np.asarray(test_x).shape[0]
        return pred.icdf(torch.ones(m) * (1 - alpha)).cpu(), pred.icdf(torch.ones(m) * alpha).cpu()

    def state_dict(self):
        return {'model': {k: self.theta[0, lo:hi].cpu().clone() for k, (lo, hi) in self.layout.slices.items()},
                'optimizer': {'exp_avg': self.exp_avg.cpu().clone(), 'exp_avg_sq': self.exp_avg_sq.cpu().clone(),
                              'step': self.opt_step, 'lr': self.lr_scheduler.lr}}

    def load_state_dict(self, state_dict):
        for k, (lo, hi) in self.layout.slices.items():
            self.theta[0, lo:hi] = state_dict['model'][k].to(self.dtype).to(self.device)
        o = state_dict['optimizer']
        self.exp_avg.copy_(o['exp_avg']); self.exp_avg_sq.copy_(o['exp_avg_sq'])
        self.opt_step, self.lr_scheduler.lr = int(o['step']), float(o['lr'])
