"""Base class of the meta-learners: seeding, normalisation, evaluation metrics, confidence intervals.
Mirrors RegressionModelMetaLearned (meta_learn/abstract.py:117-272) -- host-side plumbing only."""
import math

import numpy as np
import torch

from . import _lib as L
from .config import get_device
from .util import _handle_input_dimensionality, get_logger


EVAL_COV_BYTES = 1 << 30       # predictive covariances held at once by eval_datasets


class RegressionModelMetaLearned:

    def __init__(self, normalize_data=True, random_seed=None):
        self.normalize_data = normalize_data
        self.logger = get_logger()
        self.input_dim = None
        self.output_dim = None
        self.device = get_device()
        self.dtype = torch.float32
        from . import parallel
        if random_seed is None:
            random_seed = parallel.broadcast_seed(None)   # several ranks must draw the same task batches: agree on rank 0's seed
        if random_seed is not None:                       # abstract.py:125-129
            torch.manual_seed(random_seed)
            self.rds_numpy = np.random.RandomState(random_seed + 1)
        else:
            self.rds_numpy = np.random

    def predict(self, context_x, context_y, test_x, **kwargs):
        raise NotImplementedError

    def eval(self, context_x, context_y, test_x, test_y, flatten_y=True, **kwargs):
        """abstract.py:134-163 -> (avg joint log-likelihood per test point, rmse, calibration error)"""
        context_x, context_y = _handle_input_dimensionality(context_x, context_y)
        test_x, test_y = _handle_input_dimensionality(test_x, test_y)
        test_y_tensor = torch.from_numpy(test_y).float().flatten().to(self.device)
        pred_dist = self.predict(context_x, context_y, test_x, return_density=True, **kwargs)
        avg_log_likelihood = torch.mean(pred_dist.log_prob(test_y_tensor) / test_y_tensor.shape[0])
        rmse = torch.mean(torch.pow(pred_dist.mean - test_y_tensor, 2)).sqrt()
        calibr_error = _calib_error(pred_dist, test_y_tensor)
        return avg_log_likelihood.cpu().item(), rmse.cpu().item(), calibr_error.cpu().item()

    def eval_datasets(self, test_tuples, flatten_y=True, **kwargs):
        """abstract.py:165-181: mean over the test tasks of eval()'s three metrics.  Learners whose predictive parameters do not
        depend on the task (_eval_params: MAP -> the parameter row, SVGD -> the particles, VI -> posterior samples drawn per task in
        task order) evaluate all test tasks of equal shape in ONE batched pass over T*P posterior GPs (this runs every log_period
        inside meta_fit); others loop over eval()."""
        assert (all([len(valid_tuple) == 4 for valid_tuple in test_tuples]))
        params = self._eval_params(**kwargs) if flatten_y else None
        if params is None:
            ll_list, rmse_list, calibr_err_list = list(zip(*[self.eval(*t, flatten_y=flatten_y, **kwargs) for t in test_tuples]))
            return np.mean(ll_list), np.mean(rmse_list), np.mean(calibr_err_list)
        theta, mixture, per_task = params
        tuples, groups = [], {}
        for i, (cx, cy, tx, ty) in enumerate(test_tuples):
            cx, cy = _handle_input_dimensionality(cx, cy)
            tx, ty = _handle_input_dimensionality(tx, ty)
            assert tx.shape[1] == cx.shape[1]
            tuples.append((cx, cy, tx, ty))
            groups.setdefault((cx.shape[0], tx.shape[0]), []).append(i)
        if per_task:                                      # a callable: T*P parameter rows, P fresh ones per task, drawn in task order
            theta = theta(len(tuples))
            P = theta.shape[0] // len(tuples)
            theta = theta.view(len(tuples), P, -1)
        else:
            P = theta.shape[0]
        metrics = np.empty((len(tuples), 3), dtype=np.float64)
        for (_, m), members in groups.items():
            per_pass = max(1, int(EVAL_COV_BYTES // (P * m * m * 4)))        # bound the [T*P, m, m] covariances
            for lo in range(0, len(members), per_pass):
                ids = members[lo:lo + per_pass]
                rows = theta[ids].reshape(len(ids) * P, -1) if per_task else theta
                metrics[ids] = self._eval_tasks(rows, mixture, [tuples[i] for i in ids], per_task)
        return np.mean(metrics[:, 0]), np.mean(metrics[:, 1]), np.mean(metrics[:, 2])

    def _eval_params(self, **kwargs):
        """what predict(**kwargs) conditions on, for the batched pass: (theta[P,D], mixture, False) when every task uses the same
        parameter rows; (callable T -> theta[T*P,D], mixture, True) when each predict() call draws its own; None = loop over eval()"""
        return None

    def _eval_tasks(self, theta, mixture, tuples, per_task=False):
        """eval() of T equally shaped test tasks in one pass -> float64 [T, 3] (avg joint log-likelihood, rmse, calibration error)"""
        T = len(tuples)
        P = theta.shape[0] // T if per_task else theta.shape[0]
        ctx = [self._prepare_data_per_task(cx, cy) for cx, cy, _, _ in tuples]
        cx = self._to_device(np.stack([c[0] for c in ctx]))
        cy = self._to_device(np.stack([c[1] for c in ctx]))
        tx = self._to_device(np.stack([self._normalize_data(X=t[2], Y=None).astype(np.float32) for t in tuples]))
        ty = torch.from_numpy(np.stack([t[3].flatten() for t in tuples])).float().to(self.dtype).to(self.device)      # [T,m]
        m = ty.shape[1]
        mu, var, cov, _ = self.engine.predict_tasks(theta, cx, cy, tx, want_cov=True, theta_per_task=per_task)
        y_mean, y_std = float(self.y_mean.reshape(-1)[0]), float(self.y_std.reshape(-1)[0])
        mu3, var3 = mu.view(T, P, m), var.view(T, P, m)
        resid = ((ty - y_mean) / y_std).unsqueeze(1) - mu3
        logp, _, _ = L.mvn_logprob_dense(cov, resid.reshape(T * P, m).contiguous(), 1.0)         # cov is consumed
        logp = logp.view(T, P) - m * math.log(y_std)
        ll = torch.logsumexp(logp, dim=1) - math.log(P) if mixture else logp[:, 0]
        mean = (mu3 * y_std + y_mean).mean(1) if mixture else mu3[:, 0] * y_std + y_mean
        rmse = torch.mean(torch.pow(mean - ty, 2), dim=1).sqrt()
        calib = L.calib_error(L.mixture_cdf(mu3, var3, ty, y_mean, y_std))
        return torch.stack([ll / m, rmse, calib], dim=1).double().cpu().numpy()

    def confidence_intervals(self, context_x, context_y, test_x, confidence=0.9, **kwargs):
        """abstract.py:183-204 -> (ucb, lcb)"""
        pred_dist = self.predict(context_x, context_y, test_x, return_density=True, **kwargs)
        alpha = (1 - confidence) / 2
        m = _handle_input_dimensionality(np.asarray(test_x)).shape[0]
        ucb = pred_dist.icdf(torch.ones(m) * (1 - alpha))
        lcb = pred_dist.icdf(torch.ones(m) * alpha)
        return ucb.cpu(), lcb.cpu()

    # -- normalisation: pooled z-scoring over all meta-training points (contract of abstract.py:212-258) --------------------
    def _compute_normalization_stats(self, meta_train_tuples):
        """x_mean/x_std [d], y_mean/y_std [1] over the points of ALL tasks (population std + 1e-8); identity if not normalising"""
        pooled = [np.concatenate(cols, axis=0) for cols in zip(*(_handle_input_dimensionality(x, y) for x, y in meta_train_tuples))]
        stats = []
        for a in pooled:
            if self.normalize_data:
                stats.append((a.mean(axis=0), a.std(axis=0) + 1e-8))
            else:
                stats.append((np.zeros(a.shape[1]), np.ones(a.shape[1])))
        (self.x_mean, self.x_std), (self.y_mean, self.y_std) = stats

    def _normalize_data(self, X, Y=None):
        if not hasattr(self, 'x_std'):
            raise AssertionError('requires computing normalization stats beforehand')
        Xn = (X - self.x_mean) / self.x_std
        return Xn if Y is None else (Xn, (Y - self.y_mean) / self.y_std)

    def _check_meta_data_shapes(self, meta_train_data):
        """every task as 2-D (x, y) in place; all tasks share the input / output widths, which are recorded"""
        meta_train_data[:] = [_handle_input_dimensionality(x, y) for x, y in meta_train_data]
        widths = {(x.shape[-1], y.shape[-1]) for x, y in meta_train_data}
        assert len(widths) == 1, 'tasks disagree on the input / output dimensionality: %s' % sorted(widths)
        (self.input_dim, self.output_dim), = widths

    def _prepare_data_per_task(self, x_data, y_data, flatten_y=True):
        """numpy in, normalised float32 numpy out (the device copy is made by TaskBatch / predict)"""
        x_data, y_data = _handle_input_dimensionality(x_data, y_data)
        x_data, y_data = self._normalize_data(x_data, y_data)
        if flatten_y:
            assert y_data.shape[1] == 1
            y_data = y_data.flatten()
        return x_data.astype(np.float32), y_data.astype(np.float32)

    def _to_device(self, arr):
        return torch.from_numpy(np.ascontiguousarray(arr)).to(self.dtype).to(self.device)

    def _prepare_predict(self, context_x, context_y, test_x):
        context_x, context_y = _handle_input_dimensionality(context_x, context_y)
        test_x = _handle_input_dimensionality(test_x)
        assert test_x.shape[1] == context_x.shape[1]
        cx, cy = self._prepare_data_per_task(context_x, context_y)
        tx = self._normalize_data(X=test_x, Y=None).astype(np.float32)
        return self._to_device(cx), self._to_device(cy), self._to_device(tx)


def _calib_error(pred_dist, test_t_tensor):
    """abstract.py:260-272 on the vectorised (marginal) predictive"""
    return L.calib_error(pred_dist.cdf(test_t_tensor))
