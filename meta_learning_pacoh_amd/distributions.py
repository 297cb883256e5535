"""Predictive distribution objects returned by predict(..., return_density=True).

They play the role of AffineTransformedDistribution(MultivariateNormal) (meta_learn/models.py:15-43) and
EqualWeightedMixtureDist(batched=True) (models.py:74-140) in the reference: .mean, .stddev, .variance,
.log_prob (JOINT Gaussian log-density over all test points, per component), .cdf / .icdf (marginals).
The joint log-density runs the dense HIP Cholesky kernel on the predictive covariance."""
import math

import torch

from . import _lib as L


class GaussianPredictive:
    """P Gaussian components over m test points, held in NORMALISED space with the affine
    un-normalisation y = y_mean + y_std * y_n applied on access.  P == 1 & mixture=False is the MAP case."""

    def __init__(self, mu_n, var_n, cov_n, y_mean, y_std, mixture):
        self._mu_n, self._var_n, self._cov_n = mu_n, var_n, cov_n        # [P,m], [P,m], [P,m,m]
        self.y_mean, self.y_std = float(y_mean), float(y_std)
        self.mixture = mixture
        self.num_dists = mu_n.shape[0]

    # -- component moments in original units ------------------------------------------------------
    @property
    def _means(self):
        return self._mu_n * self.y_std + self.y_mean

    @property
    def _vars(self):
        return self._var_n * self.y_std ** 2

    @property
    def mean(self):
        m = self._means
        return m.mean(0) if self.mixture else m[0]

    @property
    def variance(self):
        if not self.mixture:
            return self._vars[0]
        means = self._means
        return ((means - means.mean(0)) ** 2).mean(0) + self._vars.mean(0)      # models.py:101-115

    @property
    def stddev(self):
        return torch.sqrt(self.variance)

    # -- densities --------------------------------------------------------------------------------
    def log_prob(self, value):
        """joint log-density of the m test targets (original units); mixture: logsumexp - log P"""
        if self._cov_n is None:
            raise RuntimeError('predict(..., return_density=True) must be called to get the joint covariance')
        value = torch.as_tensor(value, dtype=self._mu_n.dtype, device=self._mu_n.device).flatten()
        m = value.shape[0]
        resid = ((value - self.y_mean) / self.y_std).unsqueeze(0) - self._mu_n              # [P,m]
        logp, _, _ = L.mvn_logprob_dense(self._cov_n.clone(), resid.contiguous(), 1.0)
        logp = logp - m * math.log(self.y_std)
        if not self.mixture:
            return logp[0]
        return torch.logsumexp(logp, dim=0, keepdim=True) - math.log(self.num_dists)

    def cdf(self, value):
        """marginal cdf per test point (mixture: mean over components, models.py:124-131)"""
        value = torch.as_tensor(value, dtype=self._mu_n.dtype, device=self._mu_n.device)
        return L.mixture_cdf(self._mu_n.contiguous(), self._var_n.contiguous(), value, self.y_mean, self.y_std)

    def icdf(self, quantile):
        """marginal quantiles per test point: Gaussian closed form for one component, the reference's bisection (models.py:136-140,
        util.py:9-42) for the mixture -- one kernel launch either way"""
        quantile = torch.as_tensor(quantile, dtype=self._mu_n.dtype, device=self._mu_n.device)
        return L.mixture_icdf(self._mu_n.contiguous(), self._var_n.contiguous(), quantile, self.y_mean, self.y_std,
                              closed_form=not self.mixture)
