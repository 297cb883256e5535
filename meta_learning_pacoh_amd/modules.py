"""`mean_module` / `covar_module` arguments given as OBJECTS.  The reference accepts gpytorch.means.Mean / gpytorch.kernels.Kernel
instances besides the strings (meta_learn/GPR_meta_mll.py:207-251, GPR_mll.py:45-90) and hands them to ExactGP as they are.  The
HIP path implements two kernel families -- ARD-RBF on raw inputs or on learned features, and (round 3) the cosine kernel on raw
inputs, both optionally scaled -- and zero / constant / network means, so an object is accepted exactly when it denotes one of
those: ZeroMean, ConstantMean, RBFKernel, CosineKernel (the reference's own tests/test_GPR.py:95-101 hands one to the single-task
learner) and ScaleKernel(...) of either, recognised by class name anywhere in the object's MRO (gpytorch itself is not needed, and
not present, on the GPU box).  Their current raw hyper-parameters become the initial values.  Anything else raises
NotImplementedError: it cannot be evaluated by these kernels."""
import math


def _names(obj):
    return {c.__name__ for c in type(obj).__mro__}


def _scalar(v):
    if v is None:
        return None
    if hasattr(v, 'detach'):
        v = v.detach().reshape(-1)
        return float(v[0]) if v.numel() > 0 else None
    try:
        return float(v)
    except (TypeError, ValueError):
        return None


def _vector(v):
    if v is None:
        return None
    if hasattr(v, 'detach'):
        return [float(t) for t in v.detach().reshape(-1)]
    return None


def resolve_mean_module(mean_module):
    """-> (one of 'NN' | 'constant' | 'zero', initial values {'constant_mean': c} or {})"""
    if isinstance(mean_module, str):
        return mean_module, {}
    names = _names(mean_module)
    if 'ZeroMean' in names:
        return 'zero', {}
    if 'ConstantMean' in names:
        c = _scalar(getattr(mean_module, 'constant', None))
        if c is None:
            c = _scalar(getattr(mean_module, 'raw_constant', None))
        return 'constant', ({} if c is None else {'constant_mean': c})
    raise NotImplementedError('mean_module object of type %s: the HIP path evaluates zero, constant and neural-network means only'
                              % type(mean_module).__name__)


def resolve_covar_module(covar_module):
    """-> ('NN' | 'SE' | 'COS', initial raw values {'lengthscale_raw': [...], 'outputscale_raw': v}, learn_outputscale)"""
    if isinstance(covar_module, str):
        return covar_module, {}, True
    names = _names(covar_module)
    base, scaled = covar_module, False
    if 'ScaleKernel' in names:
        base, scaled = getattr(covar_module, 'base_kernel', None), True
    base_names = _names(base) if base is not None else set()
    if 'RBFKernel' in base_names:
        kind, raw = 'SE', 'raw_lengthscale'
    elif 'CosineKernel' in base_names:                         # k = cos(pi |x - x'| / period_length): ONE raw period parameter
        kind, raw = 'COS', 'raw_period_length'
    else:
        raise NotImplementedError('covar_module object of type %s: the HIP path evaluates the (scaled) ARD-RBF and cosine kernels only'
                                  % type(covar_module).__name__)
    init = {}
    ls = _vector(getattr(base, raw, None))
    if ls:
        init['lengthscale_raw'] = ls[:1] if kind == 'COS' else ls
    if scaled:
        os_ = _scalar(getattr(covar_module, 'raw_outputscale', None))
        if os_ is not None:
            init['outputscale_raw'] = os_
    else:
        init['outputscale_raw'] = math.log(math.e - 1.0)          # softplus^-1(1): a plain kernel has unit output scale, not learned
    return kind, init, scaled


def apply_initial_values(theta, layout, init):
    """write the initial raw values taken from module objects into the flat parameter vector theta[D] (host tensor)"""
    for name, val in init.items():
        if name not in layout.slices:
            continue
        lo, hi = layout.slices[name]
        if isinstance(val, (list, tuple)):
            if len(val) == 1:
                val = val * (hi - lo)
            assert len(val) == hi - lo, '%s: %d initial values for %d parameters' % (name, len(val), hi - lo)
            for k, v in enumerate(val):
                theta[lo + k] = v
        else:
            theta[lo:hi] = val
