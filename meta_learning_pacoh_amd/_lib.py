"""ctypes binding of libpacoh_gp.so (include/pacoh_gp.h) for PyTorch-ROCm tensors.

PyTorch is used for device memory, streams and torch.distributed only; every arithmetic step of the
hot path goes through the C ABI below.  There is NO fallback: if the library is missing or a tensor
is not on a HIP device the call raises.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('PACOH_LIB') or os.path.join(_HERE, 'lib', 'libpacoh_gp.so')     # PACOH_LIB: an experimental build (_build.py --variant)

F32, F64 = 0, 1
MEAN_ZERO, MEAN_VECTOR, MEAN_CONST = 0, 1, 2
KERNEL_RBF, KERNEL_COSINE, KERNEL_SHIFT = 0, 1, 8          # PACOH_KERNEL_* (pacoh_gp.h): the family rides in the bits above f


def _kf(f, kernel):
    """the `f` argument of a GP entry point: feature count | kernel family"""
    return int(f) | (int(kernel) << KERNEL_SHIFT)
ERRORS = {-1: 'PACOH_EINVAL (bad argument)', -2: 'PACOH_ELIMIT (shape outside kernel limits)',
          -3: 'PACOH_EDTYPE', -4: 'PACOH_ELAUNCH (HIP launch failed)', -5: 'PACOH_ENOCOMM (librccl not loadable)'}
COMM_ID_BYTES = 128

_c = ctypes
_vp, _i, _l, _d, _sz = _c.c_void_p, _c.c_int, _c.c_long, _c.c_double, _c.c_size_t
_ip = _c.POINTER(_c.c_int32)

# symbol -> (restype, argtypes); must list every function declared in include/pacoh_gp.h
SIGNATURES = {
    'pacoh_abi_version': (_i, []),
    'pacoh_reload_env': (None, []),
    'pacoh_gram_rbf_ard': (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'pacoh_gp_small_max_n': (_i, [_i, _i]),
    'pacoh_gp_lml_fwd': (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    'pacoh_gp_lml_fwdbwd': (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                 _i, _i, _i, _i, _i, _vp]),
    'pacoh_gp_predict_workspace_bytes': (_sz, [_i, _i, _i, _i, _i]),
    'pacoh_gp_predict': (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                              _i, _i, _i, _i, _i, _i, _vp]),
    'pacoh_gp_lml_dense_workspace_bytes': (_sz, [_i, _i, _i, _i, _i]),
    'pacoh_gp_lml_dense': (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz,
                                _i, _i, _i, _i, _i, _vp]),
    'pacoh_gp_predict_dense_workspace_bytes': (_sz, [_i, _i, _i, _i]),
    'pacoh_gp_predict_dense': (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                    _i, _i, _i, _i, _i, _i, _vp]),
    'pacoh_mvn_logprob_dense': (_i, [_vp, _vp, _vp, _vp, _vp, _d, _i, _i, _i, _vp]),
    'pacoh_mlp_fwd_workspace_bytes': (_sz, [_i, _i, _i, _i, _ip, _i, _i, _i]),
    'pacoh_mlp_fwd': (_i, [_vp, _i, _vp, _l, _i, _i, _ip, _i, _i, _vp, _vp, _i, _i, _i, _vp]),
    'pacoh_mlp_bwd_workspace_bytes': (_sz, [_i, _i, _i, _i, _ip, _i, _i, _i]),
    'pacoh_mlp_bwd': (_i, [_vp, _i, _vp, _l, _i, _i, _ip, _i, _i, _vp, _vp, _l, _i, _vp, _i, _i, _i, _vp]),
    'pacoh_mlp_bwd_hyper': (_i, [_vp, _i, _vp, _l, _i, _i, _ip, _i, _i, _vp, _vp, _l, _vp, _vp, _i, _i,
                                 _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _d, _vp, _vp, _vp, _i, _i, _vp, _i, _vp]),
    'pacoh_mlp_stash_bytes': (_sz, [_i, _i, _i, _i, _ip, _i, _i, _i]),
    'pacoh_mlp_fwd_stash': (_i, [_vp, _i, _vp, _l, _i, _i, _ip, _i, _i, _vp, _vp, _vp, _i, _i, _i, _vp]),
    'pacoh_mlp_fused_path': (_i, [_i, _i, _i, _i, _ip, _i, _i, _i]),
    'pacoh_mlp2_fwd_workspace_bytes': (_sz, [_i, _i, _i, _i, _ip, _i, _i, _i, _i]),
    'pacoh_mlp2_stash_bytes': (_sz, [_i, _i, _i, _i, _ip, _i, _i, _i, _i]),
    'pacoh_mlp2_fwd': (_i, [_vp, _i, _vp, _l, _i, _i, _ip, _i, _l, _i, _vp, _l, _i, _vp, _vp, _vp, _i, _i, _i, _vp]),
    'pacoh_mlp2_fwd_svgd': (_i, [_vp, _i, _vp, _l, _i, _i, _ip, _i, _l, _i, _vp, _l, _i, _vp, _vp, _vp, _i, _i,
                                 _vp, _vp, _i, _i, _vp, _i, _vp]),
    'pacoh_mlp2_bwd_workspace_bytes': (_sz, [_i, _i, _i, _i, _ip, _i, _i, _i, _i]),
    'pacoh_mlp2_bwd': (_i, [_vp, _i, _vp, _l, _i, _i, _ip, _i, _l, _i, _vp, _l, _i, _vp, _vp, _l, _i, _vp, _vp, _i, _i, _i, _vp]),
    'pacoh_mlp2_bwd_hyper': (_i, [_vp, _i, _vp, _l, _i, _i, _ip, _i, _l, _i, _vp, _l, _i, _vp, _vp, _l, _i, _vp, _vp, _i, _i,
                                  _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _d, _vp, _vp, _vp, _i, _i, _vp, _i, _vp]),
    'pacoh_softplus_fwd': (_i, [_vp, _vp, _d, _l, _i, _vp]),
    'pacoh_softplus_bwd': (_i, [_vp, _vp, _vp, _i, _l, _i, _vp]),
    'pacoh_hyper_fwd': (_i, [_vp, _l, _i, _i, _i, _i, _i, _d, _vp, _vp, _vp, _i, _vp]),
    'pacoh_hyper_bwd': (_i, [_vp, _l, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _l, _vp, _vp, _d, _vp, _vp, _vp, _i, _i, _vp, _i, _vp]),
    'pacoh_step_select': (_i, [_vp, _i, _vp, _i, _vp, _l, _vp, _vp, _vp, _vp, _i, _vp]),
    'pacoh_scale_dev': (_i, [_vp, _vp, _l, _i, _vp]),
    'pacoh_step_begin': (_i, [_vp, _i, _vp, _i, _vp, _l, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i,
                              _vp, _l, _i, _i, _i, _i, _i, _d, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _vp]),
    'pacoh_step_begin_vi': (_i, [_vp, _i, _vp, _i, _vp, _l, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i,
                                 _vp, _i, _i, _vp, _vp, _i, _i, _i, _i, _d, _vp, _vp, _vp, _i, _i, _vp]),
    'pacoh_svgd_update_dev_workspace_bytes': (_sz, [_i, _i, _i]),
    'pacoh_svgd_update_dev': (_i, [_vp, _vp, _vp, _vp, _d, _d, _i, _vp, _d, _d, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _i, _vp]),
    'pacoh_svgd_dist_advance': (_i, [_vp, _vp, _i, _i, _vp, _i, _vp]),
    'pacoh_svgd_update_next': (_i, [_vp, _vp, _vp, _vp, _d, _d, _i, _d, _d, _vp, _vp, _vp, _vp, _i, _i,
                                    _vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i,
                                    _i, _i, _i, _i, _d, _vp, _vp, _vp, _i, _i, _vp]),
    'pacoh_prior_logprob_grad': (_i, [_vp, _vp, _vp, _vp, _vp, _d, _i, _i, _i, _vp]),
    'pacoh_prior_score_dev': (_i, [_vp, _vp, _vp, _vp, _d, _vp, _i, _i, _i, _vp]),
    'pacoh_svgd_workspace_bytes': (_sz, [_i, _i, _i]),
    'pacoh_svgd_phi': (_i, [_vp, _vp, _d, _i, _vp, _vp, _vp, _i, _i, _i, _vp]),
    'pacoh_svgd_update': (_i, [_vp, _vp, _vp, _vp, _d, _d, _i, _d, _d, _d, _d, _l, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    'pacoh_svgd_imq_workspace_bytes': (_sz, [_i, _i, _i]),
    'pacoh_svgd_phi_imq': (_i, [_vp, _vp, _d, _d, _d, _i, _vp, _vp, _vp, _i, _i, _i, _vp]),
    'pacoh_adam_step': (_i, [_vp, _vp, _vp, _vp, _d, _d, _d, _d, _d, _l, _l, _i, _vp]),
    'pacoh_adam_step_dev': (_i, [_vp, _vp, _vp, _vp, _vp, _d, _d, _l, _vp, _vp, _vp, _i, _vp]),
    'pacoh_vi_update_dev_workspace_bytes': (_sz, [_i, _i]),
    'pacoh_vi_update_dev': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _d, _vp, _d, _d, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    'pacoh_axpy': (_i, [_vp, _vp, _d, _l, _i, _vp]),
    'pacoh_vi_sample': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    'pacoh_vi_grad': (_i, [_vp, _vp, _vp, _d, _vp, _i, _i, _i, _vp]),
    'pacoh_vi_sample_full': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    'pacoh_vi_grad_full': (_i, [_vp, _vp, _vp, _d, _vp, _i, _i, _i, _vp]),
    'pacoh_gather_tasks': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'pacoh_reduce_tasks': (_i, [_vp, _vp, _d, _i, _i, _i, _i, _i, _vp]),
    'pacoh_mixture_cdf': (_i, [_vp, _vp, _vp, _vp, _d, _d, _i, _i, _i, _i, _vp]),
    'pacoh_mixture_icdf': (_i, [_vp, _vp, _vp, _vp, _d, _d, _d, _d, _d, _i, _i, _i, _i, _i, _vp]),
    'pacoh_calib_error': (_i, [_vp, _vp, _i, _i, _i, _vp]),
    'pacoh_map_persist_supported': (_i, [_i, _i, _i, _i, _ip, _i, _i, _ip, _i, _i, _i]),
    'pacoh_map_persist': (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _vp, _i, _vp, _i, _i, _i, _i, _ip, _i, _i, _i, _ip, _i, _i,
                               _i, _i, _i, _d, _ip, _ip, _i, _d, _d, _vp, _vp, _vp, _i, _vp]),
    'pacoh_map_task_workspace_bytes': (_sz, [_i, _i, _i, _i, _i, _ip, _i, _i, _ip, _i, _i, _i, _i]),
    'pacoh_map_task_setup': (_i, [_vp, _i, _i, _i, _i, _i, _i, _ip, _i, _i, _i, _ip, _i, _i, _vp, _sz, _i, _vp]),
    'pacoh_svgd_task_workspace_bytes': (_sz, [_i, _i, _i, _i, _i, _i, _ip, _i, _i, _ip, _i, _i, _i, _i]),
    'pacoh_svgd_task_setup': (_i, [_i, _i, _i, _i, _i, _i, _i, _ip, _i, _i, _i, _ip, _i, _i, _vp, _sz, _i, _vp]),
    'pacoh_svgd_task_step': (_i, [_vp, _l, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _ip, _i, _i, _i, _ip, _i, _i, _vp, _vp, _vp, _i, _i, _i,
                                  _vp, _l, _vp, _d, _vp, _vp, _sz, _vp, _vp, _i, _vp, _i, _i, _vp]),
    'pacoh_map_task_step': (_i, [_vp, _l, _vp, _vp, _vp, _i, _i, _i, _i, _i, _ip, _i, _i, _i, _ip, _i, _i, _vp, _vp, _vp, _i, _i, _i,
                                 _vp, _l, _vp, _d, _vp, _vp, _sz, _vp, _i, _vp]),
    'pacoh_comm_unique_id': (_i, [_vp]),
    'pacoh_comm_init': (_i, [_vp, _i, _i, _c.POINTER(_vp)]),
    'pacoh_allreduce_sum': (_i, [_vp, _l, _i, _vp, _vp]),
    'pacoh_comm_destroy': (_i, [_vp]),
}

_lib = None
ABI_VERSION = 14              # pacoh_abi_version() of the library this table was written for


def load_library():
    """Load (once) and return the ctypes handle; raises if the HIP library has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                'libpacoh_gp.so not found at %s -- build it with `python -c "import __graft_entry__ as g; '
                'g.build()"` (needs hipcc). There is no CPU fallback for the PACOH GP path.' % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            if not hasattr(lib, name):
                raise RuntimeError('libpacoh_gp.so does not export %s (stale build?)' % name)
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        if lib.pacoh_abi_version() != ABI_VERSION:           # same symbols, other argument lists: calling on would corrupt memory
            raise RuntimeError('%s has ABI version %d, this binding expects %d (stale build?)'
                               % (LIB_PATH, lib.pacoh_abi_version(), ABI_VERSION))
        _lib = lib
    return _lib


def reload_env():
    """the library reads its PACOH_* switches once at load time (csrc/switches.h); a test or tool that changes os.environ afterwards
    calls this to have the change seen"""
    load_library().pacoh_reload_env()


def dtype_code(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.float64:
        return F64
    raise TypeError('PACOH kernels support float32/float64 tensors, got %s' % t.dtype)


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)
_raw_device = getattr(torch._C, '_cuda_getDevice', None)


def _ptr(t, like=None):
    """device pointer of a contiguous HIP tensor (None -> NULL)"""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError('PACOH kernels need tensors on a HIP device (got %s); there is no CPU path' % t.device)
    if not t.is_contiguous():
        raise RuntimeError('PACOH kernels need contiguous tensors')
    if like is not None and t.dtype != like.dtype and t.dtype not in (torch.int32,):
        raise TypeError('mixed dtypes in one PACOH call: %s vs %s' % (t.dtype, like.dtype))
    if _raw_device is not None and t.device.index != _raw_device():
        # launches go to the CURRENT device's stream: a pointer into another device's memory would fault or be read cross-device
        raise RuntimeError('tensor on %s but the current HIP device is %d: call torch.cuda.set_device first' % (t.device, _raw_device()))
    return ctypes.c_void_p(t.data_ptr())




def _stream():
    """hipStream_t of torch's current stream on the current device (follows torch.cuda.stream() / graph capture).  The raw
    accessors skip ~8 us of Python per call (a tenth of the host time of a step); the public API is the fallback"""
    if _raw_stream is not None and _raw_device is not None:
        return ctypes.c_void_p(_raw_stream(_raw_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


# Optional per-kernel timing with HIP events on the launch stream (bench.py sets PROFILE = {}):
# name -> list of (start_event, end_event) around the C-ABI call.
PROFILE = None


class _Roctx:
    """roctx ranges around every C-ABI call (step_begin / mlp_fwd / gp_lml_fwdbwd / mlp_bwd / hyper_bwd / svgd_phi ... -- the names
    of the per-kernel breakdown), so that `rocprofv3 --marker-trace --kernel-trace -- python3 bench.py` attributes the kernels of
    a step to its phases.  No-ops unless PACOH_ROCTX=1; ranges are host-side, so the attributed pass is the eager one
    (PACOH_NO_GRAPH=1 or bench.py's per-kernel pass) -- a graph replay is one host call.  The profiler's own marker library is
    preferred (rocprofv3 intercepts it), the classic libroctx64 is the fallback."""
    push = pop = None

    @classmethod
    def setup(cls):
        if os.environ.get('PACOH_ROCTX', '0') != '1':
            return
        for name in ('librocprofiler-sdk-roctx.so', 'librocprofiler-sdk-roctx.so.1', 'libroctx64.so', 'libroctx64.so.4'):
            try:
                lib = ctypes.CDLL(name)
                push, pop = lib.roctxRangePushA, lib.roctxRangePop
            except (OSError, AttributeError):
                continue
            push.argtypes, push.restype, pop.argtypes, pop.restype = [ctypes.c_char_p], ctypes.c_int, [], ctypes.c_int
            cls.push, cls.pop = push, pop
            return
        raise RuntimeError('PACOH_ROCTX=1 but neither librocprofiler-sdk-roctx nor libroctx64 could be loaded')


_Roctx.setup()


class _Timed:
    def __init__(self, name):
        self.name = name

    def __enter__(self):
        if _Roctx.push is not None:
            _Roctx.push(('pacoh:' + self.name).encode())
        if PROFILE is not None:
            self.start = torch.cuda.Event(enable_timing=True)
            self.end = torch.cuda.Event(enable_timing=True)
            self.start.record()
        return self

    def __exit__(self, *exc):
        if PROFILE is not None:
            self.end.record()
            PROFILE.setdefault(self.name, []).append((self.start, self.end))
        if _Roctx.pop is not None:
            _Roctx.pop()
        return False


class _Range:
    def __init__(self, name=None):
        self.name = name

    def __enter__(self):
        if self.name is not None:
            _Roctx.push(self.name)
        return self

    def __exit__(self, *exc):
        if self.name is not None:
            _Roctx.pop()
        return False


_NULL_CTX = _Range()


def roctx_range(name):
    """context manager: a named roctx range around a phase of a step (no-op unless PACOH_ROCTX=1)"""
    return _Range(('pacoh:step:' + name).encode()) if _Roctx.push is not None else _NULL_CTX


def profile_summary():
    """{name: (launches, total_ms)} -- call after torch.cuda.synchronize()"""
    out = {}
    for name, evs in (PROFILE or {}).items():
        out[name] = (len(evs), sum(s.elapsed_time(e) for s, e in evs))
    return out


def _check(rc, what):
    if rc != 0:
        raise RuntimeError('%s failed: %s' % (what, ERRORS.get(rc, rc)))


def _hidden_arr(hidden):
    arr = (ctypes.c_int32 * max(1, len(hidden)))(*hidden)
    return arr


# ------------------------------------------------------------------------------------------------
# thin typed wrappers (shapes are validated here, the C side validates limits)
# ------------------------------------------------------------------------------------------------

def gram_rbf_ard(z1, z1_div, z2, z2_div, lengthscale, outputscale, noise, add_noise_diag, B, P, kernel=KERNEL_RBF):
    lib = load_library()
    n, f = z1.shape[-2], z1.shape[-1]
    m = z2.shape[-2]
    K = torch.empty(B, n, m, dtype=z1.dtype, device=z1.device)
    with _Timed('gram_rbf_ard'):
        _check(lib.pacoh_gram_rbf_ard(_ptr(z1), z1_div, _ptr(z2, z1), z2_div, _ptr(lengthscale, z1), _ptr(outputscale, z1),
                                      _ptr(noise, z1), int(bool(add_noise_diag)), _ptr(K), B, P, n, m, _kf(f, kernel), dtype_code(z1),
                                      _stream()), 'pacoh_gram_rbf_ard')
    return K


FORCE_DENSE = False          # tests: route every LML / predictive call through the large-n (HBM-resident) path


def gp_small_max_n(dtype, want_grad):
    return load_library().pacoh_gp_small_max_n(F32 if dtype == torch.float32 else F64, int(want_grad))


def gp_lml_fwd(z, z_div, mean, mean_mode, y, y_div, lengthscale, outputscale, noise, B, P, n_valid=None,
               want_alpha=False, want_L=False, kernel=KERNEL_RBF):
    lib = load_library()
    n, f = z.shape[-2], z.shape[-1]
    dev, dt = z.device, z.dtype
    lml = torch.empty(B, dtype=dt, device=dev)
    alpha = torch.empty(B, n, dtype=dt, device=dev) if want_alpha else None
    L = torch.empty(B, n, n, dtype=dt, device=dev) if want_L else None
    info = torch.empty(B, dtype=torch.int32, device=dev)
    if FORCE_DENSE or n > gp_small_max_n(dt, False):
        if want_alpha or want_L:
            raise RuntimeError('alpha / L outputs are only available on the small-n path (n <= %d)' % gp_small_max_n(dt, False))
        lml = _gp_lml_dense(z, z_div, mean, mean_mode, y, y_div, lengthscale, outputscale, noise, n_valid, None, B, P, info, False,
                            kernel=kernel)[0]
        return lml, None, None, info
    with _Timed('gp_lml_fwd'):
        _check(lib.pacoh_gp_lml_fwd(_ptr(z), z_div, _ptr(mean, z), mean_mode, _ptr(y, z), y_div, _ptr(lengthscale, z),
                                    _ptr(outputscale, z), _ptr(noise, z), _ptr(n_valid), _ptr(lml), _ptr(alpha), _ptr(L),
                                    _ptr(info), B, P, n, _kf(f, kernel), dtype_code(z), _stream()), 'pacoh_gp_lml_fwd')
    return lml, alpha, L, info


_DENSE_WS = {}
DENSE_WS_BYTES = 8 << 30          # scratch budget of the large-n path; larger meta-batches are processed in slabs of whole tasks


def _workspace(key, nbytes, device):
    """grow-only scratch buffer per (purpose, device): the dense path needs O(B n^2) bytes"""
    ws = _DENSE_WS.get((key, device))
    if ws is None or ws.numel() < nbytes:
        ws = _DENSE_WS[(key, device)] = torch.empty(max(1, nbytes), dtype=torch.uint8, device=device)
    return ws


def _gp_lml_dense(z, z_div, mean, mean_mode, y, y_div, lengthscale, outputscale, noise, n_valid, g_lml, B, P, info,
                  want_grad, want_dz=True, kernel=KERNEL_RBF):
    """large-n path (matrices materialised in HBM): -> (lml, d_z, d_mean, d_ls, d_os, d_noise)"""
    lib = load_library()
    n, f = z.shape[-2], z.shape[-1]
    dev, dt = z.device, z.dtype
    code = dtype_code(z)
    lml = torch.empty(B, dtype=dt, device=dev)
    d_z = d_mean = d_ls = d_os = d_noise = None
    if want_grad:
        d_z = torch.empty(B, n, f, dtype=dt, device=dev) if want_dz else None
        if mean_mode == MEAN_VECTOR:
            d_mean = torch.empty(B, n, dtype=dt, device=dev)
        elif mean_mode == MEAN_CONST:
            d_mean = torch.empty(B, dtype=dt, device=dev)
        d_ls = torch.empty(B, f, dtype=dt, device=dev)
        d_os = torch.empty(B, dtype=dt, device=dev) if outputscale is not None else None
        d_noise = torch.empty(B, dtype=dt, device=dev)
    # The scratch is O(B n^2).  The entry point itself runs the batch in slabs of whole tasks when it is handed less than the whole
    # batch needs, and pads context sizes with misaligned rows onto the left-looking kernels (round 6: both lived here before): the
    # binding only bounds the buffer -- DENSE_WS_BYTES, but never less than one task's share
    need = lib.pacoh_gp_lml_dense_workspace_bytes(B, n, _kf(f, kernel), code, int(want_grad))
    if need > DENSE_WS_BYTES and B % P == 0 and B > P:
        one = max(1, lib.pacoh_gp_lml_dense_workspace_bytes(P, n, _kf(f, kernel), code, int(want_grad)))
        need = min(need, max(one, DENSE_WS_BYTES // one * one))
    ws = _workspace('lml', need, dev)
    with _Timed('gp_lml_dense'):
        _check(lib.pacoh_gp_lml_dense(_ptr(z), z_div, _ptr(mean, z), mean_mode, _ptr(y, z), y_div, _ptr(lengthscale, z),
                                      _ptr(outputscale, z), _ptr(noise, z), _ptr(n_valid), _ptr(g_lml, z), _ptr(lml), _ptr(d_z),
                                      _ptr(d_mean), _ptr(d_ls), _ptr(d_os), _ptr(d_noise), _ptr(info), _ptr(ws), need,
                                      B, P, n, _kf(f, kernel), code, _stream()), 'pacoh_gp_lml_dense')
    return lml, d_z, d_mean, d_ls, d_os, d_noise


def gp_lml_fwdbwd(z, z_div, mean, mean_mode, y, y_div, lengthscale, outputscale, noise, B, P, n_valid=None,
                  g_lml=None, want_dz=True, kernel=KERNEL_RBF):
    lib = load_library()
    n, f = z.shape[-2], z.shape[-1]
    dev, dt = z.device, z.dtype
    lml = torch.empty(B, dtype=dt, device=dev)
    d_z = torch.empty(B, n, f, dtype=dt, device=dev) if want_dz else None
    if mean_mode == MEAN_VECTOR:
        d_mean = torch.empty(B, n, dtype=dt, device=dev)
    elif mean_mode == MEAN_CONST:
        d_mean = torch.empty(B, dtype=dt, device=dev)
    else:
        d_mean = None
    d_ls = torch.empty(B, f, dtype=dt, device=dev)
    d_os = torch.empty(B, dtype=dt, device=dev) if outputscale is not None else None
    d_noise = torch.empty(B, dtype=dt, device=dev)
    info = torch.empty(B, dtype=torch.int32, device=dev)
    if FORCE_DENSE or n > gp_small_max_n(dt, True):
        return _gp_lml_dense(z, z_div, mean, mean_mode, y, y_div, lengthscale, outputscale, noise, n_valid, g_lml, B, P, info,
                             True, want_dz, kernel=kernel) + (info,)
    with _Timed('gp_lml_fwdbwd'):
        _check(lib.pacoh_gp_lml_fwdbwd(_ptr(z), z_div, _ptr(mean, z), mean_mode, _ptr(y, z), y_div, _ptr(lengthscale, z),
                                       _ptr(outputscale, z), _ptr(noise, z), _ptr(n_valid), _ptr(g_lml, z), _ptr(lml),
                                       _ptr(d_z), _ptr(d_mean), _ptr(d_ls), _ptr(d_os), _ptr(d_noise), _ptr(info),
                                       B, P, n, _kf(f, kernel), dtype_code(z), _stream()), 'pacoh_gp_lml_fwdbwd')
    return lml, d_z, d_mean, d_ls, d_os, d_noise, info


def gp_predict(z_ctx, z_div, mean_ctx, mean_mode, y, y_div, z_tst, zt_div, mean_tst, lengthscale, outputscale, noise,
               B, P, n_valid=None, want_cov=False, kernel=KERNEL_RBF):
    lib = load_library()
    n, f = z_ctx.shape[-2], z_ctx.shape[-1]
    m = z_tst.shape[-2]
    dev, dt = z_ctx.device, z_ctx.dtype
    code = dtype_code(z_ctx)
    mu = torch.empty(B, m, dtype=dt, device=dev)
    var = torch.empty(B, m, dtype=dt, device=dev)
    cov = torch.empty(B, m, m, dtype=dt, device=dev) if want_cov else None
    info = torch.empty(B, dtype=torch.int32, device=dev)
    if FORCE_DENSE or n > gp_small_max_n(dt, False):
        ws = _workspace('predict', lib.pacoh_gp_predict_dense_workspace_bytes(B, n, m, code), dev)
        with _Timed('gp_predict_dense'):
            _check(lib.pacoh_gp_predict_dense(_ptr(z_ctx), z_div, _ptr(mean_ctx, z_ctx), mean_mode, _ptr(y, z_ctx), y_div,
                                              _ptr(z_tst, z_ctx), zt_div, _ptr(mean_tst, z_ctx), _ptr(lengthscale, z_ctx),
                                              _ptr(outputscale, z_ctx), _ptr(noise, z_ctx), _ptr(n_valid), _ptr(mu), _ptr(var),
                                              _ptr(cov), _ptr(info), _ptr(ws), B, P, n, m, _kf(f, kernel), code, _stream()),
                   'pacoh_gp_predict_dense')
        return mu, var, cov, info
    ws_bytes = lib.pacoh_gp_predict_workspace_bytes(B, n, m, code, int(want_cov))
    ws = torch.empty(max(1, ws_bytes), dtype=torch.uint8, device=dev) if want_cov else None
    with _Timed('gp_predict'):
        _check(lib.pacoh_gp_predict(_ptr(z_ctx), z_div, _ptr(mean_ctx, z_ctx), mean_mode, _ptr(y, z_ctx), y_div,
                                    _ptr(z_tst, z_ctx), zt_div, _ptr(mean_tst, z_ctx), _ptr(lengthscale, z_ctx),
                                    _ptr(outputscale, z_ctx), _ptr(noise, z_ctx), _ptr(n_valid), _ptr(mu), _ptr(var),
                                    _ptr(cov), _ptr(info), _ptr(ws), B, P, n, m, _kf(f, kernel), code, _stream()), 'pacoh_gp_predict')
    return mu, var, cov, info


def mvn_logprob_dense(A, resid, scale=1.0, want_alpha=False):
    """A [B,n,n] is DESTROYED (overwritten with its Cholesky factor)."""
    lib = load_library()
    B, n = A.shape[0], A.shape[-1]
    logp = torch.empty(B, dtype=A.dtype, device=A.device)
    alpha = torch.empty(B, n, dtype=A.dtype, device=A.device) if want_alpha else None
    info = torch.empty(B, dtype=torch.int32, device=A.device)
    with _Timed('mvn_logprob_dense'):
        _check(lib.pacoh_mvn_logprob_dense(_ptr(A), _ptr(resid, A), _ptr(logp), _ptr(alpha), _ptr(info), float(scale),
                                           B, n, dtype_code(A), _stream()), 'pacoh_mvn_logprob_dense')
    return logp, alpha, info


def _mlp_fwd_ws(need, device, holder):
    """scratch of the layer-wise MLP forward (only shapes outside the register-resident kernels need any).  It lives in the
    CALLER's `holder` dict (an engine's workspace table), keyed by size class, and a buffer is never dropped once handed out: a
    captured step graph bakes the raw pointer in, so a later, larger eager call (predict with a big test set) must not free what
    the graph still writes to.  holder=None: a fresh buffer per call."""
    if need == 0:
        return None
    if holder is None:
        return torch.empty(need, dtype=torch.uint8, device=device)
    pool = holder.setdefault(('mlp_fwd_ws', device), [])
    for ws in pool:
        if ws.numel() >= need:
            return ws
    ws = torch.empty(need, dtype=torch.uint8, device=device)
    pool.append(ws)
    return ws


def mlp_fwd(x, x_div, theta_block, theta_stride, P, d_in, hidden, d_out, B, n, ws_holder=None, stash=None):
    """stash (mlp_stash()): receives the activations the matching mlp_bwd_hyper(stash=...) would otherwise recompute"""
    lib = load_library()
    out = torch.empty(B, n, d_out, dtype=x.dtype, device=x.device)
    harr, code = _hidden_arr(hidden), dtype_code(x)
    ws = _mlp_fwd_ws(lib.pacoh_mlp_fwd_workspace_bytes(B, P, n, d_in, harr, len(hidden), d_out, code), x.device, ws_holder)
    with _Timed('mlp_fwd'):
        if stash is not None:
            _check(lib.pacoh_mlp_fwd_stash(_ptr(x), x_div, ctypes.c_void_p(theta_block.data_ptr()), theta_stride, P, d_in,
                                           harr, len(hidden), d_out, _ptr(out), _ptr(ws), _ptr(stash), B, n, code, _stream()),
                   'pacoh_mlp_fwd_stash')
        else:
            _check(lib.pacoh_mlp_fwd(_ptr(x), x_div, ctypes.c_void_p(theta_block.data_ptr()), theta_stride, P, d_in,
                                     harr, len(hidden), d_out, _ptr(out), _ptr(ws), B, n, code, _stream()),
                   'pacoh_mlp_fwd')
    return out


def mlp_stash(x, P, d_in, hidden, d_out, B, n, stash=None):
    """activation stash for a mlp_fwd / mlp_bwd_hyper pair of ONE network of this shape (reused if large enough) | None"""
    lib = load_library()
    need = lib.pacoh_mlp_stash_bytes(B, P, n, d_in, _hidden_arr(hidden), len(hidden), d_out, dtype_code(x))
    if need == 0 or need > MLP_STASH_MAX_BYTES:
        return None
    if stash is None or stash.numel() < need:
        stash = torch.empty(need, dtype=torch.uint8, device=x.device)
    return stash


def mlp_bwd(x, x_div, theta_block, theta_stride, P, d_in, hidden, d_out, g_out, d_theta_block, d_theta_stride,
            accumulate, B, n, workspace=None):
    lib = load_library()
    code = dtype_code(x)
    harr = _hidden_arr(hidden)
    need = lib.pacoh_mlp_bwd_workspace_bytes(B, P, n, d_in, harr, len(hidden), d_out, code)
    if workspace is None or workspace.numel() * workspace.element_size() < need:
        workspace = torch.empty(max(1, need), dtype=torch.uint8, device=x.device)
    with _Timed('mlp_bwd'):
        _check(lib.pacoh_mlp_bwd(_ptr(x), x_div, ctypes.c_void_p(theta_block.data_ptr()), theta_stride, P, d_in, harr,
                                 len(hidden), d_out, _ptr(g_out, x), ctypes.c_void_p(d_theta_block.data_ptr()),
                                 d_theta_stride, int(bool(accumulate)), _ptr(workspace), B, n, code, _stream()),
               'pacoh_mlp_bwd')
    return workspace


def mlp_bwd_hyper(x, x_div, theta, lo, P, d_in, hidden, d_out, g_out, grad, B, n, T, off_ls, f, off_os, off_noise, off_const,
                  d_ls, d_os, d_noise, d_const, lml=None, lik=None, lik_scale=1.0, info=None, fail_flag=None, kernel=KERNEL_RBF,
                  workspace=None, svgd_bw=None, opt=None, stash=None):
    """mlp_bwd of the ONE network whose block starts at column lo of theta[P, D] (gradient into grad[:, lo:]) + hyper_bwd on the
    rows, in one C-ABI call (one launch less on the fused path); returns the workspace for reuse"""
    lib = load_library()
    code, harr = dtype_code(x), _hidden_arr(hidden)
    D = theta.shape[1]
    es = theta.element_size()
    need = lib.pacoh_mlp_bwd_workspace_bytes(B, P, n, d_in, harr, len(hidden), d_out, code)
    if workspace is None or workspace.numel() * workspace.element_size() < need:
        workspace = torch.empty(max(1, need), dtype=torch.uint8, device=x.device)
    with _Timed('mlp_bwd'):
        _check(lib.pacoh_mlp_bwd_hyper(_ptr(x), x_div, ctypes.c_void_p(theta.data_ptr() + lo * es), D, P, d_in, harr, len(hidden), d_out,
                                       _ptr(g_out, x), ctypes.c_void_p(grad.data_ptr() + lo * es), grad.shape[1], _ptr(workspace),
                                       _ptr(stash), B, n, _ptr(theta, x), _ptr(grad, x), T, off_ls, _kf(f, kernel), off_os, off_noise, off_const,
                                       _ptr(d_ls, x), _ptr(d_os, x), _ptr(d_noise, x), _ptr(d_const, x), _ptr(lml, x), _ptr(lik, x),
                                       float(lik_scale), _ptr(info if fail_flag is not None else None),
                                       _ptr(fail_flag if info is not None else None), *_svgd_bw(svgd_bw), _opt_ptr(opt), code, _stream()),
               'pacoh_mlp_bwd_hyper')
    return workspace


MLP_STASH_MAX_BYTES = 8 << 30     # no activation stash beyond this (the backward then recomputes, as in rounds 1-2)


def mlp2_stash(x, P, d_in, hidden, d_out_a, d_out_b, B, n, stash=None):
    """activation stash for a mlp2_fwd / mlp2_bwd pair of these shapes (reused if large enough) or None if this shape keeps none"""
    lib = load_library()
    need = lib.pacoh_mlp2_stash_bytes(B, P, n, d_in, _hidden_arr(hidden), len(hidden), d_out_a, d_out_b, dtype_code(x))
    if need == 0 or need > MLP_STASH_MAX_BYTES:
        return None
    if stash is None or stash.numel() < need:
        stash = torch.empty(need, dtype=torch.uint8, device=x.device)
    return stash


def mlp2_fwd(x, x_div, theta, P, d_in, hidden, off_a, d_out_a, off_b, d_out_b, B, n, ws_holder=None, stash=None, svgd_tail=None):
    """two networks of the same hidden shape (blocks at element offsets off_a / off_b of the rows of theta[P, D]) on the
    same inputs: -> (out_a[B,n,d_out_a], out_b[B,n,d_out_b]); one launch on the fused fp32 path.  stash (mlp2_stash()) receives
    the activations the matching mlp2_bwd(stash=...) would otherwise recompute.  svgd_tail = (particles, workspace, counter): the
    pipelined SVGD step's distance matrix + counter increment ride in the same launch (svgd_dist_advance)"""
    lib = load_library()
    out_a = torch.empty(B, n, d_out_a, dtype=x.dtype, device=x.device)
    out_b = torch.empty(B, n, d_out_b, dtype=x.dtype, device=x.device)
    harr, code = _hidden_arr(hidden), dtype_code(x)
    ws = _mlp_fwd_ws(lib.pacoh_mlp2_fwd_workspace_bytes(B, P, n, d_in, harr, len(hidden), d_out_a, d_out_b, code), x.device, ws_holder)
    with _Timed('mlp_fwd'):
        if svgd_tail is not None:
            sv_X, sv_ws, ctr = svgd_tail
            _check(lib.pacoh_mlp2_fwd_svgd(_ptr(x), x_div, _ptr(theta, x), theta.shape[1], P, d_in, harr, len(hidden), off_a, d_out_a,
                                           _ptr(out_a), off_b, d_out_b, _ptr(out_b), _ptr(ws), _ptr(stash), B, n, _ptr(sv_X, x),
                                           _ptr(sv_ws), sv_X.shape[0], sv_X.shape[1], _ptr(ctr), code, _stream()), 'pacoh_mlp2_fwd_svgd')
        else:
            _check(lib.pacoh_mlp2_fwd(_ptr(x), x_div, _ptr(theta, x), theta.shape[1], P, d_in, harr, len(hidden), off_a, d_out_a,
                                      _ptr(out_a), off_b, d_out_b, _ptr(out_b), _ptr(ws), _ptr(stash), B, n, code, _stream()), 'pacoh_mlp2_fwd')
    return out_a, out_b


def mlp2_bwd(x, x_div, theta, P, d_in, hidden, off_a, d_out_a, g_a, off_b, d_out_b, g_b, d_theta, accumulate, B, n,
             workspace=None, stash=None):
    """backward of mlp2_fwd into the rows of d_theta[P, D] (blocks at off_a / off_b); returns the workspace for reuse"""
    lib = load_library()
    harr, code = _hidden_arr(hidden), dtype_code(x)
    need = lib.pacoh_mlp2_bwd_workspace_bytes(B, P, n, d_in, harr, len(hidden), d_out_a, d_out_b, code)
    if workspace is None or workspace.numel() * workspace.element_size() < need:
        workspace = torch.empty(max(1, need), dtype=torch.uint8, device=x.device)
    with _Timed('mlp_bwd'):
        _check(lib.pacoh_mlp2_bwd(_ptr(x), x_div, _ptr(theta, x), theta.shape[1], P, d_in, harr, len(hidden), off_a, d_out_a,
                                  _ptr(g_a, x), off_b, d_out_b, _ptr(g_b, x), _ptr(d_theta, x), d_theta.shape[1],
                                  int(bool(accumulate)), _ptr(workspace), _ptr(stash), B, n, code, _stream()), 'pacoh_mlp2_bwd')
    return workspace


def mlp2_bwd_hyper(x, x_div, theta, P, d_in, hidden, off_a, d_out_a, g_a, off_b, d_out_b, g_b, d_theta, B, n, T, off_ls, f, off_os,
                   off_noise, off_const, d_ls, d_os, d_noise, d_const, lml=None, lik=None, lik_scale=1.0, info=None, fail_flag=None,
                   workspace=None, stash=None, svgd_bw=None, opt=None):
    """mlp2_bwd + hyper_bwd (grad = d_theta) in one C-ABI call: the gradient epilogue of a step; returns the workspace for reuse.
    svgd_bw = (SVGD workspace, P, D): the step's median bandwidth is computed by one more workgroup (hyper_bwd)"""
    lib = load_library()
    harr, code = _hidden_arr(hidden), dtype_code(x)
    need = lib.pacoh_mlp2_bwd_workspace_bytes(B, P, n, d_in, harr, len(hidden), d_out_a, d_out_b, code)
    if workspace is None or workspace.numel() * workspace.element_size() < need:
        workspace = torch.empty(max(1, need), dtype=torch.uint8, device=x.device)
    with _Timed('mlp_bwd'):
        _check(lib.pacoh_mlp2_bwd_hyper(_ptr(x), x_div, _ptr(theta, x), theta.shape[1], P, d_in, harr, len(hidden), off_a, d_out_a,
                                        _ptr(g_a, x), off_b, d_out_b, _ptr(g_b, x), _ptr(d_theta, x), d_theta.shape[1], 0,
                                        _ptr(workspace), _ptr(stash), B, n, T, off_ls, f, off_os, off_noise, off_const,
                                        _ptr(d_ls, x), _ptr(d_os, x), _ptr(d_noise, x), _ptr(d_const, x), _ptr(lml, x), _ptr(lik, x),
                                        float(lik_scale), _ptr(info if fail_flag is not None else None),
                                        _ptr(fail_flag if info is not None else None), *_svgd_bw(svgd_bw), _opt_ptr(opt), code, _stream()),
               'pacoh_mlp2_bwd_hyper')
    return workspace


class StepNext(ctypes.Structure):
    """pacoh_step_next (include/pacoh_gp.h): the pipelined feed carried by the gradient epilogue of a PACOH-MAP iteration"""
    _fields_ = [('counter', _vp), ('sc2', _vp), ('n_sc', _i), ('idx_all', _vp), ('tb', _i), ('sc_all', _vp), ('x', _vp), ('y', _vp),
                ('n_valid', _vp), ('out_x', _vp), ('out_y', _vp), ('out_n_valid', _vp), ('n', _i), ('d', _i), ('noise_floor', _d),
                ('ls', _vp), ('os', _vp), ('noise', _vp)]


class AdamInline(ctypes.Structure):
    """pacoh_adam_inline (include/pacoh_gp.h): the AdamW step of a PACOH-MAP iteration folded into the gradient epilogue"""
    _fields_ = [('param', _vp), ('exp_avg', _vp), ('exp_avg_sq', _vp), ('scalars', _vp), ('beta1', _d), ('beta2', _d),
                ('n_seg', _i), ('seg_lo', _i * 4), ('seg_hi', _i * 4), ('step_counter', _vp), ('loss_cum', _vp),
                ('next', ctypes.POINTER(StepNext))]


class MapPersistPlan:
    """argument block of pacoh_map_persist for one learner (include/pacoh_gp.h, "K whole PACOH-MAP iterations per launch"): the
    parameter layout of the learner's engine.ParamLayout in the entry point's terms.  supported(): does the persistent kernel
    take this shape?"""

    def __init__(self, layout, tasks, tb, noise_floor, segments, dtype):
        lay = layout
        self.n, self.d, self.tb, self.dtype = int(tasks.n), int(tasks.x.shape[2]), int(tb), dtype
        self.mean_mode = {'NN': MEAN_VECTOR, 'constant': MEAN_CONST, 'zero': MEAN_ZERO}[lay.mean_module]
        self.off_mean = (lay.block_range('mean_nn.')[0] if lay.mean_module == 'NN'
                         else (lay.slices['constant_mean'][0] if lay.mean_module == 'constant' else -1))
        self.mean_hidden = list(lay.mean_nn_layers) if lay.mean_module == 'NN' else []
        self.kernel_nn = int(lay.covar_module == 'NN')
        self.off_kernel = lay.block_range('kernel_nn.')[0] if self.kernel_nn else -1
        self.kernel_hidden = list(lay.kernel_nn_layers) if self.kernel_nn else []
        self.f = int(lay.feature_dim)
        self.off_ls = lay.slices['lengthscale_raw'][0]
        self.off_os = lay.slices['outputscale_raw'][0] if lay.with_outputscale else -1
        self.off_noise = lay.slices['noise_raw'][0]
        self.noise_floor = float(noise_floor)
        self.rbf = lay.kernel_code == KERNEL_RBF
        self.D = lay.D
        k = len(segments)
        self.n_seg = k
        self.seg_lo = (ctypes.c_int32 * 4)(*([int(lo) for lo, _ in segments] + [0] * (4 - k))) if k <= 4 else None
        self.seg_hi = (ctypes.c_int32 * 4)(*([int(hi) for _, hi in segments] + [0] * (4 - k))) if k <= 4 else None
        self._mh, self._kh = _hidden_arr(self.mean_hidden), _hidden_arr(self.kernel_hidden)

    def supported(self):
        if not self.rbf or self.n_seg > 4 or self.n_seg < 1:
            return False
        code = F32 if self.dtype == torch.float32 else F64
        return bool(load_library().pacoh_map_persist_supported(self.n, self.d, self.tb, self.mean_mode, self._mh, len(self.mean_hidden),
                                                               self.kernel_nn, self._kh, len(self.kernel_hidden), self.f, code))


def map_persist(plan, theta, exp_avg, exp_avg_sq, tasks, idx_rows, sc_rows, K, loss_last, loss_cum, fail_flag, beta1=0.9, beta2=0.999):
    """K PACOH-MAP iterations in one launch (pacoh_map_persist): rows 0..K-1 of idx_rows [>= K, tb] (int64, device) / sc_rows [>= K, SC_COUNT]"""
    lib = load_library()
    assert theta.shape[0] == 1 and idx_rows.dtype == torch.int64 and idx_rows.shape[1] == plan.tb
    with _Timed('map_persist'):
        _check(lib.pacoh_map_persist(_ptr(theta), _ptr(exp_avg, theta), _ptr(exp_avg_sq, theta), plan.D, _ptr(tasks.x, theta), _ptr(tasks.y, theta),
                                     _ptr(tasks.n_valid) if tasks.ragged else None, plan.n, plan.d, _ptr(idx_rows), plan.tb,
                                     _ptr(sc_rows, theta), sc_rows.shape[1], int(K),
                                     plan.mean_mode, plan.off_mean, plan._mh, len(plan.mean_hidden),
                                     plan.kernel_nn, plan.off_kernel, plan._kh, len(plan.kernel_hidden), plan.f,
                                     plan.off_ls, plan.off_os, plan.off_noise, plan.noise_floor,
                                     plan.seg_lo, plan.seg_hi, plan.n_seg, float(beta1), float(beta2),
                                     _ptr(loss_last, theta), _ptr(loss_cum, theta), _ptr(fail_flag), dtype_code(theta), _stream()),
               'pacoh_map_persist')


def map_task_workspace(plan, tb, device, workspace=None, any_size=False):
    """workspace of map_task_step for a batch of tb tasks (None: the task-fused kernel does not take this shape, or -- unless any_size --
    its workgroups would not all be resident at once, where the four-launch iteration is faster)"""
    need = load_library().pacoh_map_task_workspace_bytes(plan.D, plan.n, plan.d, int(tb), plan.mean_mode, plan._mh, len(plan.mean_hidden), plan.kernel_nn,
                                                        plan._kh, len(plan.kernel_hidden), plan.f, int(bool(any_size)),
                                                        F32 if plan.dtype == torch.float32 else F64)
    if need == 0 or not plan.rbf:
        return None
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(need, dtype=torch.uint8, device=device)
    return workspace


def map_task_setup(plan, theta, tb, workspace):
    """(re)build the parameter image of map_task_step's workspace from theta: before the first step on it and after any change of theta
    that did not come from map_task_step itself (pacoh_map_task_setup)"""
    lib = load_library()
    with _Timed('map_task_setup'):
        _check(lib.pacoh_map_task_setup(_ptr(theta), plan.D, plan.n, plan.d, int(tb), plan.mean_mode, plan.off_mean, plan._mh, len(plan.mean_hidden),
                                        plan.kernel_nn, plan.off_kernel, plan._kh, len(plan.kernel_hidden), plan.f, _ptr(workspace),
                                        workspace.numel(), dtype_code(theta), _stream()), 'pacoh_map_task_setup')


def map_task_step(plan, theta, batch, hypers, grad, lik, lik_scale, fail_flag, workspace, opt):
    """one PACOH-MAP iteration's likelihood, gradient and update as two launches (pacoh_map_task_step): batch = the gathered TaskBatch,
    hypers = (ls [1, f], os [1] | None, noise [1]) transformed, grad [1, D], lik [1]; opt = adam_inline(...) with its step_next"""
    lib = load_library()
    ls, os_, noise = hypers
    with _Timed('map_task_step'):
        _check(lib.pacoh_map_task_step(_ptr(theta), theta.shape[1], _ptr(batch.x, theta), _ptr(batch.y, theta),
                                       _ptr(batch.n_valid) if batch.n_valid is not None else None, plan.n, plan.d, int(batch.T),
                                       plan.mean_mode, plan.off_mean, plan._mh, len(plan.mean_hidden),
                                       plan.kernel_nn, plan.off_kernel, plan._kh, len(plan.kernel_hidden), plan.f,
                                       _ptr(ls, theta), _ptr(os_, theta), _ptr(noise, theta), plan.off_ls, plan.off_os, plan.off_noise,
                                       _ptr(grad, theta), grad.shape[1], _ptr(lik, theta), float(lik_scale), _ptr(fail_flag),
                                       _ptr(workspace), workspace.numel(), _opt_ptr(opt), dtype_code(theta), _stream()),
               'pacoh_map_task_step')


def svgd_task_workspace(plan, P, tb, device, workspace=None, any_size=False):
    """workspace of svgd_task_step for P parameter rows and a batch of tb tasks, its gather map written (None: the caller runs the
    general launch sequence -- the shape is outside the task-fused kernel's plan or, unless any_size, its workgroups would not all be
    resident at once, beyond which the throughput kernels win)"""
    lib = load_library()
    code = F32 if plan.dtype == torch.float32 else F64
    need = lib.pacoh_svgd_task_workspace_bytes(plan.D, int(P), plan.n, plan.d, int(tb), plan.mean_mode, plan._mh, len(plan.mean_hidden),
                                               plan.kernel_nn, plan._kh, len(plan.kernel_hidden), plan.f, int(bool(any_size)), code)
    if need == 0 or not plan.rbf:
        return None
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(need, dtype=torch.uint8, device=device)
    _check(lib.pacoh_svgd_task_setup(plan.D, int(P), plan.n, plan.d, int(tb), plan.mean_mode, plan.off_mean, plan._mh, len(plan.mean_hidden),
                                     plan.kernel_nn, plan.off_kernel, plan._kh, len(plan.kernel_hidden), plan.f, _ptr(workspace),
                                     workspace.numel(), code, _stream()), 'pacoh_svgd_task_setup')
    return workspace


def svgd_task_step(plan, theta, batch, hypers, grad, lik, lik_scale, fail_flag, workspace, svgd=None):
    """the likelihood half of a PACOH-SVGD / PACOH-VI step as two launches (pacoh_svgd_task_step): theta [P, D] particles / posterior
    samples, batch = the gathered TaskBatch, hypers = (ls [P, f], os [P] | None, noise [P]) transformed, grad [P, D], lik [P];
    svgd = (particles, svgd workspace, step counter, want_bandwidth): the SVGD step's distance matrix in extra workgroups"""
    lib = load_library()
    ls, os_, noise = hypers
    P, D = theta.shape
    sx = sws = ctr = None
    want_bw = 0
    if svgd is not None:
        sx, sws, ctr, want_bw = _ptr(svgd[0], theta), _ptr(svgd[1]), _ptr(svgd[2]), int(bool(svgd[3]))
    with _Timed('svgd_task_step'):
        _check(lib.pacoh_svgd_task_step(_ptr(theta), theta.stride(0), P, _ptr(batch.x, theta), _ptr(batch.y, theta),
                                        _ptr(batch.n_valid) if (batch.n_valid is not None and batch.ragged) else None, plan.n, plan.d, int(batch.T),
                                        plan.mean_mode, plan.off_mean, plan._mh, len(plan.mean_hidden),
                                        plan.kernel_nn, plan.off_kernel, plan._kh, len(plan.kernel_hidden), plan.f,
                                        _ptr(ls, theta), _ptr(os_, theta), _ptr(noise, theta), plan.off_ls, plan.off_os, plan.off_noise,
                                        _ptr(grad, theta), grad.stride(0), _ptr(lik, theta), float(lik_scale), _ptr(fail_flag),
                                        _ptr(workspace), workspace.numel(), sx, sws, D, ctr, want_bw, dtype_code(theta), _stream()),
               'pacoh_svgd_task_step')


def mlp_fused_path(B, P, n, d_in, hidden, d_out, dtype):
    """do networks of this shape run on the fused fp32 kernels? (host-side query)"""
    code = F32 if dtype == torch.float32 else F64
    return bool(load_library().pacoh_mlp_fused_path(B, P, n, d_in, _hidden_arr(hidden), len(hidden), d_out, code))


def step_next(feed, tasks, noise_floor):
    """pacoh_step_next of a pipelined engine.StepFeed (feed.pipeline() has run): the next step's operands are fetched, and the
    transformed hyper-parameters published into feed.hyp, by the gradient epilogue that carries this block"""
    o = StepNext()
    ls, os_, noise = feed.hyp
    nv, onv = (tasks.n_valid, feed.batch.n_valid) if tasks.ragged else (None, None)
    keep = (feed.ctr, feed.sc2, feed.idx_all, feed.sc_all, tasks.x, tasks.y, nv, feed.batch.x, feed.batch.y, onv, ls, os_, noise)
    ptr = lambda t: t.data_ptr() if t is not None else None
    o.counter, o.sc2, o.n_sc = ptr(feed.ctr), ptr(feed.sc2), feed.sc2.shape[1]
    o.idx_all, o.tb, o.sc_all = ptr(feed.idx_all), feed.tb, ptr(feed.sc_all)
    o.x, o.y, o.n_valid, o.out_x, o.out_y, o.out_n_valid = ptr(tasks.x), ptr(tasks.y), ptr(nv), ptr(feed.batch.x), ptr(feed.batch.y), ptr(onv)
    o.n, o.d, o.noise_floor = tasks.x.shape[1], tasks.x.shape[2], float(noise_floor)
    o.ls, o.os, o.noise = ptr(ls), ptr(os_), ptr(noise)
    o._keep = keep
    return o


def adam_inline(param, exp_avg, exp_avg_sq, scalars, segments, step_counter=None, loss_cum=None, beta1=0.9, beta2=0.999, next_feed=None):
    """argument block for hyper_bwd / mlp_bwd_hyper / mlp2_bwd_hyper(opt=...): param / exp_avg / exp_avg_sq [1, D], scalars = the
    PACOH_SC_ADAM block of the step's scalar row (device), segments = trained column ranges [(lo, hi), ...] (at most 4);
    next_feed = step_next(...): the pipelined feed rides along (fused networks only).  The object keeps references to the
    tensors: it holds raw pointers"""
    assert 1 <= len(segments) <= 4 and param.shape[0] == 1
    o = AdamInline()
    o.next = ctypes.pointer(next_feed) if next_feed is not None else None
    o._next = next_feed
    o.param, o.exp_avg, o.exp_avg_sq, o.scalars = (t.data_ptr() for t in (param, exp_avg, exp_avg_sq, scalars))
    o.beta1, o.beta2, o.n_seg = float(beta1), float(beta2), len(segments)
    for k, (lo, hi) in enumerate(segments):
        o.seg_lo[k], o.seg_hi[k] = int(lo), int(hi)
    o.step_counter = step_counter.data_ptr() if step_counter is not None else None
    o.loss_cum = loss_cum.data_ptr() if loss_cum is not None else None
    o._keep = (param, exp_avg, exp_avg_sq, scalars, step_counter, loss_cum)
    return o


def _opt_ptr(opt):
    return ctypes.cast(ctypes.pointer(opt), _vp) if opt is not None else None


def _svgd_bw(svgd_bw):
    """(workspace pointer, P, D) arguments of the optional bandwidth block"""
    if svgd_bw is None:
        return None, 0, 0
    ws, P, D = svgd_bw
    return _ptr(ws), int(P), int(D)


def softplus_fwd(raw, floor=0.0):
    lib = load_library()
    out = torch.empty_like(raw)
    with _Timed('softplus_fwd'):
        _check(lib.pacoh_softplus_fwd(_ptr(raw), _ptr(out), float(floor), raw.numel(), dtype_code(raw), _stream()),
               'pacoh_softplus_fwd')
    return out


def softplus_bwd(raw, g, d_raw=None, accumulate=False):
    lib = load_library()
    if d_raw is None:
        d_raw = torch.empty_like(raw)
        accumulate = False
    with _Timed('softplus_bwd'):
        _check(lib.pacoh_softplus_bwd(_ptr(raw), _ptr(g, raw), ctypes.c_void_p(d_raw.data_ptr()), int(bool(accumulate)),
                                      raw.numel(), dtype_code(raw), _stream()), 'pacoh_softplus_bwd')
    return d_raw


def hyper_fwd(theta, off_ls, f, off_os, off_noise, noise_floor, kernel=KERNEL_RBF):
    lib = load_library()
    P, D = theta.shape
    ls = torch.empty(P, f, dtype=theta.dtype, device=theta.device)
    os_ = torch.empty(P, dtype=theta.dtype, device=theta.device) if off_os >= 0 else None
    noise = torch.empty(P, dtype=theta.dtype, device=theta.device)
    with _Timed('hyper_fwd'):
        _check(lib.pacoh_hyper_fwd(_ptr(theta), D, P, off_ls, _kf(f, kernel), off_os, off_noise, float(noise_floor), _ptr(ls), _ptr(os_),
                                   _ptr(noise), dtype_code(theta), _stream()), 'pacoh_hyper_fwd')
    return ls, os_, noise


def hyper_bwd(theta, T, off_ls, f, off_os, off_noise, off_const, d_ls, d_os, d_noise, d_const, grad, lml=None, lik=None,
              lik_scale=1.0, info=None, fail_flag=None, kernel=KERNEL_RBF, svgd_bw=None, opt=None):
    """lml [T*P] and lik [P] (both or neither): lik[p] = lik_scale * sum_t lml[t, p] in the same launch;
    info [T*P] and fail_flag [1] (int32, both or neither): fail_flag |= any(info < 0);
    svgd_bw = (SVGD workspace, P, D), P <= 64: one more workgroup computes the SVGD step's median bandwidth from the distance matrix
    in that workspace into its bandwidth slot (svgd_update_next(bandwidth_ready=True))"""
    lib = load_library()
    P, D = theta.shape
    with _Timed('hyper_bwd'):
        _check(lib.pacoh_hyper_bwd(_ptr(theta), D, P, T, off_ls, _kf(f, kernel), off_os, off_noise, off_const, _ptr(d_ls, theta),
                                   _ptr(d_os, theta), _ptr(d_noise, theta), _ptr(d_const, theta), _ptr(grad, theta),
                                   grad.shape[1], _ptr(lml, theta), _ptr(lik, theta), float(lik_scale),
                                   _ptr(info if fail_flag is not None else None), _ptr(fail_flag if info is not None else None),
                                   *_svgd_bw(svgd_bw), _opt_ptr(opt), dtype_code(theta), _stream()), 'pacoh_hyper_bwd')


def prior_logprob_grad(theta, prior_mean, prior_std, grad=None, grad_scale=1.0, logp_out=None):
    lib = load_library()
    P, D = theta.shape
    logp = logp_out if logp_out is not None else torch.empty(P, dtype=theta.dtype, device=theta.device)
    with _Timed('prior_logprob_grad'):
        _check(lib.pacoh_prior_logprob_grad(_ptr(theta), _ptr(prior_mean, theta), _ptr(prior_std, theta), _ptr(logp),
                                            _ptr(grad, theta), float(grad_scale), P, D, dtype_code(theta), _stream()),
               'pacoh_prior_logprob_grad')
    return logp


def prior_score_dev(theta, prior_mean, prior_std, score, prior_factor, score_scale):
    """score := score_scale[0] * score + prior_factor * d log prior / d theta, the pre-factor in device memory (a captured step)"""
    P, D = theta.shape
    with _Timed('prior_logprob_grad'):
        _check(load_library().pacoh_prior_score_dev(_ptr(theta), _ptr(prior_mean, theta), _ptr(prior_std, theta), _ptr(score, theta),
                                                    float(prior_factor), _ptr(score_scale, theta), P, D, dtype_code(theta), _stream()),
               'pacoh_prior_score_dev')


def svgd_phi(X, score, bandwidth=None, neg=False, workspace=None):
    lib = load_library()
    P, D = X.shape
    code = dtype_code(X)
    need = lib.pacoh_svgd_workspace_bytes(P, D, code)
    if workspace is None or workspace.numel() * workspace.element_size() < need:
        workspace = torch.empty(need, dtype=torch.uint8, device=X.device)
    phi = torch.empty_like(X)
    bw_out = torch.empty(1, dtype=X.dtype, device=X.device)
    bw = -1.0 if bandwidth is None else float(bandwidth)
    with _Timed('svgd_phi'):
        _check(lib.pacoh_svgd_phi(_ptr(X), _ptr(score, X), bw, int(bool(neg)), _ptr(phi), _ptr(bw_out), _ptr(workspace),
                                  P, D, code, _stream()), 'pacoh_svgd_phi')
    return phi, bw_out, workspace


def svgd_update(X, score, prior_mean, prior_std, prior_factor, bandwidth, optimizer, lr, step, exp_avg, exp_avg_sq,
                workspace=None, beta1=0.9, beta2=0.999, eps=1e-8):
    """one fused SVGD step (RBF kernel): prior score + phi + optimizer -> (X_new, bandwidth, workspace)"""
    lib = load_library()
    P, D = X.shape
    code = dtype_code(X)
    need = lib.pacoh_svgd_workspace_bytes(P, D, code)
    if workspace is None or workspace.numel() * workspace.element_size() < need:
        workspace = torch.empty(need, dtype=torch.uint8, device=X.device)
    X_out = torch.empty_like(X)
    bw_out = torch.empty(1, dtype=X.dtype, device=X.device)
    bw = -1.0 if bandwidth is None else float(bandwidth)
    with _Timed('svgd_phi'):
        _check(lib.pacoh_svgd_update(_ptr(X), _ptr(score, X), _ptr(prior_mean, X), _ptr(prior_std, X), float(prior_factor), bw,
                                     int(optimizer == 'Adam'), float(lr), float(beta1), float(beta2), float(eps), int(step),
                                     _ptr(exp_avg, X), _ptr(exp_avg_sq, X), _ptr(X_out), _ptr(bw_out), _ptr(workspace), P, D, code,
                                     _stream()), 'pacoh_svgd_update')
    return X_out, bw_out, workspace


SC_SCORE_SCALE, SC_LR, SC_ADAM, SC_COUNT = 0, 1, 4, 8          # PACOH_SC_* of include/pacoh_gp.h
SVGD_MAX_PARTICLES = 1024                                      # PACOH_SVGD_MAX_PARTICLES (RBF and IMQ kernel)


def step_scalars(score_scale, lr, step, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0):
    """one row of step scalars (host side, float64): see pacoh_step_select in include/pacoh_gp.h"""
    row = [0.0] * SC_COUNT
    row[SC_SCORE_SCALE], row[SC_LR] = score_scale, lr
    row[SC_ADAM:SC_ADAM + 4] = adam_scalars(lr, step, beta1, beta2, eps, weight_decay)
    return row


def step_scalar_rows(score_scale, lrs, first_step, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0):
    """step_scalars() for k consecutive steps at once: score_scale [k] (or a float), lrs [k] -> float64 array [k, SC_COUNT]"""
    import numpy as np
    lrs = np.asarray(lrs, dtype=np.float64)
    k = lrs.shape[0]
    steps = first_step + np.arange(k, dtype=np.float64)
    rows = np.zeros((k, SC_COUNT), dtype=np.float64)
    rows[:, SC_SCORE_SCALE], rows[:, SC_LR] = score_scale, lrs
    rows[:, SC_ADAM] = 1.0 - lrs * weight_decay
    rows[:, SC_ADAM + 1] = lrs / (1.0 - beta1 ** steps)
    rows[:, SC_ADAM + 2] = (1.0 - beta2 ** steps) ** 0.5
    rows[:, SC_ADAM + 3] = eps
    return rows


def step_select(idx_all, sc_all, counter, idx_out, sc_out, aux_all=None, aux_out=None):
    """idx_out := idx_all[counter], sc_out := sc_all[counter], aux_out := aux_all[counter], counter += 1 (graph-capturable)"""
    lib = load_library()
    tb = idx_all.shape[1] if idx_all is not None else 0
    n_aux = aux_all[0].numel() if aux_all is not None else 0
    with _Timed('step_select'):
        _check(lib.pacoh_step_select(_ptr(idx_all), tb, _ptr(sc_all), sc_all.shape[1], _ptr(aux_all, sc_all), n_aux, _ptr(counter),
                                     _ptr(idx_out), _ptr(sc_out), _ptr(aux_out, sc_all), dtype_code(sc_all), _stream()), 'pacoh_step_select')


def svgd_update_workspace(X, workspace=None):
    """workspace of svgd_update_dev / of the distance part of step_begin for particles X"""
    lib = load_library()
    need = lib.pacoh_svgd_update_dev_workspace_bytes(X.shape[0], X.shape[1], dtype_code(X))
    if workspace is None or workspace.numel() * workspace.element_size() < need:
        workspace = torch.empty(need, dtype=torch.uint8, device=X.device)
    return workspace


def step_begin(feed, tasks, out, theta=None, hyper=None, hyper_out=None, advance=True, svgd=None):
    """first launch of a captured step: feed (engine.StepFeed) row -> feed.sc / feed.aux, task gather tasks[idx] -> out = (x, y, n_valid
    | None), hyper-parameter transforms of theta -> hyper_out = (ls, os | None, noise) with hyper = (off_ls, f, off_os, off_noise,
    noise_floor): ONE launch.  advance=False leaves the counter to the step's last launch (svgd_update_dev / adam_step_dev with
    step_counter=feed.ctr); svgd = (particles, workspace) adds their distance matrix + snapshot (svgd_update_dev(dist_done=True))"""
    lib = load_library()
    sv_X = sv_ws = None
    sv_P = sv_D = 0
    if svgd is not None:
        sv_X, sv_ws = svgd
        sv_P, sv_D = sv_X.shape
    tb = feed.tb
    n_aux = feed.aux_all[0].numel() if feed.aux_all is not None else 0
    x = y = nv = ox = oy = onv = None
    n = d = 0
    if tb > 0:
        x, y, nv = tasks.x, tasks.y, (tasks.n_valid if tasks.ragged else None)
        ox, oy, onv = out
        n, d = x.shape[1], x.shape[2]
    off_ls = f = off_os = off_noise = 0
    floor = 0.0
    ls = os_ = noise = None
    P, stride = 0, 0
    if theta is not None:
        off_ls, f, off_os, off_noise, floor = hyper[:5]
        f = _kf(f, hyper[5] if len(hyper) > 5 else KERNEL_RBF)
        ls, os_, noise = hyper_out
        P, stride = theta.shape
    with _Timed('step_begin'):
        _check(lib.pacoh_step_begin(_ptr(feed.idx_all), tb, _ptr(feed.sc_all), feed.sc_all.shape[1], _ptr(feed.aux_all, feed.sc_all), n_aux,
                                    _ptr(feed.ctr), _ptr(feed.ticket), _ptr(feed.sc), _ptr(feed.aux, feed.sc_all),
                                    _ptr(x, feed.sc_all), _ptr(y, feed.sc_all), _ptr(nv), _ptr(ox), _ptr(oy), _ptr(onv), n, d,
                                    _ptr(theta, feed.sc_all), stride, P, off_ls, f, off_os, off_noise, float(floor), _ptr(ls), _ptr(os_),
                                    _ptr(noise), int(bool(advance)), _ptr(sv_X, feed.sc_all), _ptr(sv_ws), sv_P, sv_D,
                                    dtype_code(feed.sc_all), _stream()), 'pacoh_step_begin')


def step_begin_vi(feed, tasks, out, posterior, theta_out, log_q_out, hyper=None, hyper_out=None, advance=False):
    """step_begin for a PACOH-VI step (diagonal posterior [2, D]): feed row -> feed.sc / feed.aux (the step's noise), task gather,
    AND the step's samples theta_out[S, D], log_q_out[S] and their transformed hyper-parameters hyper_out = (ls, os | None, noise)
    with hyper = (off_ls, f, off_os, off_noise, noise_floor, kernel): one launch instead of three"""
    lib = load_library()
    tb = feed.tb
    S, D = theta_out.shape
    x = y = nv = ox = oy = onv = None
    n = d = 0
    if tb > 0:
        x, y, nv = tasks.x, tasks.y, (tasks.n_valid if tasks.ragged else None)
        ox, oy, onv = out
        n, d = x.shape[1], x.shape[2]
    off_ls = f = off_os = off_noise = 0
    floor = 0.0
    ls = os_ = noise = None
    if hyper is not None:
        off_ls, f, off_os, off_noise, floor = hyper[:5]
        f = _kf(f, hyper[5] if len(hyper) > 5 else KERNEL_RBF)
        ls, os_, noise = hyper_out
    with _Timed('step_begin'):
        _check(lib.pacoh_step_begin_vi(_ptr(feed.idx_all), tb, _ptr(feed.sc_all), feed.sc_all.shape[1], _ptr(feed.aux_all, feed.sc_all),
                                       feed.aux_all[0].numel(), _ptr(feed.ctr), _ptr(feed.ticket), _ptr(feed.sc), _ptr(feed.aux, feed.sc_all),
                                       _ptr(x, feed.sc_all), _ptr(y, feed.sc_all), _ptr(nv), _ptr(ox), _ptr(oy), _ptr(onv), n, d,
                                       _ptr(posterior, feed.sc_all), S, D, _ptr(theta_out, feed.sc_all), _ptr(log_q_out, feed.sc_all),
                                       off_ls, f, off_os, off_noise, float(floor), _ptr(ls), _ptr(os_), _ptr(noise),
                                       int(bool(advance)), dtype_code(feed.sc_all), _stream()), 'pacoh_step_begin_vi')


def scale_dev(buf, scalar):
    """buf *= scalar[0] with the scalar in device memory"""
    with _Timed('scale_dev'):
        _check(load_library().pacoh_scale_dev(_ptr(buf), _ptr(scalar, buf), buf.numel(), dtype_code(buf), _stream()), 'pacoh_scale_dev')


def svgd_update_dev(X, score, prior_mean, prior_std, prior_factor, bandwidth, optimizer, scalars, exp_avg, exp_avg_sq,
                    workspace=None, bw_out=None, beta1=0.9, beta2=0.999, dist_done=False, step_counter=None):
    """the fused SVGD step with its step-dependent scalars in device memory; X is updated in place -> (bandwidth, workspace).
    dist_done: step_begin(svgd=(X, workspace)) already filled the workspace; step_counter: see step_begin(advance=False)"""
    lib = load_library()
    P, D = X.shape
    code = dtype_code(X)
    assert not dist_done or workspace is not None
    workspace = svgd_update_workspace(X, workspace)
    if bw_out is None:
        bw_out = torch.empty(1, dtype=X.dtype, device=X.device)
    bw = -1.0 if bandwidth is None else float(bandwidth)
    with _Timed('svgd_phi'):
        _check(lib.pacoh_svgd_update_dev(_ptr(X), _ptr(score, X), _ptr(prior_mean, X), _ptr(prior_std, X), float(prior_factor), bw,
                                         int(optimizer == 'Adam'), _ptr(scalars, X), float(beta1), float(beta2), _ptr(exp_avg, X),
                                         _ptr(exp_avg_sq, X), _ptr(bw_out), _ptr(workspace), P, D, int(bool(dist_done)),
                                         _ptr(step_counter), code, _stream()),
               'pacoh_svgd_update_dev')
    return bw_out, workspace


def svgd_dist_advance(X, workspace, counter):
    """pipelined SVGD step without a paired forward pass to ride in: *counter += 1, squared distances + snapshot of X -> workspace"""
    with _Timed('svgd_dist'):
        _check(load_library().pacoh_svgd_dist_advance(_ptr(X), _ptr(workspace), X.shape[0], X.shape[1], _ptr(counter), dtype_code(X),
                                                      _stream()), 'pacoh_svgd_dist_advance')


def svgd_update_next(X, score, prior_mean, prior_std, prior_factor, bandwidth, optimizer, exp_avg, exp_avg_sq, workspace, bw_out,
                     feed, tasks, hyper, beta1=0.9, beta2=0.999, bandwidth_ready=False):
    """the update of a pipelined SVGD step (include/pacoh_gp.h, "The pipelined SVGD step"): svgd_update_dev(dist_done) with the
    scalars of feed.sc2[counter & 1]; writes the updated particles' transformed hyper-parameters into feed.hyp with hyper =
    (off_ls, f, off_os, off_noise, noise_floor, kernel) and fetches the next row's scalars and task batch into feed.sc2 / feed.batch"""
    lib = load_library()
    P, D = X.shape
    bw = -1.0 if bandwidth is None else float(bandwidth)
    x = y = nv = ox = oy = onv = None
    n = d = 0
    if feed.tb > 0:
        x, y, nv = tasks.x, tasks.y, (tasks.n_valid if tasks.ragged else None)
        ox, oy, onv = feed.batch.x, feed.batch.y, feed.batch.n_valid
        n, d = x.shape[1], x.shape[2]
    off_ls, f, off_os, off_noise, floor, kernel = hyper
    ls, os_, noise = feed.hyp
    with _Timed('svgd_phi'):
        _check(lib.pacoh_svgd_update_next(_ptr(X), _ptr(score, X), _ptr(prior_mean, X), _ptr(prior_std, X), float(prior_factor), bw,
                                          int(optimizer == 'Adam'), float(beta1), float(beta2), _ptr(exp_avg, X), _ptr(exp_avg_sq, X),
                                          _ptr(bw_out), _ptr(workspace), P, D,
                                          _ptr(feed.ctr), _ptr(feed.sc2, X), feed.sc2.shape[1], _ptr(feed.idx_all), feed.tb,
                                          _ptr(feed.sc_all, X), _ptr(x, X), _ptr(y, X), _ptr(nv), _ptr(ox), _ptr(oy), _ptr(onv), n, d,
                                          off_ls, _kf(f, kernel), off_os, off_noise, float(floor), _ptr(ls), _ptr(os_), _ptr(noise),
                                          int(bool(bandwidth_ready)), dtype_code(X), _stream()), 'pacoh_svgd_update_next')


def svgd_imq_workspace(X, workspace=None):
    """scratch of pacoh_svgd_phi_imq for particles X[P, D] (uint8 tensor; an existing one is kept when it is large enough)"""
    need = load_library().pacoh_svgd_imq_workspace_bytes(X.shape[0], X.shape[1], dtype_code(X))
    if workspace is None or workspace.numel() * workspace.element_size() < need:
        workspace = torch.empty(need, dtype=torch.uint8, device=X.device)
    return workspace


def svgd_phi_imq(X, score, alpha=0.5, beta=-0.5, bandwidth=None, neg=False, workspace=None, phi_out=None, h_out=None):
    """-> (phi[P,D], h[D] | None (fixed bandwidth), workspace); phi_out / h_out: caller-owned result buffers (a captured step)"""
    lib = load_library()
    P, D = X.shape
    code = dtype_code(X)
    need = lib.pacoh_svgd_imq_workspace_bytes(P, D, code)
    if workspace is None or workspace.numel() * workspace.element_size() < need:
        workspace = torch.empty(need, dtype=torch.uint8, device=X.device)
    phi = phi_out if phi_out is not None else torch.empty_like(X)
    if bandwidth is not None:
        h_out = None
    elif h_out is None:
        h_out = torch.empty(D, dtype=X.dtype, device=X.device)
    bw = -1.0 if bandwidth is None else float(bandwidth)
    with _Timed('svgd_phi'):
        _check(lib.pacoh_svgd_phi_imq(_ptr(X), _ptr(score, X), float(alpha), float(beta), bw, int(bool(neg)), _ptr(phi),
                                      _ptr(h_out) if h_out is not None else None, _ptr(workspace), P, D, code, _stream()),
               'pacoh_svgd_phi_imq')
    return phi, h_out, workspace


def adam_step(param, grad, exp_avg, exp_avg_sq, lr, step, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0):
    lib = load_library()
    with _Timed('adam_step'):
        _check(lib.pacoh_adam_step(_ptr(param), _ptr(grad, param), _ptr(exp_avg, param), _ptr(exp_avg_sq, param),
                                   float(lr), float(beta1), float(beta2), float(eps), float(weight_decay), int(step),
                                   param.numel(), dtype_code(param), _stream()), 'pacoh_adam_step')


def adam_scalars(lr, step, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0):
    """the four step-dependent scalars of pacoh_adam_step_dev (host side, float64)"""
    return [1.0 - lr * weight_decay, lr / (1.0 - beta1 ** step), (1.0 - beta2 ** step) ** 0.5, eps]


def adam_step_dev(param, grad, exp_avg, exp_avg_sq, scalars, beta1=0.9, beta2=0.999, step_counter=None, loss_cum=None, loss=None):
    lib = load_library()
    with _Timed('adam_step'):
        _check(lib.pacoh_adam_step_dev(_ptr(param), _ptr(grad, param), _ptr(exp_avg, param), _ptr(exp_avg_sq, param),
                                       _ptr(scalars, param), float(beta1), float(beta2), param.numel(), _ptr(step_counter),
                                       _ptr(loss_cum, param), _ptr(loss, param), dtype_code(param), _stream()), 'pacoh_adam_step_dev')


def vi_update_dev(posterior, eps, theta, score, lik, log_q, prior_mean, prior_std, prior_factor, scalars, exp_avg, exp_avg_sq,
                  loss_out, workspace, step_counter=None, beta1=0.9, beta2=0.999):
    """update half of a PACOH-VI step (diagonal posterior, Adam) in one launch: pacoh_vi_update_dev in include/pacoh_gp.h;
    workspace = vi_update_workspace(posterior), allocated (zeroed) once"""
    lib = load_library()
    S, D = theta.shape
    with _Timed('vi_update'):
        _check(lib.pacoh_vi_update_dev(_ptr(posterior), _ptr(eps, posterior), _ptr(theta, posterior), _ptr(score, posterior),
                                       _ptr(lik, posterior), _ptr(log_q, posterior), _ptr(prior_mean, posterior),
                                       _ptr(prior_std, posterior), float(prior_factor), _ptr(scalars, posterior), float(beta1),
                                       float(beta2), _ptr(exp_avg, posterior), _ptr(exp_avg_sq, posterior), _ptr(loss_out, posterior),
                                       _ptr(step_counter), _ptr(workspace), S, D, dtype_code(posterior), _stream()), 'pacoh_vi_update_dev')


def vi_update_workspace(posterior):
    lib = load_library()
    need = lib.pacoh_vi_update_dev_workspace_bytes(posterior.shape[1], dtype_code(posterior))
    return torch.zeros(need, dtype=torch.uint8, device=posterior.device)


def axpy(y, x, alpha):
    lib = load_library()
    with _Timed('axpy'):
        _check(lib.pacoh_axpy(_ptr(y), _ptr(x, y), float(alpha), y.numel(), dtype_code(y), _stream()), 'pacoh_axpy')


def vi_sample(posterior, eps, full=False):
    """posterior [2, D] (diagonal) or, with full=True, [D+1, D] (full covariance: loc row, then tril_cov)"""
    lib = load_library()
    S, D = eps.shape
    assert posterior.shape == ((D + 1, D) if full else (2, D))
    theta = torch.empty_like(eps)
    log_q = torch.empty(S, dtype=eps.dtype, device=eps.device)
    fn, name = (lib.pacoh_vi_sample_full, 'pacoh_vi_sample_full') if full else (lib.pacoh_vi_sample, 'pacoh_vi_sample')
    with _Timed('vi_sample'):
        _check(fn(_ptr(posterior), _ptr(eps, posterior), _ptr(theta), _ptr(log_q), S, D, dtype_code(eps), _stream()), name)
    return theta, log_q


def vi_grad(posterior, eps, score, prior_factor, full=False):
    lib = load_library()
    S, D = eps.shape
    assert posterior.shape == ((D + 1, D) if full else (2, D))
    grad = torch.empty_like(posterior)
    fn, name = (lib.pacoh_vi_grad_full, 'pacoh_vi_grad_full') if full else (lib.pacoh_vi_grad, 'pacoh_vi_grad')
    with _Timed('vi_grad'):
        _check(fn(_ptr(posterior), _ptr(eps, posterior), _ptr(score, posterior), float(prior_factor),
                  _ptr(grad), S, D, dtype_code(eps), _stream()), name)
    return grad


def gather_tasks(x, y, n_valid, idx):
    """x[T,n,d], y[T,n], n_valid[T] int32, idx[Tb] int64 (device) -> (x[idx], y[idx], n_valid[idx]) in one launch"""
    lib = load_library()
    Tb = int(idx.numel())
    n, d = x.shape[1], x.shape[2]
    ox = torch.empty(Tb, n, d, dtype=x.dtype, device=x.device)
    oy = torch.empty(Tb, n, dtype=x.dtype, device=x.device)
    onv = torch.empty(Tb, dtype=torch.int32, device=x.device) if n_valid is not None else None
    with _Timed('gather_tasks'):
        _check(lib.pacoh_gather_tasks(_ptr(x), _ptr(y, x), _ptr(n_valid), _ptr(idx), _ptr(ox), _ptr(oy), _ptr(onv), Tb, n, d,
                                      dtype_code(x), _stream()), 'pacoh_gather_tasks')
    return ox, oy, onv


def reduce_tasks(inp, out, scale=1.0, accumulate=False):
    """out[P,W] (+)= scale * sum_t inp[T,P,W]"""
    lib = load_library()
    T, P, W = inp.shape
    with _Timed('reduce_tasks'):
        _check(lib.pacoh_reduce_tasks(_ptr(inp), ctypes.c_void_p(out.data_ptr()), float(scale), int(bool(accumulate)),
                                      T, P, W, dtype_code(inp), _stream()), 'pacoh_reduce_tasks')
    return out


def mixture_cdf(mu_n, var_n, value, y_mean, y_std):
    """cdf of the equal-weight mixture of the P un-normalised Gaussian marginals per test point.  One task: mu_n, var_n [P,m]
    (normalised space), value [m] -> [m]; a batch of tasks: [T,P,m], value [T,m] -> [T,m]"""
    lib = load_library()
    batched = mu_n.dim() == 3
    T, (P, m) = (mu_n.shape[0] if batched else 1), mu_n.shape[-2:]
    value = value.to(mu_n.dtype).reshape(T, m).contiguous()
    out = torch.empty(T, m, dtype=mu_n.dtype, device=mu_n.device)
    with _Timed('mixture_cdf'):
        _check(lib.pacoh_mixture_cdf(_ptr(mu_n), _ptr(var_n, mu_n), _ptr(value, mu_n), _ptr(out), float(y_mean), float(y_std), T, P, m,
                                     dtype_code(mu_n), _stream()), 'pacoh_mixture_cdf')
    return out if batched else out[0]


def mixture_icdf(mu_n, var_n, quantile, y_mean, y_std, closed_form=False, lo=-1e8, hi=1e8, eps=1e-6, max_iter=10000):
    """quantiles[m] of the same marginals: bisection with the reference's stopping rule in one launch, or (closed_form, P == 1)
    the Gaussian quantile"""
    lib = load_library()
    P, m = mu_n.shape
    quantile = quantile.to(mu_n.dtype).flatten().contiguous()
    if quantile.numel() != m:
        raise ValueError('one quantile per test point expected (%d), got %d' % (m, quantile.numel()))
    out = torch.empty(m, dtype=mu_n.dtype, device=mu_n.device)
    with _Timed('mixture_icdf'):
        _check(lib.pacoh_mixture_icdf(_ptr(mu_n), _ptr(var_n, mu_n), _ptr(quantile, mu_n), _ptr(out), float(y_mean), float(y_std),
                                      float(lo), float(hi), float(eps), int(max_iter), int(bool(closed_form)), P, m,
                                      dtype_code(mu_n), _stream()), 'pacoh_mixture_icdf')
    return out


def calib_error(cdf_vals):
    """calibration RMSE over the 20 confidence levels linspace(0.05, 0.95): cdf_vals [m] -> 0-dim tensor, [T,m] -> [T]"""
    lib = load_library()
    batched = cdf_vals.dim() == 2
    cdf_vals = cdf_vals.reshape(cdf_vals.shape[0] if batched else 1, -1).contiguous()
    T, m = cdf_vals.shape
    out = torch.empty(T, dtype=cdf_vals.dtype, device=cdf_vals.device)
    with _Timed('calib_error'):
        _check(lib.pacoh_calib_error(_ptr(cdf_vals), _ptr(out), T, m, dtype_code(cdf_vals), _stream()), 'pacoh_calib_error')
    return out if batched else out[0]


def comm_unique_id():
    """PACOH_COMM_ID_BYTES bytes naming a new RCCL communicator (rank 0 calls this and ships them to the other ranks)"""
    lib = load_library()
    buf = ctypes.create_string_buffer(COMM_ID_BYTES)
    _check(lib.pacoh_comm_unique_id(buf), 'pacoh_comm_unique_id')
    return buf.raw


def comm_init(uid, rank, world_size):
    """opaque communicator handle for this rank on the current HIP device"""
    lib = load_library()
    if len(uid) != COMM_ID_BYTES:
        raise ValueError('communicator id must be %d bytes' % COMM_ID_BYTES)
    handle = ctypes.c_void_p()
    _check(lib.pacoh_comm_init(ctypes.create_string_buffer(bytes(uid), COMM_ID_BYTES), int(rank), int(world_size),
                               ctypes.byref(handle)), 'pacoh_comm_init')
    return handle


def allreduce_sum(buf, comm):
    """buf := sum over ranks of buf, in place, enqueued on the current stream (RCCL)"""
    lib = load_library()
    with _Timed('allreduce_sum'):
        _check(lib.pacoh_allreduce_sum(_ptr(buf), buf.numel(), dtype_code(buf), comm, _stream()), 'pacoh_allreduce_sum')
    return buf


def comm_destroy(comm):
    _check(load_library().pacoh_comm_destroy(comm), 'pacoh_comm_destroy')
