"""MI355X-native PACOH task-GP hot path (drop-in for the reference's meta_learn learners).

Host code in Python, arithmetic in hand-written HIP kernels (csrc/) behind the C ABI of
include/pacoh_gp.h, bound with ctypes (_lib.py).  No CPU fallback.
"""
__version__ = '0.1.0'

from .GPR_meta_mll import GPRegressionMetaLearned          # noqa: E402,F401
from .GPR_meta_svgd import GPRegressionMetaLearnedSVGD     # noqa: E402,F401
from .GPR_meta_vi import GPRegressionMetaLearnedVI         # noqa: E402,F401
from .GPR_mll import GPRegressionLearned                   # noqa: E402,F401
