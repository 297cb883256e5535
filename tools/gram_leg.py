"""bench.py's config-5 HBM report alone: the fp64 Gram leg (public entry point, one point set) and the committed traffic ratios"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from meta_learning_pacoh_amd import _lib as L  # noqa: E402

print(json.dumps(bench.cfg5_hbm_report(L)['gram']))
