import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: test needs a real MI355X (run with -m gpu on the GPU box)')
    # the hosts show more cores than the container may use (cgroup quota): keep torch's intra-op pool inside the budget, or the
    # spinning workers get the test process throttled (meta_learning_pacoh_amd.util.host_cpu_budget)
    import torch
    from meta_learning_pacoh_amd.util import host_cpu_budget
    torch.set_num_threads(max(1, min(torch.get_num_threads(), host_cpu_budget(), 8)))


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _fixed_torch_seed():
    """every test starts from the same torch CPU generator state (a few tests draw inputs without their own generator)"""
    import torch
    torch.manual_seed(1234)
    yield
