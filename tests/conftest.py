import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu via gpurun)')
    # the binding refuses a missing or stale libnmarl_hip.so: (re)build it once per session when the sources changed
    # (hipcc cross-compiles gfx950 without a GPU; on the GPU box the shipped library matches and nothing is built)
    from deeprl_network_amd import build
    if build.stale() and os.path.exists(os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')):
        build.build_native(verbose=False)
    # the in-process CPU tests run tiny per-agent ops: with the default thread pool (one thread per core) a busy host makes
    # them spin (measured: 34 min instead of 5 for the CPU suite next to two single-core jobs).  The fixture generators that
    # the regeneration tests spawn keep the default (their float64 sums depend on the thread count in the last bits).
    import torch
    torch.set_num_threads(max(1, min(2, int(os.environ.get('NMARL_TEST_THREADS', '2')))))


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN
