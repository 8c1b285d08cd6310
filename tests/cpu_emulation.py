"""TEST-ONLY: run the product's HOST logic on the CPU by patching the HIP-op
wrappers of `deeprl_network_amd.ops` with the torch-CPU restatements of
oracle/ops_ref.py.  This is how `-m "not gpu"` tests cover the Python side
(buffer indexing, quirk ordering, parameter layout, checkpoint naming) in a
container without a GPU.  The product itself has no CPU path: outside this
context manager every op raises on CPU tensors.
"""
import contextlib

from deeprl_network_amd import ops
from oracle import ops_ref

_NAMES = ['nbr_gather', 'nbr_mean', 'nbr_onehot', 'lstm_cell', 'lstm_cell_infer', 'sample_actions',
          'nstep_return', 'rmsprop_tf_clip']


@contextlib.contextmanager
def cpu_ops():
    saved = {n: getattr(ops, n) for n in _NAMES}
    try:
        for n in _NAMES:
            setattr(ops, n, getattr(ops_ref, n))
        yield
    finally:
        for n, f in saved.items():
            setattr(ops, n, f)
