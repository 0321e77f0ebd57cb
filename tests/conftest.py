import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def has_gpu() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if has_gpu():
        return
    skip = pytest.mark.skip(reason='no GPU visible')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


def golden(name):
    return np.load(os.path.join(GOLDEN, f'{name}.npz'))


_blob_cache = {}


def blob_for(arch, seed):
    """Synthetic weight blob, cached per process (the canonical net takes ~10 s to generate)."""
    from totalsegmentator2d_amd import weights
    key = (repr(arch), seed)
    if key not in _blob_cache:
        sd = weights.synthetic_state_dict(arch, seed)
        _blob_cache[key] = (sd, weights.pack_blob(arch, sd))
    return _blob_cache[key]
