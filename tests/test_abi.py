"""The C-ABI shared library loads and exports every symbol include/*.h declares (no compute calls: no GPU here)."""
import ctypes
import glob
import os
import re

import numpy as np
import pytest

from tests import cases
from tests.conftest import ROOT, has_gpu
from totalsegmentator2d_amd import _lib


def _declared_functions():
    names = []
    for h in glob.glob(os.path.join(ROOT, 'include', '*.h')):
        src = re.sub(r'/\*.*?\*/', '', open(h).read(), flags=re.S)
        names += re.findall(r'\b(ts2d_[a-z0-9_]+)\s*\(', src)
    return sorted(set(names))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    declared = _declared_functions()
    assert len(declared) >= 14
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/ts2d_engine.h but not exported"
    assert sorted(_lib.SYMBOLS) == declared            # the binding covers exactly the header
    assert lib.ts2d_abi_version() == _lib.ABI_VERSION


def test_arch_desc_layout_matches_header():
    assert ctypes.sizeof(_lib.ArchDesc) == 4 * 3 + 4 * 16 * 3 + 4 * 2 + 4 * 16 * 2       # ABI 7: + strides[16][2]
    assert _lib.ArchDesc.strides.offset == 4 * 3 + 4 * 16 * 3 + 4 * 2                      # ... appended: the ABI-6 fields keep their offsets


@pytest.mark.skipif(has_gpu(), reason='exercises the no-GPU error path')
def test_create_without_gpu_fails_loudly_not_silently():
    """No CPU fallback: on a machine without a GPU the product path raises with a retrievable message."""
    from totalsegmentator2d_amd.engine import Engine
    arch = cases.unet(2, (32, 32), 2)
    with pytest.raises(RuntimeError) as ei:
        Engine(arch, np.zeros(arch.n_params(), np.float32))
    assert 'ts2d_engine_create failed' in str(ei.value) and len(_lib.last_error()) > 0


def test_missing_library_raises(monkeypatch):
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', '/nonexistent/libts2d_engine.so')
    with pytest.raises(_lib.EngineLibraryError):
        _lib.load()
