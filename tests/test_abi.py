"""The C-ABI library loads on a machine without a GPU, exports every symbol include/multical_hip.h declares, and the
backend fails loudly (no CPU fallback) when no device is present."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "multical_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mc_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from multical_amd import _lib
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), n
    assert set(names) == set(_lib.SIGNATURES.keys())


def test_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from multical_amd.backend import HipApi, BackendError
    with pytest.raises(BackendError) as ei:
        HipApi()
    assert "no CPU fallback" in str(ei.value)
