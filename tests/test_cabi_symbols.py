"""The C-ABI library loads without a GPU and exports every symbol include/ngw.h declares (no compute calls here)."""
import ctypes as C
import os
import re

import pytest

from gym_novel_gridworlds_amd import _cabi
from gym_novel_gridworlds_amd.spec import NgwSpec

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, 'include', 'ngw.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(ngw_[a-z_0-9]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    L = _cabi.lib()
    declared = header_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(L, name), name
    assert sorted(_cabi.SYMBOLS) == declared


def test_abi_version_and_spec_layout():
    L = _cabi.lib()
    from gym_novel_gridworlds_amd.spec import ABI_VERSION
    assert L.ngw_abi_version() == ABI_VERSION == 2
    assert L.ngw_spec_size() == C.sizeof(NgwSpec)


def test_no_cpu_fallback_without_gpu():
    """Without a visible GPU the product path must fail loudly instead of computing somewhere else."""
    L = _cabi.lib()
    if L.ngw_device_count() > 0:
        pytest.skip("a GPU is visible")
    from gym_novel_gridworlds_amd import VecNovelGridworld
    with pytest.raises(_cabi.NgwError, match="no HIP device visible"):
        VecNovelGridworld(num_envs=8)


def test_product_package_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, 'gym_novel_gridworlds_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.cpp', '.hip', '.h')):
                text = open(os.path.join(dirpath, f)).read()
                assert 'ngw_oracle' not in text and 'from oracle' not in text and 'import oracle' not in text, f
