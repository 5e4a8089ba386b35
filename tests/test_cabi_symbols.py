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
    assert L.ngw_abi_version() == ABI_VERSION == 3
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


def test_step_info_decodes_the_packed_words_on_demand():
    """StepInfo of a big batch carries the packed info words (include/ngw.h NGW_INFO_*) and decodes a field when it is read."""
    import numpy as np
    from gym_novel_gridworlds_amd.vec_env import StepInfo
    from gym_novel_gridworlds_amd.spec import STEP_COSTS
    rs = np.random.RandomState(0)
    result, done, cost = rs.randint(0, 2, 100), rs.randint(0, 2, 100), rs.randint(0, len(STEP_COSTS), 100)
    msg, arg = rs.randint(0, 16, 100), rs.randint(0, 1 << 12, 100)
    words = (result | (done << 1) | (cost << 2) | (msg << 8) | (arg << 16)).astype(np.uint32)
    info = StepInfo({'_words': words})
    assert 'result' in info and 'step_cost' in info and '_words' not in info.keys() and len(info) == 5
    assert (info['result'] == result.astype(bool)).all() and info['result'].dtype == np.bool_
    assert (info['step_cost_code'] == cost).all() and (info['message_code'] == msg).all() and (info['message_arg'] == arg).all()
    assert (info['step_cost'] == np.array([float(STEP_COSTS[c]) for c in cost])).all()
    assert set(dict(info.items())) == {'result', 'step_cost_code', 'message_code', 'message_arg', 'step_cost'}
    assert info.get('nothing', 7) == 7 and set(info.copy()) == set(info)
