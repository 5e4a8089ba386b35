"""Host-side logic of the batched env that needs no GPU: the lazily widened pose arrays of a big batch's Dict observation."""
import numpy as np
import pytest

from gym_novel_gridworlds_amd.vec_env import LazyObs


def _obs(n=3):
    o = LazyObs({'map': np.zeros((n, 4, 4), np.int8), 'agent_location': np.zeros((n, 2), np.int32), 'agent_facing_id': np.zeros(n, np.int32),
                 'inventory_items_quantity': np.zeros((n, 5), np.int32)})
    o._pose, o._dirty = np.zeros((n, 4), np.uint8), False
    return o


@pytest.mark.parametrize('how', ['getitem', 'dict', 'splat', 'update', 'items', 'values', 'copy', 'iter', 'savez'])
def test_lazy_obs_widens_the_pose_on_every_way_out(how, tmp_path):
    """dict(obs), {**obs}, other.update(obs) and np.savez(**obs) take CPython's fast merge path for dict subclasses unless __iter__ is
    overridden: every one of them must hand out the pose of the LAST step, not of whichever step somebody last indexed."""
    o = _obs()
    o._pose[:] = [[1, 2, 3, 0], [4, 5, 1, 0], [6, 7, 2, 0]]
    o._dirty = True
    if how == 'getitem':
        loc, fac = o['agent_location'], o['agent_facing_id']
    elif how == 'dict':
        d = dict(o); loc, fac = d['agent_location'], d['agent_facing_id']
    elif how == 'splat':
        d = {**o}; loc, fac = d['agent_location'], d['agent_facing_id']
    elif how == 'update':
        d = {}; d.update(o); loc, fac = d['agent_location'], d['agent_facing_id']
    elif how == 'items':
        d = {k: v for k, v in o.items()}; loc, fac = d['agent_location'], d['agent_facing_id']
    elif how == 'values':
        vals = list(o.values()); loc, fac = vals[1], vals[2]
    elif how == 'copy':
        d = o.copy(); loc, fac = d['agent_location'], d['agent_facing_id']
    elif how == 'iter':
        d = {k: dict.__getitem__(o, k) for k in o}; loc, fac = d['agent_location'], d['agent_facing_id']
    else:
        np.savez(tmp_path / 'o.npz', **o)
        z = np.load(tmp_path / 'o.npz'); loc, fac = z['agent_location'], z['agent_facing_id']
    assert (loc == [[1, 2], [4, 5], [6, 7]]).all() and (fac == [3, 1, 2]).all()
