"""Randomised call sequences over every test configuration, HIP path (through the C-ABI) against the CPU oracle.

Each case draws a configuration, a batch size, a horizon, a prepared-episode setting and a sequence of calls - host steps,
device steps, fused rollouts (generated and supplied actions), masked and full resets, state injection - and compares the
whole state with the oracle's after every call.  The default budget is a few seconds per chunk of configurations;
NGW_FUZZ_SECONDS=<s> runs longer (a soak on the GPU box), NGW_FUZZ_SEED=<n> draws other sequences.  Needs an MI355X."""
import os
import time

import numpy as np
import pytest

import ngw_testlib as T
from gym_novel_gridworlds_amd import VecNovelGridworld
from oracle.ngw_oracle import Oracle

pytestmark = pytest.mark.gpu
STATE_KEYS = ('map', 'loc', 'facing', 'inv', 'selected', 'step_count', 'episode')
CFGS = sorted(T.CFGS)
BUDGET = float(os.environ.get('NGW_FUZZ_SECONDS', '0'))
SEED = int(os.environ.get('NGW_FUZZ_SEED', '1000'))


def check(v, o, where, lid=None):
    if lid is not None:                                    # fused LidarInFront epilogue: the observation every launch left behind
        from oracle.ngw_oracle import lidar
        cc, S, K = lid
        got = v.lidar_observation()
        got = v.lidar_widen(got) if isinstance(got, tuple) else got
        exp = lidar(cc, S, K, o.st.map, o.st.loc, o.st.facing, o.st.inv)
        bad = np.nonzero((got != exp).any(1))[0]
        assert bad.size == 0, "%s: lidar rows differ for %d envs, first env %d" % (where, bad.size, bad[0])
    hs = v.get_state()
    st = o.st
    os_ = dict(map=st.map, loc=st.loc, facing=st.facing, inv=st.inv, selected=st.selected, step_count=st.step_count, episode=st.episode)
    for k in STATE_KEYS:
        bad = np.nonzero((hs[k] != os_[k]).reshape(len(hs[k]), -1).any(1))[0]
        assert bad.size == 0, "%s: %s differs for %d envs, first env %d" % (where, k, bad.size, bad[0])


def both_reset(v, o, mask, tag):
    """reset() on both sides; a placement that cannot succeed ("Cannot place items, increase map size!", pogostick_v1_env.py:167)
    must fail on both - the case ends there."""
    try:
        v.reset(mask)
    except AssertionError:
        assert (o.reset(mask) if mask is not None else o.reset()) != 0, tag + ': only the HIP path failed to place'
        return False
    assert (o.reset(mask) if mask is not None else o.reset()) == 0, tag + ': only the oracle failed to place'
    return True


def one_case(rs, cfg, case):
    import torch
    spec = T.build_spec(cfg)
    A = len(spec.actions_id)
    S = spec.map_size
    n = int(rs.choice([1, 37, 64, 65, 128, 200, 513, 1024, 1500])) if S <= 16 else int(rs.choice([1, 64, 130, 256, 300]))   # (whole-wavefront batches: the write-through step also delivers the lidar rows)
    horizon = int(rs.choice([0, 7, 23, 64, 90]))
    autoreset = bool(rs.randint(0, 4)) or horizon > 0
    prefetch = rs.choice(['auto', 0, 3, 16])
    prefetch = prefetch if prefetch == 'auto' else int(prefetch)
    seed, base = int(rs.randint(0, 2 ** 31)), int(rs.randint(0, 10 ** 6))
    depth = int(rs.choice([0, 0, 1, 2, 4]))                 # prepared episodes per env (0 = automatic)
    term = bool(rs.randint(0, 5) == 0)                      # terminal-observation capture on (the cold paths copy the rows an episode ended in first)
    # half the cases take the big-batch form of the host step (one page-locked block, delta refresh) at these small sizes too
    zc = os.environ.pop('NGW_ZC_BYTES', None)
    small_block = bool(rs.randint(0, 2))
    if small_block:
        os.environ['NGW_ZC_BYTES'] = '2048'
    try:
        v = _make(spec, n, seed, autoreset, horizon, prefetch, base, depth, term)
    finally:
        os.environ.pop('NGW_ZC_BYTES', None)
        if zc is not None:
            os.environ['NGW_ZC_BYTES'] = zc
    return _run_case(rs, cfg, case, spec, v, n, A, S, horizon, autoreset, prefetch, depth, seed, base, small_block)


def _make(spec, n, seed, autoreset, horizon, prefetch, base, depth, term=False):
    return VecNovelGridworld(spec=spec, num_envs=n, seed=seed, autoreset=autoreset, horizon=horizon, reset_prefetch=prefetch, env_index_base=base,
                          reset_prefetch_depth=depth, terminal_capture=term)


def _run_case(rs, cfg, case, spec, v, n, A, S, horizon, autoreset, prefetch, depth, seed, base, small_block):
    import torch
    o = Oracle(spec.compile(), n, seed=seed, autoreset=autoreset, horizon=horizon, env_index_base=base)
    lid = None
    if rs.randint(0, 3) == 0:                               # one case in three runs with the fused lidar epilogue
        from gym_novel_gridworlds_amd.lidar import LidarConfig
        lc = LidarConfig(spec, int(rs.choice([4, 8, 8])))
        fmt = [np.int16, np.int32, 'packed'][int(rs.randint(0, 3))]
        v.lidar_configure(lc, fused=True, dtype=fmt)
        lid = (lc.compile(spec), S, len(spec.items_id))
    tag = '%s case %d (n=%d H=%d auto=%d prefetch=%s depth=%d lidar=%d block=%d)' % (cfg, case, n, horizon, autoreset, prefetch, depth, lid is not None, small_block)
    if not both_reset(v, o, None, tag):
        v.close()
        return
    check(v, o, tag + ' reset', lid)
    t_roll = 0
    ofail = [0]                                            # the oracle's own "cannot place" reports (same-step autoreset inside step / rollout)

    def ostep(a):
        ofail[0] |= int(o.step(a) != 0)

    try:
        for call in range(int(rs.randint(4, 10))):
            kind = rs.choice(['step', 'step', 'dev', 'rollout', 'rollact', 'mask', 'reset', 'state'])
            if not autoreset and kind in ('rollout', 'rollact') and rs.randint(0, 2):
                kind = 'step'
            if kind == 'step':
                for _ in range(int(rs.randint(1, 12))):
                    a = rs.randint(0, A, size=n).astype(np.int32)
                    ostep(a)
                    obs, reward, done, info = v.step(a)
                    assert (reward == o.reward).all() and (done == o.done.astype(bool)).all() and (info['message_code'] == o.msg_code).all(), tag
                    if not ofail[0]:
                        assert (obs['map'].reshape(n, -1) == o.st.map).all() and (obs['inventory_items_quantity'] == o.st.inv).all(), tag + ': host observation'
                        assert (obs['agent_location'] == o.st.loc).all() and (obs['agent_facing_id'] == o.st.facing).all(), tag + ': host observation'
                        if lid is not None:                 # the rows the host step itself delivered (write-through / copy behind the launch)
                            from oracle.ngw_oracle import lidar as _lidar
                            got = v.lidar_observation()
                            got = v.lidar_widen(got) if isinstance(got, tuple) else got
                            assert (got == _lidar(lid[0], lid[1], lid[2], o.st.map, o.st.loc, o.st.facing, o.st.inv)).all(), tag + ': lidar rows of the host step'
            elif kind == 'dev':
                k = int(rs.randint(1, 9))
                an = rs.randint(0, A, size=(k, n)).astype(np.int32)
                acts = torch.from_numpy(an).cuda()
                torch.cuda.synchronize()
                for i in range(k):
                    ostep(an[i])
                v.step_device_many(acts.data_ptr(), n, k)
            elif kind == 'rollout':
                k = int(rs.randint(1, 70))
                ofail[0] |= int(o.rollout(k, seed ^ 77, t_roll) != 0)
                v.rollout(k, action_seed=seed ^ 77, t0=t_roll)
                t_roll += k
            elif kind == 'rollact':
                k = int(rs.randint(1, 40))
                an = rs.randint(0, A, size=(k, n)).astype(np.int32)
                acts = torch.from_numpy(an).cuda()
                torch.cuda.synchronize()
                for i in range(k):
                    ostep(an[i])
                v.rollout_actions(acts.data_ptr(), n, k)
            elif kind in ('mask', 'reset'):
                m = (rs.randint(0, 3, size=n) == 0).astype(np.uint8) if kind == 'mask' else None
                if not both_reset(v, o, m, tag):
                    break
            else:                                          # inject step counts (and with them where the horizons fall)
                sc = rs.randint(0, max(horizon, 5), size=n).astype(np.int32)
                v.set_state(0, step_count=sc); o.st.step_count[:] = sc
            if ofail[0]:                                   # an in-step reset could not place its items: the HIP path must say so too
                with pytest.raises(AssertionError, match='Cannot place items'):
                    v.get_state(); v._raise_flags()
                break
            check(v, o, '%s call %d %s' % (tag, call, kind), None if kind == 'state' else lid)
        else:
            assert v.error_flags() == 0, tag
    except AssertionError as ex:
        if 'Cannot place items' not in str(ex) or not ofail[0]:
            raise
    v.close()


@pytest.mark.parametrize('chunk', range(6))
def test_random_call_sequences_match_oracle(chunk):
    rs = np.random.RandomState(SEED + chunk)
    mine = CFGS[chunk::6]
    t_end = time.time() + (BUDGET / 6 if BUDGET > 0 else 0)
    case = 0
    while True:
        for cfg in mine:
            one_case(rs, cfg, case)
            case += 1
        if time.time() >= t_end:
            break
