"""VecNovelGridworld - N concurrent environments on one MI355X behind the reference's step()/reset() contract.

Batched mirror of the reference `gym.Env` surface (gym_novel_gridworlds/envs/pogostick_v1_env.py):
`reset()` :86, `step(action_id)` :230 -> `(obs, reward, done, info)`, `get_observation()` :214 with the Dict
observation of map / agent_location / agent_facing_id / inventory_items_quantity, plus `inject_novelty`
semantics through `novelty=`.  All computation happens in the HIP kernels behind the C-ABI (`_cabi.py`)."""
import ctypes as C

import numpy as np

from . import _cabi
from .novelty import apply_novelty
from .spec import F_INVALID_ACTION, F_PLACEMENT, STEP_COSTS, EnvSpec, make_spec

_COST_F64 = np.array([float(c) for c in STEP_COSTS], np.float64)
PLACEMENT_MESSAGE = "Cannot place items, increase map size!"          # pogostick_v1_env.py:167


class _DevArray:
    """Zero-copy view of a device buffer for torch.as_tensor (CUDA array interface, also honoured on ROCm)."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {'shape': tuple(shape), 'typestr': typestr, 'data': (int(ptr), False),
                                         'version': 2, 'strides': None}


class StepInfo(dict):
    """info of a batched step: 'result' bool[N], 'step_cost_code', 'message_code', 'message_arg' - and 'step_cost' f64[N].  Fields
    that are not there yet are made when they are first asked for: 'step_cost' is looked up from the codes (a 65 536-element
    gather per step costs more than the step kernel; few callers read it), and on big batches all four code arrays are
    decoded from the packed info words ('_words', include/ngw.h NGW_INFO_*) only then - the C-side decode of 65 536 words
    costs 40 us per step, and a rollout loop reads reward and done.  Behaves as a dict that has the keys."""
    __slots__ = ()
    LAZY = ('result', 'step_cost_code', 'message_code', 'message_arg', 'step_cost')

    def __missing__(self, key):
        if key == 'final_observation' and dict.__contains__(self, '_final_fn'):      # terminal observations (set_terminal_capture): fetched when asked for
            v = dict.__getitem__(self, '_final_fn')()
            dict.__setitem__(self, key, v)
            return v
        if key in StepInfo.LAZY:
            if key == 'step_cost':
                v = _COST_F64[self['step_cost_code']]
            else:
                w = dict.__getitem__(self, '_words')
                v = {'result': lambda: (w & 1).astype(np.bool_), 'step_cost_code': lambda: ((w >> 2) & 63).astype(np.uint8),
                     'message_code': lambda: ((w >> 8) & 255).astype(np.uint16), 'message_arg': lambda: (w >> 16).astype(np.uint16)}[key]()
            dict.__setitem__(self, key, v)
            return v
        raise KeyError(key)

    def __contains__(self, key):
        return key in StepInfo.LAZY or dict.__contains__(self, key) or (key == 'final_observation' and dict.__contains__(self, '_final_fn'))

    def get(self, key, default=None):
        return self[key] if key in self else default

    def _all(self):
        for k in StepInfo.LAZY:
            self[k]
        return self

    def keys(self):
        return [k for k in dict.keys(self._all()) if k not in ('_words', '_final_fn')]

    def items(self):
        return [(k, v) for k, v in dict.items(self._all()) if k not in ('_words', '_final_fn')]

    def values(self):
        return [v for k, v in dict.items(self._all()) if k not in ('_words', '_final_fn')]

    def __iter__(self):
        return iter(self.keys())

    def __len__(self):
        return len(self.keys())

    def copy(self):
        return StepInfo(dict(self.items()))


class LazyObs(dict):
    """The Dict observation of a big batch's step(): 'map' and 'inventory_items_quantity' are sections of the page-locked block the
    step refreshes; the agent's pose crosses PCIe as four bytes per env (ngw_step_host_packed) and 'agent_location' [N, 2] int32 /
    'agent_facing_id' [N] int32 are widened from those bytes into the same two arrays when somebody reads them (a 65 536-env widening
    costs more than the step kernel; a loop that feeds a policy from the map or the lidar rows never asks).  A dict in every other way."""
    __slots__ = ('_pose', '_dirty')

    def _sync(self):
        if self._dirty:
            self._dirty = False
            pose = self._pose
            loc = dict.__getitem__(self, 'agent_location')
            loc[:, 0] = pose[:, 0]
            loc[:, 1] = pose[:, 1]
            dict.__getitem__(self, 'agent_facing_id')[:] = pose[:, 2]

    def __getitem__(self, key):
        if key != 'map' and key != 'inventory_items_quantity':
            self._sync()
        return dict.__getitem__(self, key)

    def get(self, key, default=None):
        return self[key] if key in self else default

    def items(self):
        self._sync()
        return dict.items(self)

    def values(self):
        self._sync()
        return dict.values(self)

    def copy(self):
        self._sync()
        return dict(dict.items(self))

    # dict(obs), {**obs}, other.update(obs), np.savez(**obs): CPython merges a dict SUBCLASS through the fast path that bypasses
    # __getitem__ unless the subclass overrides __iter__ - with these two the merge walks keys() + __getitem__, which sync
    def __iter__(self):
        self._sync()
        return dict.__iter__(self)

    def keys(self):
        self._sync()
        return dict.keys(self)


class VecNovelGridworld:
    def __init__(self, env_id='NovelGridworld-Pogostick-v1', num_envs=1, map_size=None, novelty=None, device=0,
                 seed=0, autoreset=False, horizon=0, env_index_base=0, spec=None, reset_prefetch='auto', reset_prefetch_depth=0,
                 terminal_capture=False):
        if spec is None:
            spec = make_spec(env_id, map_size)
            if novelty:
                novs = [novelty] if isinstance(novelty[0], str) else list(novelty)
                for nv in novs:
                    apply_novelty(spec, *nv)
        assert isinstance(spec, EnvSpec)
        self.num_envs = int(num_envs)
        self.device = int(device)
        self.seed = int(seed)
        self.env_index_base = int(env_index_base)
        self.autoreset, self.horizon = bool(autoreset), int(horizon)
        # what the caller chose for the prepared next episodes ('auto' / 0 = the library's own defaults); rebuild() re-applies it
        self._prefetch_arg, self._depth_arg = reset_prefetch, int(reset_prefetch_depth)
        self.terminal_capture = bool(terminal_capture)         # keep the observation an episode ENDED in (set_terminal_capture)
        self.lidar, self.lidar_fused, self.lidar_len, self.lidar_dtype, self.lidar_packed = None, False, 0, np.dtype(np.int16), False   # set by lidar_configure()
        self._h = C.c_void_p()
        self._open(spec)

    def _open(self, spec):
        """Create the device handle for `spec` and apply this env's settings to it."""
        self.spec = spec
        self.cspec = spec.compile()
        self.map_size = spec.map_size
        self.n_items = len(spec.items_id)
        self.items_id = dict(spec.items_id)
        self.actions_id = dict(spec.actions_id)
        self.single_action_space_n = spec.action_space_n
        from . import spaces
        self.action_space = spaces.Discrete(spec.action_space_n)           # per env; NOT grown by axe/additem
        self.observation_space = spaces.Dict({'map': spaces.Box(low=0, high=spec.max_items, shape=(spec.map_size, spec.map_size, 1))})
        import os
        self._zc_bytes = int(os.environ.get('NGW_ZC_BYTES', 256 << 10)) or 1     # (read where ngw_create reads it)
        L = _cabi.lib()
        _cabi.check(L.ngw_create(C.byref(self.cspec), self.num_envs, self.device, self.seed, self.env_index_base,
                                 C.byref(self._h)))
        _cabi.check(L.ngw_set_autoreset(self._h, int(self.autoreset), self.horizon))
        # 'auto' = the library's own default: ngw_set_autoreset switched prepared next episodes on (a refill every 3/4 horizon)
        # unless the episodes are too short for that cadence to keep up (rows would go stale and resets simply run inline)
        self.reset_prefetch = self._default_prefetch()
        self._prefetch_user = False
        if self._depth_arg:
            self.set_reset_prefetch_depth(self._depth_arg)
        if self._prefetch_arg != 'auto':
            self.set_reset_prefetch(self._prefetch_arg)
        self._flags_word = C.c_uint32(0)
        self._host = None                                     # host mirrors of the host API: allocated on first use
        if self.__dict__.get('_stream_arg'):                  # the caller's stream and output bindings travel with the env (rebuild)
            self.set_stream(self._stream_arg)
        if self.__dict__.get('_rollout_out_args'):
            self.rollout_outputs(*self._rollout_out_args)
        if self.terminal_capture:
            self.set_terminal_capture(True)
        if self.lidar is not None:                            # the observation setup travels with the env (rebuild)
            self.lidar_configure(self.lidar, fused=self.lidar_fused, dtype='packed' if self.lidar_packed else self.lidar_dtype)

    def rebuild(self, spec):
        """The same batched env - same object, same shard of the global env index space (`env_index_base`), same autoreset /
        horizon / prepared-episode / lidar / terminal-capture settings, same stream and rollout-output bindings - on an edited spec.
        In place: the reference's novelty wrappers mutate the env they wrap and keep its identity (novelty_wrappers.py:1586-1674), so a
        rank-local shard stays that shard.  The NEW handle is created first: a spec the library refuses (ngw_create's checks) raises
        and leaves the env as it was - the reference, too, asserts before it changes anything.  The state is undefined until the
        next reset(), as after construction."""
        old_h, old_attrs = self._h, dict(self.__dict__)
        for name in VecNovelGridworld._HOST_ATTRS + ('_host', '_step_args', '_step1_args', '_step1_fn', '_step1_mv', '_state1_mv', '_reset1_args', '_last_state_views', '_lidar_host', '_view_host', '_last_actions', '_packed_block', '_packed_call', '_steps_stale'):
            self.__dict__.pop(name, None)
        self._h = C.c_void_p()
        try:
            self._open(spec)
        except Exception:
            if self._h:
                _cabi.lib().ngw_destroy(self._h)
            self.__dict__.clear()
            self.__dict__.update(old_attrs)                   # the old handle, spec and host mirrors: nothing happened
            raise
        if old_h:
            _cabi.lib().ngw_destroy(old_h)
        return self

    _HOST_ATTRS = ('_obs', '_reward', '_done', '_act_pinned', '_sel_host', '_steps_host', '_result', '_cost', '_msg', '_arg', '_flags_np', '_info_words')

    def __getattr__(self, name):
        # The host mirrors (10 MB page-locked at 65 536 envs, 157 B per env) exist only for the host API; a handle that is
        # only ever driven through device pointers (step_device / rollout / device_observation) never pays for them.
        if name in VecNovelGridworld._HOST_ATTRS:
            if self.__dict__.get('_host') is None:
                N, S, K = self.num_envs, self.map_size, self.n_items
                if self._one_block_path():
                    self._make_packed_host()
                    return self.__dict__[name]
                # ONE page-locked block laid out as ngw_host_step_layout says: a big batch's step() then comes back with a single
                # copy across PCIe (one pack launch on the device) instead of nine
                offs = (C.c_uint64 * 11)()
                _cabi.check(_cabi.lib().ngw_host_step_layout(self._h, offs))
                block = _cabi.pinned_array((int(offs[10]),), np.uint8)

                def sec(i, shape, dt):
                    nb = int(np.prod(shape)) * np.dtype(dt).itemsize
                    return block[int(offs[i]):int(offs[i]) + nb].view(dt).reshape(shape)
                self.__dict__['_host'] = dict(
                    _obs={'map': sec(0, (N, S, S), np.int8), 'agent_location': sec(1, (N, 2), np.int32),
                          'agent_facing_id': sec(2, (N,), np.int32), 'inventory_items_quantity': sec(3, (N, K), np.int32)},
                    _reward=sec(4, (N,), np.int32), _done=sec(5, (N,), np.uint8), _flags_np=sec(7, (1,), np.uint32),
                    # the packed info words of the last step(): big batches (the one-block path of ngw_step_host) decode them lazily
                    _info_words=sec(6, (N,), np.uint32) if self._one_block_path() else None,
                    _sel_host=sec(8, (N,), np.uint8), _steps_host=sec(9, (N,), np.int32),     # selected item / step_count after the last step()
                    _act_pinned=_cabi.pinned_array((N,), np.int32),
                    _result=np.zeros(N, np.uint8), _cost=np.zeros(N, np.uint8), _msg=np.zeros(N, np.uint16), _arg=np.zeros(N, np.uint16))
                self.__dict__.update(self.__dict__['_host'])   # plain attributes from now on (no __getattr__ detour per access)
            return self.__dict__[name]
        raise AttributeError(name)

    def _make_packed_host(self):
        """Host mirrors of a big batch: ONE page-locked block in the narrow wire format (include/ngw.h ngw_host_step_layout_packed) -
        map and inventory refreshed by deltas, pose as four bytes per env, reward int32, done uint8, packed info words."""
        N, S, K = self.num_envs, self.map_size, self.n_items
        offs = (C.c_uint64 * 8)()
        _cabi.check(_cabi.lib().ngw_host_step_layout_packed(self._h, offs))
        block = _cabi.pinned_array((int(offs[7]),), np.uint8)

        def sec(i, shape, dt):
            nb = int(np.prod(shape)) * np.dtype(dt).itemsize
            return block[int(offs[i]):int(offs[i]) + nb].view(dt).reshape(shape)
        pose = sec(2, (N, 4), np.uint8)
        obs = LazyObs({'map': sec(0, (N, S, S), np.int8), 'agent_location': np.zeros((N, 2), np.int32),
                       'agent_facing_id': np.zeros(N, np.int32), 'inventory_items_quantity': sec(1, (N, K), np.int32)})
        obs._pose, obs._dirty = pose, False
        self.__dict__['_host'] = dict(
            _obs=obs, _reward=sec(3, (N,), np.int32), _done=sec(4, (N,), np.uint8), _info_words=sec(5, (N,), np.uint32), _flags_np=sec(6, (1,), np.uint32),
            _sel_host=pose[:, 3], _steps_host=np.zeros(N, np.int32), _act_pinned=np.zeros(N, np.int32),
            _result=np.zeros(N, np.uint8), _cost=np.zeros(N, np.uint8), _msg=np.zeros(N, np.uint16), _arg=np.zeros(N, np.uint16))
        self.__dict__.update(self.__dict__['_host'])
        self.__dict__['_packed_block'] = block
        self.__dict__['_steps_stale'] = False

    # ------------------------------------------------------------------ lifecycle
    def close(self):
        if getattr(self, '_h', None) is not None and self._h:
            _cabi.lib().ngw_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:       # noqa: BLE001 - interpreter shutdown
            pass

    def set_autoreset(self, autoreset, horizon=0):
        self.autoreset, self.horizon = bool(autoreset), int(horizon)
        _cabi.check(_cabi.lib().ngw_set_autoreset(self._h, int(self.autoreset), self.horizon))
        if not self.__dict__.get('_prefetch_user'):
            self.reset_prefetch = self._default_prefetch()

    def set_stream(self, hip_stream_ptr):
        _cabi.check(_cabi.lib().ngw_set_stream(self._h, C.c_void_p(hip_stream_ptr)))
        self._stream_arg = hip_stream_ptr                     # (rebuild() puts the new handle on the same stream)

    def stream_order(self, other_stream_ptr, handle_waits):
        """Order this env's stream and another HIP stream behind each other without a host wait (include/ngw.h ngw_stream_order)."""
        _cabi.check(_cabi.lib().ngw_stream_order(self._h, C.c_void_p(int(other_stream_ptr)) if other_stream_ptr else None, int(bool(handle_waits))))

    # ------------------------------------------------------------------ reference surface, batched
    def reset(self, mask=None, copy=False):
        """reset() for all envs (or mask != 0).  Returns the Dict observation (host arrays)."""
        self._lidar_rows_fresh = False
        m = None
        if mask is not None:
            m = np.ascontiguousarray(mask, np.uint8)
            assert m.shape == (self.num_envs,)
        _cabi.check(_cabi.lib().ngw_reset(self._h, _cabi._ptr(m, np.uint8)))
        self._raise_flags()
        return self.get_observation(copy)

    def reset1(self):
        """reset() of a one-env handle for the gym.Env adapter: one C-ABI call (ngw_reset_host) that fills the host mirrors
        last_state() shows; a placement that cannot succeed raises like reset()."""
        self._lidar_rows_fresh = False
        args = self.__dict__.get('_reset1_args')
        if args is None:
            assert self.num_envs == 1
            o, p = self._obs, _cabi._ptr
            args = self._reset1_args = (self._h, None, p(o['map'], np.int8), p(o['agent_location'], np.int32), p(o['agent_facing_id'], np.int32),
                                        p(o['inventory_items_quantity'], np.int32), p(self._sel_host, np.uint8), p(self._steps_host, np.int32),
                                        p(self._flags_np, np.uint32))
        rc = _cabi.lib().ngw_reset_host(*args)
        if rc:
            _cabi.check(rc)
        if self._flags_np[0]:
            self._raise_flags()

    def step(self, actions, copy=False, with_obs=True):
        """step(action_id) for every env: (obs, reward[N] i32, done[N] bool, info) with host arrays.
        `with_obs=False` leaves the Dict observation on the device (obs is None): observation wrappers that return their
        own observation (LidarInFront, AgentMap) use it to skip the largest copy of the call.

        info = {'result' bool[N], 'step_cost' f64[N], 'step_cost_code', 'message_code', 'message_arg'};
        `messages(info, actions)` formats the reference's strings lazily."""
        a = np.ascontiguousarray(actions, np.int32)
        assert a.shape == (self.num_envs,)
        o = self._obs
        block = self.__dict__.get('_packed_block')
        if block is not None:
            # big batch, narrow wire format (ngw_step_host_packed): the actions are narrowed to bytes on their way into a buffer the step
            # kernel reads in place, map / inventory come back as deltas, pose + reward + done + info as 11 B per env in one copy
            fn = self.__dict__.get('_packed_call')
            if fn is None:                                            # (the library entry and the block's pointer: looked up / built once)
                fn = self.__dict__['_packed_call'] = (_cabi.lib().ngw_step_host_packed, _cabi._ptr(block, np.uint8))
            rc = fn[0](self._h, a.ctypes.data, fn[1], 1 if with_obs else 0)
            if rc:
                _cabi.check(rc)
            self._last_actions = a
            o._dirty = True
            self._steps_stale = True
            self._lidar_rows_fresh = self.lidar_fused                  # (ngw_lidar_host_rows: the rows of this state are in the host buffer)
            obs = None if not with_obs else ({k: v.copy() for k, v in o.items()} if copy else o)
            reward, done = (self._reward.copy(), self._done.view(np.bool_).copy()) if copy else (self._reward, self._done.view(np.bool_))
            info = StepInfo({'_words': self._info_words.copy() if copy else self._info_words})
            if self._flags_np[0]:
                self._raise_flags()
            if self.terminal_capture:
                dict.__setitem__(info, '_final_fn', self.terminal_observation)
                info['_final_observation'] = done
            return obs, reward, done, info
        self._lidar_rows_fresh = False
        self._act_pinned[...] = a
        cache = self.__dict__.setdefault('_step_args', {})
        args = cache.get(bool(with_obs))
        if args is None:                                             # the host arrays never move: the argument list is built once
            p = _cabi._ptr
            lazy = with_obs and self._info_words is not None         # big batch, one-block path: info decoded on demand from the words
            args = cache[bool(with_obs)] = (
                self._h, p(self._act_pinned, np.int32), p(o['map'], np.int8) if with_obs else None, p(o['agent_location'], np.int32),
                p(o['agent_facing_id'], np.int32), p(o['inventory_items_quantity'], np.int32), p(self._reward, np.int32),
                p(self._done, np.uint8), None if lazy else p(self._result, np.uint8), None if lazy else p(self._cost, np.uint8),
                None if lazy else p(self._msg, np.uint16), None if lazy else p(self._arg, np.uint16),
                p(self._flags_np, np.uint32), p(self._sel_host, np.uint8), p(self._steps_host, np.int32))
        rc = _cabi.lib().ngw_step_host(*args)                        # actions in, launch, observation + outputs out: one sync
        if rc:
            _cabi.check(rc)
        self._last_actions = a
        obs = None if not with_obs else ({k: v.copy() for k, v in o.items()} if copy else o)
        if with_obs and self._info_words is not None:
            reward, done = (self._reward.copy(), self._done.view(np.bool_).copy()) if copy else (self._reward, self._done.view(np.bool_))
            info = StepInfo({'_words': self._info_words.copy() if copy else self._info_words})
        else:
            reward, done, info = self._step_out_views(copy)
        if self._flags_np[0]:
            self._raise_flags()
        if self.terminal_capture:                             # gym.vector convention: info['final_observation'] (rows valid where the mask is set)
            dict.__setitem__(info, '_final_fn', self.terminal_observation)
            info['_final_observation'] = done
        return obs, reward, done, info

    def set_terminal_capture(self, on=True):
        """Keep, for every env that ends an episode in a step() under autoreset, the observation that episode ENDED in (the step
        itself returns the next episode's first observation): include/ngw.h ngw_set_terminal_capture.  step() then puts
        info['_final_observation'] (= done) and a lazily fetched info['final_observation'] into its info; terminal_observation() reads
        the side set directly.  Off by default.  Fused rollouts capture too: afterwards row e is the state env e's last finished episode ended in."""
        _cabi.check(_cabi.lib().ngw_set_terminal_capture(self._h, int(bool(on))))
        self.terminal_capture = bool(on)

    def terminal_observation(self, device=False):
        """Dict observation rows of the states episodes ended in ([N, ...] arrays like get_observation(); row e is meaningful for
        the envs whose `done` the last step() set, and keeps its value until env e ends an episode again).  device=True: zero-copy
        torch views of the side set."""
        N, S, K = self.num_envs, self.map_size, self.n_items
        if device:
            import torch
            p = [C.c_void_p() for _ in range(4)]
            _cabi.check(_cabi.lib().ngw_terminal_device_ptrs(self._h, *[C.byref(x) for x in p]))
            dev = 'cuda:%d' % self.device
            return {'map': torch.as_tensor(_DevArray(p[0].value, (N, S, S), '|i1'), device=dev),
                    'agent_location': torch.as_tensor(_DevArray(p[1].value, (N, 2), '<i4'), device=dev),
                    'agent_facing_id': torch.as_tensor(_DevArray(p[2].value, (N,), '<i4'), device=dev),
                    'inventory_items_quantity': torch.as_tensor(_DevArray(p[3].value, (N, K), '<i4'), device=dev)}
        out = {'map': np.zeros((N, S, S), np.int8), 'agent_location': np.zeros((N, 2), np.int32), 'agent_facing_id': np.zeros(N, np.int32),
               'inventory_items_quantity': np.zeros((N, K), np.int32)}
        _cabi.check(_cabi.lib().ngw_get_terminal_obs(self._h, _cabi._ptr(out['map'], np.int8), _cabi._ptr(out['agent_location'], np.int32),
                                                     _cabi._ptr(out['agent_facing_id'], np.int32), _cabi._ptr(out['inventory_items_quantity'], np.int32)))
        return out

    def _one_block_path(self):
        """Does ngw_step_host take its one-block path (pack + one copy, delta refresh) for this env's full step()?  The rule of
        ngw_abi.cpp: more than one wavefront of envs and more output bytes than the zero-copy limit (NGW_ZC_BYTES, 256 KiB)."""
        n, S2, K = self.num_envs, self.map_size ** 2, self.n_items
        total = sum((b + 255) & ~255 for b in (n * S2, n * 8, n * 4, n * K * 4, n * 4, n, 4, n, n * 4))
        return n > 64 and total > self._zc_bytes

    def refresh_host(self):
        """The next step() copies the whole observation to the host again instead of only what changed since the last step()
        (include/ngw.h ngw_host_step_layout) - call it after writing into the host arrays step() returned."""
        _cabi.check(_cabi.lib().ngw_host_mirror_invalidate(self._h))

    def step1(self, action):
        """step() of a one-env handle for the gym.Env adapter: same C-ABI call, but the argument list is built once and the
        outputs come back as Python scalars: (reward, done, result, cost code, message code, message arg)."""
        self._lidar_rows_fresh = False
        args = self.__dict__.get('_step1_args')
        if args is None:
            assert self.num_envs == 1
            o, p = self._obs, _cabi._ptr
            args = self._step1_args = (
                self._h, p(self._act_pinned, np.int32), p(o['map'], np.int8), p(o['agent_location'], np.int32),
                p(o['agent_facing_id'], np.int32), p(o['inventory_items_quantity'], np.int32), p(self._reward, np.int32),
                p(self._done, np.uint8), p(self._result, np.uint8), p(self._cost, np.uint8), p(self._msg, np.uint16), p(self._arg, np.uint16),
                p(self._flags_np, np.uint32), p(self._sel_host, np.uint8), p(self._steps_host, np.int32))
            self._step1_fn = _cabi.lib().ngw_step_host
            # memoryviews of the one-element host buffers: indexing them yields Python ints without a numpy scalar in between
            self._step1_mv = tuple(memoryview(x) for x in (self._act_pinned, self._reward, self._done, self._result, self._cost, self._msg,
                                                             self._arg, self._flags_np))
        act, reward, done, result, cost, msg, arg, flags = self._step1_mv
        act[0] = action
        rc = self._step1_fn(*args)
        if rc:
            _cabi.check(rc)
        if flags[0]:
            self._raise_flags()
        return (reward[0], done[0] != 0, result[0] != 0, cost[0], msg[0], arg[0])

    def last_state1(self):
        """The one env's state after the last step1() / reset1() as plain Python values - (map row bytes, r, c, facing,
        inventory row bytes, selected item id, step_count) - read from the host buffers that call filled."""
        mv = self.__dict__.get('_state1_mv')
        if mv is None:
            assert self.num_envs == 1
            o = self._obs
            mv = self._state1_mv = (memoryview(o['map']).cast('B'), memoryview(o['agent_location']).cast('B').cast('i'),
                                    memoryview(o['agent_facing_id']), memoryview(o['inventory_items_quantity']).cast('B'),
                                    memoryview(self._sel_host), memoryview(self._steps_host))
        m, loc, fac, inv, sel, steps = mv
        return m.tobytes(), loc[0], loc[1], fac[0], inv.tobytes(), sel[0], steps[0]

    def last_state(self):
        """State after the last step() as get_state() would return it, from the host buffers that call filled (no device
        traffic; `episode` is not part of it)."""
        if self.__dict__.get('_packed_block') is not None:
            self._obs._sync()
            if self._steps_stale:                             # (step_count does not travel in the narrow wire format: fetched when asked for)
                _cabi.check(_cabi.lib().ngw_get_state(self._h, 0, self.num_envs, None, None, None, None, None, _cabi._ptr(self._steps_host, np.int32), None))
                self._steps_stale = False
        st = self.__dict__.get('_last_state_views')
        if st is None:                                        # views of the host buffers every step fills: built once
            o = self._obs
            st = self._last_state_views = dict(map=o['map'].reshape(self.num_envs, -1), loc=o['agent_location'], facing=o['agent_facing_id'],
                                               inv=o['inventory_items_quantity'], selected=self._sel_host, step_count=self._steps_host)
        return st

    def get_observation(self, copy=False):
        o = self._obs
        if isinstance(o, LazyObs):
            o._dirty = False                                  # (the pose arrays are filled from the device below)
        _cabi.check(_cabi.lib().ngw_get_obs(self._h, _cabi._ptr(o['map'], np.int8), _cabi._ptr(o['agent_location'], np.int32),
                                            _cabi._ptr(o['agent_facing_id'], np.int32),
                                            _cabi._ptr(o['inventory_items_quantity'], np.int32)))
        return {k: v.copy() for k, v in o.items()} if copy else o

    def get_step_out(self, copy=False):
        _cabi.check(_cabi.lib().ngw_get_step_out(self._h, _cabi._ptr(self._reward, np.int32), _cabi._ptr(self._done, np.uint8),
                                                 _cabi._ptr(self._result, np.uint8), _cabi._ptr(self._cost, np.uint8),
                                                 _cabi._ptr(self._msg, np.uint16), _cabi._ptr(self._arg, np.uint16)))
        return self._step_out_views(copy)

    def _step_out_views(self, copy=False):
        info = StepInfo({'result': self._result.view(np.bool_), 'step_cost_code': self._cost, 'message_code': self._msg, 'message_arg': self._arg})
        reward, done = self._reward, self._done.view(np.bool_)      # (the kernels store 0 / 1)
        if copy:
            reward, done = reward.copy(), done.copy()
            info = StepInfo({k: v.copy() for k, v in dict.items(info)})
        return reward, done, info

    def messages(self, info, actions):
        return [self.spec.format_message(int(a), int(c), int(g))
                for a, c, g in zip(actions, info['message_code'], info['message_arg'])]

    def step_costs(self, info):
        """step_cost as the reference's Python objects (float or int)."""
        return [STEP_COSTS[int(c)] for c in info['step_cost_code']]

    # ------------------------------------------------------------------ device-resident path
    def step_device(self, actions_ptr):
        """One batched step with int32 actions already in HBM (`actions_ptr` = device address, e.g. tensor.data_ptr())."""
        self._lidar_rows_fresh = False
        _cabi.check(_cabi.lib().ngw_step_device(self._h, C.c_void_p(int(actions_ptr))))

    def step_device_many(self, actions_ptr, step_stride, n_steps):
        """n_steps batched steps from one call: step i reads int32 actions at device address actions_ptr + 4 * i * step_stride."""
        self._lidar_rows_fresh = False
        _cabi.check(_cabi.lib().ngw_step_device_many(self._h, C.c_void_p(int(actions_ptr)), int(step_stride), int(n_steps)))

    def _default_prefetch(self):
        """The refill cadence the library chose for this autoreset / horizon setting (include/ngw.h ngw_set_autoreset)."""
        v = C.c_int32()
        _cabi.check(_cabi.lib().ngw_get_reset_prefetch(self._h, C.byref(v)))
        return int(v.value)

    def set_reset_prefetch(self, every_n_steps):
        """Keep every env's NEXT episode prepared in shadow buffers and re-prepare consumed ones every `every_n_steps`
        batched steps (include/ngw.h ngw_set_reset_prefetch): resets become a row copy, which matters when episode ends
        are spread over the batch (a few envs per step).  Bit-identical results; 0 switches it off."""
        _cabi.check(_cabi.lib().ngw_set_reset_prefetch(self._h, int(every_n_steps)))
        self.reset_prefetch = int(every_n_steps)
        self._prefetch_user = True
        self._prefetch_arg = int(every_n_steps)

    def set_reset_prefetch_depth(self, depth):
        """How many episodes ahead are kept prepared per env: 1, 2, 4 or 8; 0 = automatic (the default: 1, growing by itself when
        envs end episodes faster than a refill comes round).  include/ngw.h ngw_set_reset_prefetch_depth; results do not depend on it."""
        _cabi.check(_cabi.lib().ngw_set_reset_prefetch_depth(self._h, int(depth)))
        self._depth_arg = int(depth)

    @property
    def refill_cadence(self):
        """Batched steps between two refills right now (0 = prepared episodes off): under the default setting the library adapts it
        to how fast episodes end (diagnostic entry point ngw_debug_refill_cadence)."""
        f = _cabi.lib().ngw_debug_refill_cadence
        f.argtypes, f.restype = [C.c_void_p], C.c_int
        return int(f(self._h))

    @property
    def step_reads_map_in_place(self):
        """True if step launches run the kernel that reads the few map cells a step needs straight from HBM, False if they stage the
        wave's maps through LDS (include/ngw.h ngw_step_kernel_info)."""
        v = C.c_int32()
        _cabi.check(_cabi.lib().ngw_step_kernel_info(self._h, C.byref(v)))
        return bool(v.value)

    @property
    def reset_prefetch_depth(self):
        v = C.c_int32()
        _cabi.check(_cabi.lib().ngw_get_reset_prefetch_depth(self._h, C.byref(v)))
        return int(v.value)

    def rollout(self, n_steps, action_seed=1234, t0=0):
        """Fused mode: n_steps steps in one launch with on-device uniform actions."""
        self._lidar_rows_fresh = False
        _cabi.check(_cabi.lib().ngw_rollout(self._h, int(n_steps), int(action_seed), int(t0)))

    def rollout_actions(self, actions_ptr, step_stride, n_steps):
        """Fused mode with the caller's actions: n_steps steps in one launch, step t reads int32 actions at device address
        actions_ptr + 4 * t * step_stride (e.g. a [T, N] int32 tensor: data_ptr(), N, T)."""
        self._lidar_rows_fresh = False
        _cabi.check(_cabi.lib().ngw_rollout_actions(self._h, C.c_void_p(int(actions_ptr)), int(step_stride), int(n_steps)))

    def rollout_outputs(self, reward_rows_ptr=0, done_rows_ptr=0, row_stride=0, accumulate=False):
        """Per-step outputs of the fused rollouts (include/ngw.h ngw_rollout_outputs): int32 reward rows and uint8 done rows
        [T, row_stride] in device memory (0 = off), and per-env episode accumulators kept across rollout calls."""
        _cabi.check(_cabi.lib().ngw_rollout_outputs(self._h, C.c_void_p(int(reward_rows_ptr)) if reward_rows_ptr else None,
                                                    C.c_void_p(int(done_rows_ptr)) if done_rows_ptr else None, int(row_stride), int(bool(accumulate))))
        self._rollout_out_args = (reward_rows_ptr, done_rows_ptr, row_stride, accumulate)   # (rebuild() re-binds them)

    def episode_stats(self, clear=False):
        """dict of int32 [N] host arrays: return / length of the running episodes, sum of returns / count of the finished ones."""
        out = {k: np.zeros(self.num_envs, np.int32) for k in ('run_return', 'run_length', 'sum_return', 'n_episodes')}
        _cabi.check(_cabi.lib().ngw_episode_stats(self._h, *[_cabi._ptr(out[k], np.int32) for k in ('run_return', 'run_length', 'sum_return', 'n_episodes')],
                                                  int(bool(clear))))
        return out

    def sync(self):
        _cabi.check(_cabi.lib().ngw_sync(self._h))

    def error_flags(self):
        f = C.c_uint32(0)
        _cabi.check(_cabi.lib().ngw_error_flags(self._h, C.byref(f)))
        return f.value

    def _raise_flags(self):
        f = self.error_flags()
        if f & F_PLACEMENT:
            raise AssertionError(PLACEMENT_MESSAGE)
        if f & F_INVALID_ACTION:
            raise ValueError("action id outside [0, %d) is not in list" % len(self.actions_id))

    def device_observation(self):
        """Current observation buffers as torch tensors (zero copy; valid until the next step/reset)."""
        import torch
        p = [C.c_void_p() for _ in range(4)]
        _cabi.check(_cabi.lib().ngw_obs_device_ptrs(self._h, *[C.byref(x) for x in p]))
        N, S, K = self.num_envs, self.map_size, self.n_items
        dev = 'cuda:%d' % self.device
        return {'map': torch.as_tensor(_DevArray(p[0].value, (N, S, S), '|i1'), device=dev),
                'agent_location': torch.as_tensor(_DevArray(p[1].value, (N, 2), '<i4'), device=dev),
                'agent_facing_id': torch.as_tensor(_DevArray(p[2].value, (N,), '<i4'), device=dev),
                'inventory_items_quantity': torch.as_tensor(_DevArray(p[3].value, (N, K), '<i4'), device=dev)}

    def device_outputs(self):
        import torch
        p = [C.c_void_p() for _ in range(3)]
        _cabi.check(_cabi.lib().ngw_out_device_ptrs(self._h, *[C.byref(x) for x in p]))
        N, dev = self.num_envs, 'cuda:%d' % self.device
        return {'reward': torch.as_tensor(_DevArray(p[0].value, (N,), '<i4'), device=dev),
                'done': torch.as_tensor(_DevArray(p[1].value, (N,), '|u1'), device=dev),
                'info': torch.as_tensor(_DevArray(p[2].value, (N,), '<i4'), device=dev)}

    # ------------------------------------------------------------------ multi-GPU observation stack (dist.py)
    def pack_layout(self):
        """Byte offsets of the seven payload sections + the payload size (include/ngw.h ngw_pack_layout)."""
        offs = (C.c_uint64 * 8)()
        _cabi.check(_cabi.lib().ngw_pack_layout(self._h, offs))
        return [int(x) for x in offs]

    def pack_obs(self, payload_ptr):
        """One launch: observation + step outputs -> the contiguous device payload at `payload_ptr` (16-byte aligned)."""
        _cabi.check(_cabi.lib().ngw_pack_obs(self._h, C.c_void_p(int(payload_ptr))))

    def unpack_obs(self, payloads_ptr, world, dst_ptrs):
        """Root side: `world` payloads back to back -> the global arrays at dst_ptrs (map, loc, facing, inv, reward, done, info)."""
        _cabi.check(_cabi.lib().ngw_unpack_obs(self._h, C.c_void_p(int(payloads_ptr)), int(world),
                                               *[C.c_void_p(int(x)) if x else None for x in dst_ptrs]))

    # ------------------------------------------------------------------ LidarInFront observation
    def lidar_configure(self, lidar_config=None, num_beams=8, fused=False, dtype=np.int16):
        """Enable the LidarInFront observation (reference observation_wrappers.py:10-80).  `lidar_config` fixes the lidar
        item set at wrap time like the reference wrapper does; by default it is built from the current spec.
        fused=True: every reset / step / rollout launch refreshes the observation in its own epilogue (no extra launch).
        dtype: the row format on the device and of what lidar_observation() returns - np.int16 (default: a beam entry is a
        range <= 64, inventory counts saturate at 32767), np.int32, or 'packed' (uint8 beam entries + int16 inventory tail,
        70 B per env for the reference's 8 beams; lidar_observation() then returns the pair (beams uint8 [N, B * NC],
        inventory int16 [N, NI]) as views of one buffer)."""
        from .lidar import LidarConfig
        self.lidar = lidar_config if lidar_config is not None else LidarConfig(self.spec, num_beams)
        self._lidar_c = self.lidar.compile(self.spec)
        L = _cabi.lib()
        _cabi.check(L.ngw_lidar_configure(self._h, C.byref(self._lidar_c)))
        self.lidar_len = self.lidar.obs_len(self.spec)
        self.lidar_packed = isinstance(dtype, str) and dtype == 'packed'
        self.lidar_dtype = np.dtype(np.uint8) if self.lidar_packed else np.dtype(dtype)
        assert self.lidar_packed or self.lidar_dtype in (np.dtype(np.int32), np.dtype(np.int16)), "lidar dtype must be int32, int16 or 'packed'"
        _cabi.check(L.ngw_lidar_set_output(self._h, 8 if self.lidar_packed else self.lidar_dtype.itemsize * 8))
        lay = [C.c_int32() for _ in range(4)]
        _cabi.check(L.ngw_lidar_row_layout(self._h, *[C.byref(x) for x in lay]))
        self.lidar_row_bytes, self._lidar_beam_bytes, self._lidar_inv_off, self._lidar_inv_bytes = [int(x.value) for x in lay]
        if self.lidar_packed:
            self._lidar_host = _cabi.pinned_array((self.num_envs, self.lidar_row_bytes), np.uint8)
        else:
            self._lidar_host = _cabi.pinned_array((self.num_envs, self.lidar_len), self.lidar_dtype)
        self.lidar_fused = bool(fused)
        _cabi.check(L.ngw_lidar_fuse(self._h, int(self.lidar_fused)))
        # fused: the packed host step of a big batch brings the rows of the state it produced across with its own slices (one call, one
        # synchronisation): lidar_observation() then hands out this buffer as long as nothing else has stepped the env since
        self._lidar_rows_fresh = False
        if self.lidar_fused and hasattr(L, 'ngw_lidar_host_rows'):
            _cabi.check(L.ngw_lidar_host_rows(self._h, _cabi._ptr(self._lidar_host, self.lidar_dtype)))

    def _lidar_split(self, rows):
        """(beams uint8 [N, B * NC], inventory int16 [N, NI]) views of packed rows (numpy array or torch tensor [N, row_bytes])."""
        nb = self.lidar.num_beams * len(self.lidar.lidar_items_id)
        beams, tail = rows[:, :nb], rows[:, self._lidar_inv_off:]
        if isinstance(rows, np.ndarray):
            return beams, tail.view(np.int16)
        import torch
        return beams, tail.view(torch.int16)

    def lidar_observation(self, device=False, copy=False):
        """[N, num_beams * n_lidar_items + n_inventory] observation of the current state in the configured dtype (one kernel
        launch, or none in fused mode - then it is the observation the last reset / step / rollout launch produced); with the
        packed format the pair (beams, inventory), see lidar_configure."""
        if not self.lidar_fused:
            _cabi.check(_cabi.lib().ngw_lidar(self._h))
        elif not device and self.__dict__.get('_lidar_rows_fresh'):   # the last call was a packed host step: it delivered the rows already
            out = self._lidar_host.copy() if copy else self._lidar_host
            return self._lidar_split(out) if self.lidar_packed else out
        if device:
            import torch
            p = C.c_void_p()
            _cabi.check(_cabi.lib().ngw_lidar_device_ptr(self._h, C.byref(p)))
            self.sync()
            if self.lidar_packed:
                raw = torch.as_tensor(_DevArray(p.value, (self.num_envs, self.lidar_row_bytes), '|u1'), device='cuda:%d' % self.device)
                return self._lidar_split(raw)
            return torch.as_tensor(_DevArray(p.value, (self.num_envs, self.lidar_len), self.lidar_dtype.str), device='cuda:%d' % self.device)
        _cabi.check(_cabi.lib().ngw_get_lidar(self._h, _cabi._ptr(self._lidar_host, self.lidar_dtype)))
        out = self._lidar_host.copy() if copy else self._lidar_host
        return self._lidar_split(out) if self.lidar_packed else out

    def lidar_widen(self, packed_pair, dtype=np.int32):
        """The [N, L] array of `dtype` from a packed (beams, inventory) pair (host side)."""
        beams, tail = packed_pair
        out = np.empty((beams.shape[0], beams.shape[1] + tail.shape[1]), dtype)
        out[:, :beams.shape[1]] = beams
        out[:, beams.shape[1]:] = tail
        return out

    # ------------------------------------------------------------------ state (checkpoint / oracle injection)
    def get_state(self, first=0, count=None):
        count = self.num_envs - first if count is None else count
        S2, K = self.map_size ** 2, self.n_items
        st = {'map': np.zeros((count, S2), np.int8), 'loc': np.zeros((count, 2), np.int32),
              'facing': np.zeros(count, np.int32), 'inv': np.zeros((count, K), np.int32),
              'selected': np.zeros(count, np.int32), 'step_count': np.zeros(count, np.int32),
              'episode': np.zeros(count, np.uint32)}
        _cabi.check(_cabi.lib().ngw_get_state(self._h, first, count, _cabi._ptr(st['map'], np.int8), _cabi._ptr(st['loc'], np.int32),
                                              _cabi._ptr(st['facing'], np.int32), _cabi._ptr(st['inv'], np.int32),
                                              _cabi._ptr(st['selected'], np.int32), _cabi._ptr(st['step_count'], np.int32),
                                              _cabi._ptr(st['episode'], np.uint32)))
        return st

    def set_state(self, first=0, map=None, loc=None, facing=None, inv=None, selected=None, step_count=None, episode=None):
        self._lidar_rows_fresh = False
        arrs = [(map, np.int8), (loc, np.int32), (facing, np.int32), (inv, np.int32), (selected, np.int32),
                (step_count, np.int32), (episode, np.uint32)]
        conv = [None if a is None else np.ascontiguousarray(a, dt) for a, dt in arrs]
        counts = {len(a) for a in conv if a is not None}
        assert len(counts) == 1, "all state arrays must cover the same number of envs"
        count = counts.pop()
        if conv[0] is not None:
            conv[0] = conv[0].reshape(count, -1)
        _cabi.check(_cabi.lib().ngw_set_state(self._h, first, count, *[_cabi._ptr(a, dt) for a, (_, dt) in zip(conv, arrs)]))

    # ------------------------------------------------------------------ timing (bench roofline leg) / hipGraph stepping
    def timing_begin(self):
        _cabi.check(_cabi.lib().ngw_timing_begin(self._h))

    def timing_end(self):
        """Device milliseconds between timing_begin() and now on the handle's stream (HIP event pair)."""
        ms = C.c_double(0)
        _cabi.check(_cabi.lib().ngw_timing_end(self._h, C.byref(ms)))
        return ms.value

    def graph_build(self, actions_ptr, step_stride, n_steps):
        """Capture n_steps step_device launches reading actions_ptr + i * step_stride (int32 elements) into one hipGraph."""
        _cabi.check(_cabi.lib().ngw_graph_build(self._h, C.c_void_p(int(actions_ptr)), int(step_stride), int(n_steps)))

    def graph_launch(self, reps=1):
        self._lidar_rows_fresh = False
        _cabi.check(_cabi.lib().ngw_graph_launch(self._h, int(reps)))

    def timing_mark(self):
        """Record the closing event of the timing pair without waiting for it (timing_end then only reads the pair)."""
        _cabi.check(_cabi.lib().ngw_timing_mark(self._h))

    def agent_view(self, view_size=5, device=False, copy=False):
        """AgentMap window (reference observation_wrappers.py:104-121): int8 [N, 2*view_size+1, 2*view_size+1], the map
        around each agent with 0 outside the map; one gather launch on the current state."""
        _cabi.check(_cabi.lib().ngw_agent_view(self._h, int(view_size)))
        W = 2 * int(view_size) + 1
        if device:
            import torch
            p = C.c_void_p()
            _cabi.check(_cabi.lib().ngw_agent_view_device_ptr(self._h, C.byref(p)))
            return torch.as_tensor(_DevArray(p.value, (self.num_envs, W, W), '|i1'), device='cuda:%d' % self.device)
        host = getattr(self, '_view_host', None)
        if host is None or host.shape[1] != W:
            host = self._view_host = np.zeros((self.num_envs, W, W), np.int8)
        _cabi.check(_cabi.lib().ngw_get_agent_view(self._h, _cabi._ptr(host, np.int8)))
        return host.copy() if copy else host
