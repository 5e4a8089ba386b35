"""ctypes loader for the CPU oracle (oracle/ngw_oracle.c) - TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; nothing under
gym_novel_gridworlds_amd/ does.  `build()` compiles the C restatement with gcc (oracle/Makefile)."""
import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_DIR, '_build', 'libngw_oracle.so')
_lib = None

i8p, i32p, u8p, u32p = (np.ctypeslib.ndpointer(dtype=t, flags='C_CONTIGUOUS')
                        for t in (np.int8, np.int32, np.uint8, np.uint32))


def build(force=False):
    src = os.path.join(_DIR, 'ngw_oracle.c')
    hdr = os.path.join(_DIR, '..', 'include', 'ngw.h')
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(['make', '-s', '-C', _DIR], stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        vp = C.c_void_p
        L.ngwo_mt_seed.argtypes = [vp, C.c_uint32]
        L.ngwo_mt_next.argtypes = [vp]
        L.ngwo_mt_next.restype = C.c_uint32
        L.ngwo_mt_bounded.argtypes = [vp, C.c_uint32]
        L.ngwo_mt_bounded.restype = C.c_uint32
        L.ngwo_philox4x32_10.argtypes = [u32p, u32p, u32p]
        L.ngwo_rollout_action.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32]
        L.ngwo_rollout_action.restype = C.c_uint32
        st = [i8p, i32p, i32p, i32p, i32p, i32p]          # map, loc, facing, inv, selected, step_count
        L.ngwo_reset_mt.argtypes = [vp, vp] + st
        L.ngwo_reset_philox.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_uint32] + st
        L.ngwo_step.argtypes = [vp] + st + [C.c_int32, i32p, u8p, u32p]
        L.ngwo_step_batch.argtypes = [vp, C.c_int64] + st + [u32p, i32p, i32p, u8p, u32p, C.c_int, C.c_int,
                                                              C.c_uint64, C.c_int64]
        L.ngwo_step_batch.restype = C.c_uint32
        L.ngwo_reset_batch.argtypes = [vp, C.c_int64, vp] + st + [u32p, C.c_uint64, C.c_int64]
        L.ngwo_reset_batch.restype = C.c_uint32
        L.ngwo_rollout_batch.argtypes = [vp, C.c_int64, C.c_int32, C.c_int64, C.c_uint64] + st + \
            [u32p, i32p, u8p, u32p, C.c_int, C.c_int, C.c_uint64, C.c_int64]
        L.ngwo_rollout_batch.restype = C.c_uint32
        L.ngwo_lidar.argtypes = [vp, C.c_int, C.c_int, C.c_int64, i8p, i32p, i32p, i32p, i32p]
        L.ngwo_set_threads.argtypes = [C.c_int]
        L.ngwo_set_threads.restype = C.c_int
        _lib = L
    return _lib


def set_threads(n):
    """OpenMP threads used by the batched drivers; returns the count in effect."""
    return lib().ngwo_set_threads(int(n))


def lidar(ccfg, S, K, map_, loc, facing, inv):
    """LidarInFront observation of n states (compiled NgwLidarCfg) -> int32 [n, L]."""
    assert lib().ngwo_lidar_cfg_size() == C.sizeof(ccfg), "ngw_lidar_cfg layout mismatch"
    n = len(facing)
    out = np.zeros((n, ccfg.num_beams * ccfg.n_chan + ccfg.n_inv), np.int32)
    lib().ngwo_lidar(C.byref(ccfg), S, K, n, np.ascontiguousarray(map_, np.int8).reshape(n, -1), np.ascontiguousarray(loc, np.int32),
                     np.ascontiguousarray(facing, np.int32), np.ascontiguousarray(inv, np.int32), out)
    return out


def agent_view(map_, loc, view_size=5):
    """AgentMap.get_agentView for n states (reference observation_wrappers.py:104-121): pad the map with `view_size`
    zeros on every side and slice the (2 * view_size + 1)^2 window whose top-left corner, in padded coordinates, is the
    agent location.  int8 [n, W, W]."""
    m = np.asarray(map_, np.int8)
    n = m.shape[0]
    S = m.shape[-1] if m.ndim == 3 else int(round((m.size // max(n, 1)) ** 0.5))
    m = m.reshape(n, S, S)
    V, W = int(view_size), 2 * int(view_size) + 1
    ext = np.zeros((n, S + 2 * V, S + 2 * V), np.int8)
    ext[:, V:V + S, V:V + S] = m
    out = np.empty((n, W, W), np.int8)
    for e in range(n):
        r, c = int(loc[e][0]), int(loc[e][1])
        out[e] = ext[e, r:r + W, c:c + W]
    return out


class MT19937:
    """numpy-legacy global stream: np.random.seed(int) + raw words / bounded draws."""

    def __init__(self, seed):
        self.buf = C.create_string_buffer(lib().ngwo_mt_size())
        lib().ngwo_mt_seed(self.buf, seed)

    def next(self):
        return lib().ngwo_mt_next(self.buf)

    def bounded(self, mx):
        return lib().ngwo_mt_bounded(self.buf, mx)


class State:
    """SoA state of n envs in the layout of ngw_get_state / ngw_set_state."""

    def __init__(self, n, S, K):
        self.n, self.S, self.K = n, S, K
        self.map = np.zeros((n, S * S), np.int8)
        self.loc = np.zeros((n, 2), np.int32)
        self.facing = np.zeros(n, np.int32)
        self.inv = np.zeros((n, K), np.int32)
        self.selected = np.zeros(n, np.int32)
        self.step_count = np.zeros(n, np.int32)
        self.episode = np.zeros(n, np.uint32)

    def arrays(self):
        return [self.map, self.loc, self.facing, self.inv, self.selected, self.step_count]

    def copy(self):
        s = State(self.n, self.S, self.K)
        for k in ('map', 'loc', 'facing', 'inv', 'selected', 'step_count', 'episode'):
            getattr(s, k)[...] = getattr(self, k)
        return s


class Oracle:
    """Batched driver around the C restatement; `spec` is a compiled NgwSpec (ctypes struct)."""

    def __init__(self, cspec, n=1, seed=0, env_index_base=0, autoreset=False, horizon=0):
        assert lib().ngwo_spec_size() == C.sizeof(cspec), "ngw_spec layout mismatch"
        self.cspec, self.sp = cspec, C.byref(cspec)
        self.n, self.seed, self.base = n, seed, env_index_base
        self.autoreset, self.horizon = int(autoreset), int(horizon)
        self.st = State(n, cspec.map_size, cspec.n_items)
        self.reward = np.zeros(n, np.int32)
        self.done = np.zeros(n, np.uint8)
        self.info = np.zeros(n, np.uint32)

    def reset_mt(self, mt, i=0):
        st = self.st
        return lib().ngwo_reset_mt(self.sp, mt.buf, st.map[i], st.loc[i], st.facing[i:i + 1], st.inv[i],
                                   st.selected[i:i + 1], st.step_count[i:i + 1])

    def reset(self, mask=None):
        st = self.st
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8).ctypes.data
        return lib().ngwo_reset_batch(self.sp, self.n, m, *st.arrays(), st.episode, self.seed, self.base)

    def step(self, actions):
        st = self.st
        a = np.ascontiguousarray(actions, np.int32)
        return lib().ngwo_step_batch(self.sp, self.n, *st.arrays(), st.episode, a, self.reward, self.done, self.info,
                                     self.autoreset, self.horizon, self.seed, self.base)

    def rollout(self, n_steps, action_seed, t0=0):
        st = self.st
        return lib().ngwo_rollout_batch(self.sp, self.n, n_steps, t0, action_seed, *st.arrays(), st.episode,
                                        self.reward, self.done, self.info, self.autoreset, self.horizon, self.seed,
                                        self.base)

    # decoded info word (include/ngw.h NGW_INFO_*)
    @property
    def result(self):
        return (self.info & 1).astype(np.uint8)

    @property
    def cost_code(self):
        return ((self.info >> 2) & 63).astype(np.uint8)

    @property
    def msg_code(self):
        return ((self.info >> 8) & 255).astype(np.uint16)

    @property
    def msg_arg(self):
        return (self.info >> 16).astype(np.uint16)
