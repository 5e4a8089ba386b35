"""ctypes binding of the C-ABI in include/ngw.h (libngw_hip.so = HIP kernels + host glue).

This is the only way into the hot path: if the library is missing, cannot be loaded, or no GPU is visible,
the failure is loud - there is no CPU fallback anywhere in this package."""
import ctypes as C
import os

import numpy as np

from .spec import NgwSpec

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('NGW_LIB') or os.path.join(_HERE, 'libngw_hip.so')   # NGW_LIB: A/B builds while tuning

E_INVALID_ARG, E_HIP, E_INVALID_ACTION, E_PLACEMENT, E_NO_DEVICE = -1, -2, -3, -4, -5

# every symbol include/ngw.h declares (checked by tests/test_cabi_symbols.py against the header text)
SYMBOLS = ['ngw_abi_version', 'ngw_spec_size', 'ngw_last_error', 'ngw_device_count', 'ngw_create', 'ngw_destroy',
           'ngw_set_autoreset', 'ngw_set_stream', 'ngw_reset', 'ngw_step', 'ngw_step_device', 'ngw_rollout',
           'ngw_get_obs', 'ngw_get_step_out', 'ngw_get_state', 'ngw_set_state', 'ngw_obs_device_ptrs',
           'ngw_out_device_ptrs', 'ngw_sync', 'ngw_error_flags', 'ngw_timing_begin', 'ngw_timing_end',
           'ngw_graph_build', 'ngw_graph_launch', 'ngw_get_reset_prefetch', 'ngw_timing_mark', 'ngw_lidar_configure', 'ngw_lidar', 'ngw_lidar_fuse',
           'ngw_get_lidar', 'ngw_lidar_device_ptr', 'ngw_host_alloc', 'ngw_host_free', 'ngw_agent_view',
           'ngw_get_agent_view', 'ngw_agent_view_device_ptr', 'ngw_set_reset_prefetch', 'ngw_step_host', 'ngw_lidar_set_output', 'ngw_rollout_actions',
           'ngw_pack_layout', 'ngw_pack_obs', 'ngw_unpack_obs', 'ngw_rollout_outputs', 'ngw_episode_stats', 'ngw_host_step_layout', 'ngw_step_device_many',
           'ngw_set_reset_prefetch_depth', 'ngw_get_reset_prefetch_depth', 'ngw_stream_order', 'ngw_host_mirror_invalidate', 'ngw_reset_host',
           'ngw_lidar_row_layout', 'ngw_step_kernel_info', 'ngw_set_terminal_capture', 'ngw_get_terminal_obs', 'ngw_terminal_device_ptrs',
           'ngw_host_step_layout_packed', 'ngw_step_host_packed', 'ngw_lidar_host_rows']

_lib = None


class NgwError(RuntimeError):
    pass


_PTR_CACHE = {}      # id(array) -> (weakref, address): `arr.ctypes` costs ~40 us on arrays over foreign (pinned) buffers


def _ptr(arr, dtype):
    """Host pointer of a C-contiguous ndarray of `dtype`, or NULL for None."""
    if arr is None:
        return None
    ent = _PTR_CACHE.get(id(arr))
    if ent is not None and ent[0]() is arr:
        return ent[1]
    assert isinstance(arr, np.ndarray) and arr.dtype == dtype and arr.flags['C_CONTIGUOUS'], \
        "expected a C-contiguous %s array" % np.dtype(dtype).name
    addr = C.c_void_p(arr.__array_interface__['data'][0])
    if arr.base is not None and len(_PTR_CACHE) < 4096:        # long-lived views (the env's own buffers): remember
        import weakref
        try:
            _PTR_CACHE[id(arr)] = (weakref.ref(arr, lambda _r, k=id(arr): _PTR_CACHE.pop(k, None)), addr)
        except TypeError:
            pass
    return addr


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NgwError("HIP extension %s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                       "or `make -C gym_novel_gridworlds_amd/csrc` (there is no CPU fallback)" % LIB_PATH)
    # One HIP runtime per process: PyTorch-ROCm wheels bundle their own libamdhip64.so.7 / libhsa-runtime64, and a
    # second copy (the system one this library would otherwise pull in) leaves whichever initialises later without
    # GPUs.  Importing torch first makes the dynamic linker bind libngw_hip.so to torch's already-loaded runtime.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, i64, u64, i32 = C.c_void_p, C.c_int64, C.c_uint64, C.c_int32
    L.ngw_abi_version.restype = C.c_int
    L.ngw_spec_size.restype = C.c_int
    L.ngw_last_error.restype = C.c_char_p
    L.ngw_device_count.restype = C.c_int
    L.ngw_create.argtypes = [C.POINTER(NgwSpec), i64, C.c_int, u64, i64, C.POINTER(vp)]
    L.ngw_destroy.argtypes = [vp]
    L.ngw_set_autoreset.argtypes = [vp, C.c_int, C.c_int]
    L.ngw_set_stream.argtypes = [vp, vp]
    if hasattr(L, 'ngw_host_mirror_invalidate'):
        L.ngw_host_mirror_invalidate.argtypes = [vp]
    if hasattr(L, 'ngw_stream_order'):
        L.ngw_stream_order.argtypes = [vp, vp, C.c_int]
    if hasattr(L, 'ngw_set_reset_prefetch'):           # absent only in older builds loaded through NGW_LIB (A/B runs)
        L.ngw_set_reset_prefetch.argtypes = [vp, i32]
    L.ngw_reset.argtypes = [vp, vp]
    if hasattr(L, 'ngw_reset_host'):
        L.ngw_reset_host.argtypes = [vp] + [vp] * 8
    L.ngw_step.argtypes = [vp, vp]
    L.ngw_step_device.argtypes = [vp, vp]
    if hasattr(L, 'ngw_step_device_many'):
        L.ngw_step_device_many.argtypes = [vp, vp, i64, i32]
    if hasattr(L, 'ngw_step_host'):
        L.ngw_step_host.argtypes = [vp] + [vp] * 14
    L.ngw_rollout.argtypes = [vp, i32, u64, i64]
    if hasattr(L, 'ngw_rollout_actions'):
        L.ngw_rollout_actions.argtypes = [vp, vp, i64, i32]
    L.ngw_get_obs.argtypes = [vp, vp, vp, vp, vp]
    L.ngw_get_step_out.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    L.ngw_get_state.argtypes = [vp, i64, i64, vp, vp, vp, vp, vp, vp, vp]
    L.ngw_set_state.argtypes = [vp, i64, i64, vp, vp, vp, vp, vp, vp, vp]
    L.ngw_obs_device_ptrs.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    L.ngw_out_device_ptrs.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    L.ngw_sync.argtypes = [vp]
    L.ngw_error_flags.argtypes = [vp, C.POINTER(C.c_uint32)]
    L.ngw_timing_begin.argtypes = [vp]
    L.ngw_timing_end.argtypes = [vp, C.POINTER(C.c_double)]
    L.ngw_timing_mark.argtypes = [vp]
    L.ngw_graph_build.argtypes = [vp, vp, i64, i32]
    L.ngw_graph_launch.argtypes = [vp, i32]
    L.ngw_get_reset_prefetch.argtypes = [vp, C.POINTER(C.c_int32)]
    if hasattr(L, 'ngw_set_reset_prefetch_depth'):
        L.ngw_set_reset_prefetch_depth.argtypes = [vp, i32]
        L.ngw_get_reset_prefetch_depth.argtypes = [vp, C.POINTER(C.c_int32)]
    L.ngw_lidar_configure.argtypes = [vp, vp]
    L.ngw_lidar.argtypes = [vp]
    L.ngw_lidar_fuse.argtypes = [vp, C.c_int]
    if hasattr(L, 'ngw_lidar_set_output'):
        L.ngw_lidar_set_output.argtypes = [vp, C.c_int]
    if hasattr(L, 'ngw_step_host_packed'):
        L.ngw_host_step_layout_packed.argtypes = [vp, C.POINTER(u64)]
        L.ngw_step_host_packed.argtypes = [vp, vp, vp, C.c_int]
    if hasattr(L, 'ngw_set_terminal_capture'):
        L.ngw_set_terminal_capture.argtypes = [vp, C.c_int]
        L.ngw_get_terminal_obs.argtypes = [vp, vp, vp, vp, vp]
        L.ngw_terminal_device_ptrs.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    if hasattr(L, 'ngw_step_kernel_info'):
        L.ngw_step_kernel_info.argtypes = [vp, C.POINTER(C.c_int32)]
    if hasattr(L, 'ngw_lidar_row_layout'):
        L.ngw_lidar_row_layout.argtypes = [vp] + [C.POINTER(C.c_int32)] * 4
    L.ngw_get_lidar.argtypes = [vp, vp]
    if hasattr(L, 'ngw_lidar_host_rows'):
        L.ngw_lidar_host_rows.argtypes = [vp, vp]
    L.ngw_lidar_device_ptr.argtypes = [vp, C.POINTER(vp)]
    L.ngw_agent_view.argtypes = [vp, C.c_int]
    L.ngw_get_agent_view.argtypes = [vp, vp]
    L.ngw_agent_view_device_ptr.argtypes = [vp, C.POINTER(vp)]
    if hasattr(L, 'ngw_pack_obs'):
        L.ngw_pack_layout.argtypes = [vp, C.POINTER(u64)]
        L.ngw_pack_obs.argtypes = [vp, vp]
        L.ngw_unpack_obs.argtypes = [vp, vp, i32] + [vp] * 7
    if hasattr(L, 'ngw_rollout_outputs'):
        L.ngw_rollout_outputs.argtypes = [vp, vp, vp, i64, C.c_int]
        L.ngw_episode_stats.argtypes = [vp, vp, vp, vp, vp, C.c_int]
    if hasattr(L, 'ngw_host_step_layout'):
        L.ngw_host_step_layout.argtypes = [vp, C.POINTER(u64)]
    L.ngw_host_alloc.argtypes = [u64]
    L.ngw_host_alloc.restype = vp
    L.ngw_host_free.argtypes = [vp]
    if L.ngw_spec_size() != C.sizeof(NgwSpec):
        raise NgwError("ngw_spec layout mismatch: library %d bytes, binding %d bytes" % (L.ngw_spec_size(), C.sizeof(NgwSpec)))
    _lib = L
    return L


def pinned_array(shape, dtype):
    """numpy array over page-locked host memory; falls back to np.zeros if pinning fails.  The memory is freed when the LAST
    array that uses it goes away: numpy collapses the `.base` of every view (slices, reshapes) to the one frombuffer array
    created here, so the finalizer hangs on that array, not on the reshaped view handed out."""
    import weakref
    dt = np.dtype(dtype)
    n = int(np.prod(shape)) * dt.itemsize
    p = lib().ngw_host_alloc(n)
    if not p:
        return np.zeros(shape, dt)
    buf = (C.c_char * max(n, 1)).from_address(p)
    flat = np.frombuffer(buf, dtype=dt, count=int(np.prod(shape)))      # the ultimate owner every view's .base points at
    weakref.finalize(flat, _free_pinned, int(p))
    return flat.reshape(shape)


def _free_pinned(addr):
    if _lib is not None:
        _lib.ngw_host_free(C.c_void_p(addr))


def last_error():
    return lib().ngw_last_error().decode('utf-8', 'replace')


def check(rc):
    """Map C-ABI return codes to the exception types the reference raises (SURVEY.md §8(b) Errors)."""
    if rc == 0:
        return
    msg = last_error()
    if rc == E_INVALID_ACTION:
        raise ValueError(msg)                                   # pogostick_v1_env.py:236
    if rc == E_PLACEMENT:
        raise AssertionError(msg)                               # pogostick_v1_env.py:167
    if rc == E_INVALID_ARG:
        raise ValueError(msg)
    raise NgwError("ngw C-ABI call failed (%d): %s" % (rc, msg))
