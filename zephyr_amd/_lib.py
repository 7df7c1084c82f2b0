"""ctypes binding of libhelm.so (C ABI declared in include/helm.h).

The library is built in-tree by `__graft_entry__.build()` (hipcc --offload-arch=gfx950).
There is NO CPU fallback: importing the operators without the built library, or using
them without a GPU, raises.
"""
import ctypes
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libhelm.so')

HELM_MINIZEPHYR, HELM_EURUS, HELM_3D = 0, 1, 2
HELM_BICGSTAB, HELM_CGNR, HELM_AUTO, HELM_MG, HELM_DIRECT = 0, 1, 2, 3, 4
HELM_RHS_NODE_MAJOR, HELM_OUT_NODE_MAJOR, HELM_NODE_MAJOR = 1, 2, 3
METHODS = {'bicgstab': HELM_BICGSTAB, 'cgnr': HELM_CGNR, 'auto': HELM_AUTO, 'mg': HELM_MG, 'direct': HELM_DIRECT}

ERRORS = {-1: 'HELM_ERR_ARG', -2: 'HELM_ERR_DEVICE', -3: 'HELM_ERR_STATE', -4: 'HELM_ERR_UNSUPPORTED', -5: 'HELM_ERR_PML'}


class HelmError(RuntimeError):
    def __init__(self, code, text):
        self.code = code
        RuntimeError.__init__(self, '%s: %s' % (ERRORS.get(code, code), text))


class SolveOpts(ctypes.Structure):
    _fields_ = [('method', ctypes.c_int), ('rtol', ctypes.c_double), ('maxit', ctypes.c_int),
                ('check_every', ctypes.c_int), ('batch', ctypes.c_int), ('flags', ctypes.c_int)]


class SolveInfo(ctypes.Structure):
    _fields_ = [('iterations', ctypes.c_int), ('status', ctypes.c_int), ('restarts', ctypes.c_int),
                ('method', ctypes.c_int), ('relres', ctypes.c_double)]


class Timing(ctypes.Structure):
    _fields_ = [('solve_ms', ctypes.c_double), ('apply_ms', ctypes.c_double),
                ('apply_launches', ctypes.c_longlong), ('apply_bytes', ctypes.c_double),
                ('factor_ms', ctypes.c_double), ('gemm_ms', ctypes.c_double),
                ('gemm_launches', ctypes.c_longlong), ('gemm_flops', ctypes.c_double),
                ('gemm_big_ms', ctypes.c_double), ('gemm_big_launches', ctypes.c_longlong), ('gemm_big_flops', ctypes.c_double),
                ('gemm_bytes', ctypes.c_double), ('gemm_sol_ms', ctypes.c_double)]


class Tuning(ctypes.Structure):
    """helm_tuning of include/helm.h: the library's real options (each also an environment variable, named there)."""
    _fields_ = [('nd_leaf', ctypes.c_int), ('nd_ws_gb', ctypes.c_double), ('nd_sparse_rhs', ctypes.c_int), ('nd_stable', ctypes.c_int),
                ('nd_stable_thr', ctypes.c_double), ('nd_stable_safety', ctypes.c_double), ('nd_fused_leaf', ctypes.c_int),
                ('nd_fused_leaf_min', ctypes.c_int), ('nd_gjstep', ctypes.c_int), ('nd_gjstep_min', ctypes.c_int), ('nd_overlap', ctypes.c_int),
                ('nd_xcd_map', ctypes.c_int), ('nd_plans', ctypes.c_int), ('nd_direct_out', ctypes.c_int), ('nd_many', ctypes.c_int), ('nd_leaf_idle', ctypes.c_int), ('auto_direct', ctypes.c_int),
                ('auto_mg3', ctypes.c_int), ('prof_ext', ctypes.c_int), ('ws_slots', ctypes.c_int), ('pf_prio', ctypes.c_int),
                ('mg3_keep', ctypes.c_int), ('mg3_keep_levels', ctypes.c_int), ('mg3_galerkin', ctypes.c_int), ('mg3_depth_model', ctypes.c_int),
                ('mg3_bt_f32', ctypes.c_int), ('mg3_otf', ctypes.c_int), ('mg3_f32', ctypes.c_int), ('mg3_omega', ctypes.c_double), ('sync_spin_ms', ctypes.c_double), ('sync_sleep_us', ctypes.c_int)]


class RuntimeStats(ctypes.Structure):
    """helm_runtime_stats of include/helm.h: what the library made the HIP runtime create since the last reset."""
    _fields_ = [('dev_allocs', ctypes.c_longlong), ('dev_alloc_bytes', ctypes.c_double), ('dev_alloc_ms', ctypes.c_double),
                ('host_allocs', ctypes.c_longlong), ('host_alloc_bytes', ctypes.c_double), ('host_alloc_ms', ctypes.c_double),
                ('events_created', ctypes.c_longlong), ('streams_created', ctypes.c_longlong),
                ('first_launches', ctypes.c_longlong), ('first_launch_ms', ctypes.c_double),
                ('kernels_registered', ctypes.c_longlong), ('kernels_resolved', ctypes.c_longlong), ('warm_ms', ctypes.c_double),
                ('dev_frees', ctypes.c_longlong), ('dev_free_ms', ctypes.c_double),
                ('sync_calls', ctypes.c_longlong), ('sync_ms', ctypes.c_double), ('slow_syncs', ctypes.c_longlong), ('worst_sync_ms', ctypes.c_double)]


# every symbol include/helm.h declares, with its ctypes signature
_c_dp = ctypes.POINTER(ctypes.c_double)
_SIGNATURES = {
    'helm_device_count': (ctypes.c_int, []),
    'helm_version': (ctypes.c_char_p, []),
    'helm_create': (ctypes.c_void_p, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double,
                                      ctypes.c_double, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]),
    'helm_create3d': (ctypes.c_void_p, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double,
                                        ctypes.c_double, ctypes.c_double, ctypes.c_int]),
    'helm_destroy': (None, [ctypes.c_void_p]),
    'helm_last_error': (ctypes.c_char_p, [ctypes.c_void_p]),
    'helm_set_stream': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    'helm_set_model': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                      ctypes.c_void_p, ctypes.c_void_p]),
    'helm_assemble': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                     ctypes.c_double, ctypes.c_double]),
    'helm_num_blocks': (ctypes.c_int, [ctypes.c_void_p]),
    'helm_num_points': (ctypes.c_longlong, [ctypes.c_void_p]),
    'helm_get_diagonals': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    'helm_apply': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]),
    'helm_apply_device': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]),
    'helm_solve': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong,
                                  ctypes.c_double, ctypes.c_double, ctypes.POINTER(SolveOpts), ctypes.POINTER(SolveInfo)]),
    'helm_solve_device': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong,
                                         ctypes.c_double, ctypes.c_double, ctypes.POINTER(SolveOpts), ctypes.POINTER(SolveInfo)]),
    'helm_solve_coo': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_int,
                                      ctypes.c_longlong, ctypes.c_double, ctypes.c_double, ctypes.POINTER(SolveOpts), ctypes.POINTER(SolveInfo)]),
    'helm_host_alloc': (ctypes.c_void_p, [ctypes.c_size_t]),
    'helm_host_free': (None, [ctypes.c_void_p, ctypes.c_size_t]),
    'helm_rhs_from_coo_device_layout': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong,
                                                       ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong, ctypes.c_int]),
    'helm_prefactor': (ctypes.c_int, [ctypes.c_void_p]),
    'helm_prefactor_n': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    'helm_prefactor_many': (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int]),
    'helm_reserve': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong, ctypes.c_int]),
    'helm_set_tolerance_hint': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_double]),
    'helm_last_timing': (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(Timing)]),
    'helm_set_profiling': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    'helm_imaging_accumulate_device': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                                      ctypes.c_void_p, ctypes.c_void_p]),
    'helm_get_tuning': (ctypes.c_int, [ctypes.POINTER(Tuning)]),
    'helm_set_tuning': (ctypes.c_int, [ctypes.POINTER(Tuning)]),
    'helm_pool_spares': (ctypes.c_int, [ctypes.c_int, ctypes.c_int]),
    'helm_trim': (ctypes.c_int, []),
    'helm_host_trim': (ctypes.c_int, []),
    'helm_debug_ws_slots': (ctypes.c_int, [ctypes.c_int, ctypes.c_longlong]),
    'helm_debug_plan_cache': (ctypes.c_int, [ctypes.c_int]),
    'helm_debug_ws_selftest': (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_longlong]),
    'helm_debug_alloc_stats': (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(ctypes.c_longlong), ctypes.POINTER(ctypes.c_double)]),
    'helm_warm': (ctypes.c_int, [ctypes.c_int]),
    'helm_debug_runtime_stats': (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(RuntimeStats)]),
    'helm_debug_stall_watch': (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_longlong)]),
    'helm_rhs_from_coo_device': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong,
                                                ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong]),
    'helm_sample_device': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_void_p,
                                          ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]),
    'helm_direct_plan': (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]),
    'helm_direct_plan_front': (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]),
    'helm_mg3_axis': (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_int,
                                     ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                     ctypes.c_void_p]),
    'helm_debug_zgemm': (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]),
    'helm_debug_inverse': (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]),
    'helm_debug_zgemm_bench': (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                              ctypes.POINTER(ctypes.c_double)]),
    'helm_set_rhs_support': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int]),
    'helm_rhs_support_from_coo': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int]),
    'helm_debug_inverse_bench': (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]),
}

_lib = None


def exported_symbols():
    return sorted(_SIGNATURES)


def load():
    """Load libhelm.so (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        # PyTorch-ROCm wheels bundle their own libamdhip64.so.7; whichever copy of that soname is mapped first serves
        # the whole process.  Load torch's first (when it is installed) so that torch tensors / RCCL and libhelm share
        # ONE HIP runtime -- the other order leaves torch without a visible GPU.
        if os.environ.get('HELM_NO_TORCH_PRELOAD', '0') != '1':
            try:
                import torch  # noqa: F401
            except Exception:
                pass
        if not os.path.exists(LIB_PATH):
            raise ImportError('libhelm.so not found at %s: run `python -c "import __graft_entry__ as g; g.build()"` '
                              '(hipcc --offload-arch=gfx950).  There is no CPU fallback.' % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def tuning():
    'the options in force (helm_get_tuning): a Tuning structure; change fields and hand it to set_tuning()'
    t = Tuning()
    check(load().helm_get_tuning(ctypes.byref(t)))
    return t


def set_tuning(t=None):
    'replace the options for the process (None: defaults + environment again)'
    check(load().helm_set_tuning(ctypes.byref(t) if t is not None else None))


def runtime_stats(reset=False):
    'helm_debug_runtime_stats as a dict (device / pinned allocations, events, streams, first launches since the last reset)'
    st = RuntimeStats()
    check(load().helm_debug_runtime_stats(1 if reset else 0, ctypes.byref(st)))
    return {k: getattr(st, k) for k, _ in RuntimeStats._fields_}


def last_error(handle=None):
    msg = load().helm_last_error(handle)
    return msg.decode() if msg else ''


def check(rc, handle=None):
    if rc < 0:
        raise HelmError(rc, last_error(handle))
    return rc


def require_gpu():
    n = load().helm_device_count()
    if n <= 0:
        raise HelmError(-2, 'no HIP device visible (%s); libhelm has no CPU path' % last_error(None))
    return n


class _PinnedBlock(object):
    'owner of one helm_host_alloc buffer; numpy arrays made from it keep it alive (it is their .base)'

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        self.ptr = load().helm_host_alloc(self.nbytes)
        if not self.ptr:
            raise MemoryError('helm_host_alloc(%d) failed' % self.nbytes)
        self.__array_interface__ = {'shape': (self.nbytes,), 'typestr': '|u1', 'data': (self.ptr, False), 'version': 3}

    def __del__(self):
        try:
            if self.ptr:
                load().helm_host_free(self.ptr, self.nbytes)
                self.ptr = None
        except Exception:
            pass


def pinned_empty(shape, dtype=np.complex128):
    """np.empty in pinned host memory (device-to-host copies into it run at the PCIe rate, into pageable memory at a fraction of
    it); falls back to ordinary memory when pinning fails."""
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    if n == 0:
        return np.empty(shape, dtype=dtype)
    try:
        block = _PinnedBlock(n)
    except Exception:
        return np.empty(shape, dtype=dtype)
    return np.asarray(block).view(dtype).reshape(shape)


def pinned_reserve(shape, count, dtype=np.complex128):
    """Bring `count` pinned buffers of that shape into being and hand them to the library's pool: pinning GBs of host memory takes
    ~0.3 s per 4 GB, which a caller that knows the size of its results can pay before its loop instead of inside it."""
    held = [pinned_empty(shape, dtype) for _ in range(int(count))]
    n = len(held)
    del held
    return n


def to_device(arr, dev, dtype=None):
    """numpy array -> torch tensor on `dev` (contiguous, cast to dtype if given), staged through a pinned buffer of the library.

    Never hand the caller's pageable array to the runtime: for a copy of more than a few hundred KB HIP pins the caller's pages in place (a user-pointer
    registration), and when that memory goes back to the system afterwards -- numpy frees a temporary of a few MB with munmap -- the kernel driver takes EVERY
    queue of the process off the GPU until the registration has been dealt with: 15-20 ms in which nothing of this process runs, charged to whatever is
    submitted next (round 6: config 4's dpred(m), profiles/r06_config4_dpred_spread.txt; with glibc told never to unmap, all of them were gone)."""
    import torch
    a = np.ascontiguousarray(arr, dtype=dtype)
    if a.nbytes < (64 << 10):                       # (small copies go through the runtime's own staging buffer: nothing is registered)
        return torch.from_numpy(a).to(dev)
    host = pinned_empty(a.shape, a.dtype)
    host[...] = a
    return torch.from_numpy(host).to(dev)           # (synchronous: the pinned block goes back to the library's pool when `host` does)


_TORCH_NP = {'torch.complex128': np.complex128, 'torch.complex64': np.complex64, 'torch.float64': np.float64, 'torch.float32': np.float32,
             'torch.int64': np.int64, 'torch.int32': np.int32, 'torch.uint8': np.uint8}


def from_device(t):
    """torch tensor on a GPU -> numpy array, through a pinned buffer of the library (see to_device: a device-to-host copy into pageable memory registers the
    destination's pages the same way).  The array returned is ordinary memory the runtime has never seen."""
    import torch
    t = t.contiguous()
    if t.numel() * t.element_size() < (64 << 10):
        return t.cpu().numpy()
    host = pinned_empty(tuple(t.shape), _TORCH_NP[str(t.dtype)])
    torch.from_numpy(host).copy_(t)
    return np.array(host)


def wait_torch_stream(dev, spin_ms=None):
    """torch.cuda.current_stream(dev).synchronize(), polling first when helm_tuning.sync_spin_ms says so (default: it does not; include/helm.h)."""
    import time
    import torch
    st = torch.cuda.current_stream(dev)
    budget = (tuning().sync_spin_ms if spin_ms is None else spin_ms) * 1e-3
    t0 = time.perf_counter()
    while budget > 0 and not st.query():
        if time.perf_counter() - t0 > budget:
            break
    st.synchronize()


def ptr(arr):
    return arr.ctypes.data_as(ctypes.c_void_p) if arr is not None else None


def c128(arr):
    return np.ascontiguousarray(arr, dtype=np.complex128)


def f64(arr):
    return np.ascontiguousarray(arr, dtype=np.float64)
