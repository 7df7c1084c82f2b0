"""GPU-backed discretisation base classes.

Interface of zephyr/backend/discretization.py:18-169: an object built from a `systemConfig`
dict that supports `obj * rhs` -> conj(A^-1 (premul * rhs)).  The sparse LU behind the
reference's `Ainv` (external problemo.BestSolver, discretization.py:78-85) is replaced by libhelm on
the MI355X: assembly on the device, a sparse direct factorisation kept per assembled operator in 2-D
(Krylov methods in 3-D and as fallback), the stencil kernel checking the true residual.
"""
import copy
import ctypes
import os
import numpy as np
import scipy.sparse as sp

from . import _lib
from .base import BaseModelDependent
from .config import BaseSCCache


def default_device():
    for key in ('HELM_DEVICE', 'LOCAL_RANK'):
        if key in os.environ:
            try:
                return int(os.environ[key])
            except ValueError:
                pass
    return 0


class BaseDiscretization(BaseModelDependent):
    """Model properties, device operator handle and the `A^-1 * rhs` action (discretization.py:18-106)."""

    VARIANT = None

    initMap = {
        #   key            required  rename        cast
        'c':              (True,     '_c',         np.complex128),
        'rho':            (False,    '_rho',       np.float64),
        'freq':           (True,     None,         np.complex128),
        'Solver':         (False,    '_Solver',    None),       # accepted for compatibility; the GPU Krylov solver is used
        'tau':            (False,    '_tau',       np.float64),
        'premul':         (False,    '_premul',    np.complex128),
        # solver controls (additions of this implementation)
        'rtol':           (False,    '_rtol',      np.float64),
        'maxit':          (False,    '_maxit',     np.int64),
        'method':         (False,    '_method',    str),
        'batch':          (False,    '_batch',     np.int64),
        'checkEvery':     (False,    '_checkEvery', np.int64),
        'device':         (False,    '_device',    np.int64),
    }

    # ---- model properties -----------------------------------------------------------------
    @property
    def tau(self):
        'Laplace-domain damping time constant (discretization.py:33-36)'
        return getattr(self, '_tau', np.inf)

    @property
    def dampCoeff(self):
        return 1j / self.tau

    @property
    def premul(self):
        return getattr(self, '_premul', 1.)

    @property
    def c(self):
        'Complex wave velocity (discretization.py:49-55)'
        if isinstance(self._c, np.ndarray) and self._c.ndim > 0:
            return self._c
        return self._c * np.ones((self.nz, self.nx), dtype=np.complex128)

    @property
    def rho(self):
        'Bulk density; Gardner default 310 Re(c)^0.25 (discretization.py:57-72)'
        if hasattr(self, '_rho'):
            if not (isinstance(self._rho, np.ndarray) and self._rho.ndim > 0):
                return self._rho * np.ones((self.nz, self.nx), dtype=np.float64)
        else:
            self._rho = 310. * self.c.real ** 0.25
        return self._rho

    # ---- solver controls -------------------------------------------------------------------
    @property
    def rtol(self):
        return float(getattr(self, '_rtol', 1e-10))

    @property
    def maxit(self):
        return int(getattr(self, '_maxit', 200000))

    @property
    def method(self):
        return getattr(self, '_method', 'auto')

    @property
    def device(self):
        return int(getattr(self, '_device', default_device()))

    # ---- device operator --------------------------------------------------------------------
    def _model_arrays(self):
        'returns (c, rho, theta, eps, delta) host arrays (None where not applicable)'
        dims = (int(self.nz), int(self.nx))
        # no density given: the library evaluates the Gardner default on the device (same formula as the `rho` property)
        rho = _lib.f64(self.rho.reshape(dims)) if hasattr(self, '_rho') else None
        return _lib.c128(self.c.reshape(dims)), rho, None, None, None

    def _assemble_args(self):
        'returns (ky, cPML)'
        return 0.0, 0.0

    @property
    def handle(self):
        'The assembled device operator (lazily created; replaces the LU `Ainv`, discretization.py:78-85)'
        if getattr(self, '_handle', None) is None:
            lib = _lib.load()
            _lib.require_gpu()
            fs = (ctypes.c_int * 4)(*[1 if f else 0 for f in self.freeSurf])
            ky, cpml = self._assemble_args()
            h = lib.helm_create(self.device, self.VARIANT, int(self.nz), int(self.nx), float(self.dx), float(self.dz),
                                int(self.nPML), fs)
            if not h:
                raise _lib.HelmError(-2, _lib.last_error(None))
            try:
                c, rho, theta, eps, delta = self._model_arrays()
                _lib.check(lib.helm_set_model(h, _lib.ptr(c), _lib.ptr(rho), _lib.ptr(theta), _lib.ptr(eps), _lib.ptr(delta)), h)
                f = complex(self.freq)
                _lib.check(lib.helm_assemble(h, f.real, f.imag, float(self.tau), float(ky), float(cpml)), h)
            except Exception:
                lib.helm_destroy(h)
                raise
            self._handle = h
        return self._handle

    @property
    def Ainv(self):
        return self.handle

    def prefactor(self, nrhs=None):
        """Start the factorisation the next solve on this operator needs and return at once (helm_prefactor): the launches go
        to a high-priority stream of the handle and run beside the solves of other operators.  The reference builds its LU
        lazily inside the first `Disc * rhs` (discretization.py:78-85); its dispatcher overlaps frequencies with a process
        pool (distributors.py:161-168) -- `zephyr_amd.dispatch` does it with this call.  3-D operators build their multigrid
        hierarchy here, in the calling thread (helm_prefactor_n: `nrhs` = right-hand sides the solve will bring)."""
        m = str(self.method).lower()
        if m in ('auto', 'direct') or (m == 'mg' and getattr(self, 'heavyPrepare', False)):
            lib = _lib.load()
            lib.helm_set_tolerance_hint(self.handle, float(self.rtol))      # the factors are conditioned for the tolerance the solves will ask for
            _lib.check(lib.helm_prefactor_n(self.handle, int(nrhs or 0)), self.handle)

    def reserve(self, nrhs, rows=None, concurrent=1):
        """Bring into being what `concurrent` host-array solves of `nrhs` right-hand sides on this operator's GPU take from the
        library's pools (helm_reserve) -- called by the dispatcher before it starts its workers."""
        if str(self.method).lower() in ('auto', 'direct'):
            _lib.load().helm_reserve(self.handle, int(nrhs), int(rows if rows else self.nz * self.nx), int(concurrent))

    @Ainv.deleter
    def Ainv(self):
        h = getattr(self, '_handle', None)
        if h is not None:
            _lib.load().helm_destroy(h)
            self._handle = None

    @property
    def factors(self):
        'True when a device operator is resident (mirrors the LU-cache flag, discretization.py:91-96)'
        return getattr(self, '_handle', None) is not None

    @factors.deleter
    def factors(self):
        del self.Ainv

    def __del__(self):
        try:
            del self.factors
        except Exception:
            pass

    def __getstate__(self):
        state = dict(self.__dict__)
        state.pop('_handle', None)     # device handles do not pickle; they are rebuilt lazily
        state.pop('_A', None)
        return state

    # ---- operator views -----------------------------------------------------------------------
    def diagonals(self):
        'Coefficient planes from the device: (nblocks, 9, nz, nx) complex128'
        lib = _lib.load()
        h = self.handle
        nb = lib.helm_num_blocks(h)
        out = np.empty((nb, 9, int(self.nz), int(self.nx)), dtype=np.complex128)
        _lib.check(lib.helm_get_diagonals(h, _lib.ptr(out)), h)
        return out

    def applyForward(self, x, block=0, adjoint=False):
        'A x (or A^H x) on the device for one block; x: (N,) or (N, nrhs)'
        lib = _lib.load()
        h = self.handle
        x2, onedim = self._dense_rhs(x)
        X = np.ascontiguousarray(x2.T, dtype=np.complex128)
        Y = np.empty_like(X)
        _lib.check(lib.helm_apply(h, int(block), 1 if adjoint else 0, _lib.ptr(X), _lib.ptr(Y), X.shape[0]), h)
        return Y.T[:, 0] if onedim else Y.T

    @property
    def shape(self):
        return (self.nrow, self.nrow)

    def _solve_opts(self):
        o = _lib.SolveOpts()
        o.method = _lib.METHODS[self.method.lower()]
        o.rtol = self.rtol
        o.maxit = self.maxit
        o.check_every = int(getattr(self, '_checkEvery', 0))
        o.batch = int(getattr(self, '_batch', 0))
        o.flags = 0
        return o

    def _solve(self, rhs, rows):
        """rhs: (rows, nrhs) dense complex or scipy-sparse -> (rows, nrhs) complex128, C-order like the reference's `lu.solve` result.

        Arrays cross the C ABI in the reference's own layout (HELM_NODE_MAJOR): no transposes on the host or on the device.  A sparse
        right-hand side (the surveys' source matrices) is not densified on the host -- only its triplets go to the GPU (helm_solve_coo),
        where problemo's `b.toarray()` happens.  The result array lives in pinned host memory so the copy back runs at the PCIe rate."""
        lib = _lib.load()
        h = self.handle
        nrhs = int(rhs.shape[1])
        info = (_lib.SolveInfo * nrhs)()
        opts = self._solve_opts()
        opts.flags = _lib.HELM_NODE_MAJOR
        pm = complex(self.premul)
        nbytes = int(rows) * nrhs * 16
        U = _lib.pinned_empty((int(rows), nrhs), np.complex128) if nbytes >= (1 << 20) else np.empty((int(rows), nrhs), dtype=np.complex128)
        if sp.issparse(rhs):
            coo = sp.coo_matrix(rhs)
            coo.sum_duplicates()
            row = np.ascontiguousarray(coo.row, dtype=np.int64)
            col = np.ascontiguousarray(coo.col, dtype=np.int32)
            val = np.ascontiguousarray(coo.data, dtype=np.complex128)
            rc = lib.helm_solve_coo(h, _lib.ptr(row), _lib.ptr(col), _lib.ptr(val), int(coo.nnz), _lib.ptr(U), nrhs, int(rows),
                                    pm.real, pm.imag, ctypes.byref(opts), info)
        else:
            R = np.ascontiguousarray(rhs, dtype=np.complex128)
            rc = lib.helm_solve(h, _lib.ptr(R), _lib.ptr(U), nrhs, int(rows), pm.real, pm.imag, ctypes.byref(opts), info)
        _lib.check(rc, h)
        self.lastInfo = [dict(iterations=i.iterations, status=i.status, restarts=i.restarts, method=i.method,
                              relres=i.relres) for i in info]
        if rc > 0:
            worst = max(i.relres for i in info)
            raise ArithmeticError('%d of %d right-hand sides did not reach rtol=%g (worst relative residual %.3e, maxit=%d)'
                                  % (rc, nrhs, self.rtol, worst, self.maxit))
        return U

    def solveDevice(self, d_rhs, d_u, nrhs, rows=None, layout='rhs', support=None):
        '''Solve with right-hand sides and wavefields already resident in HBM.

        d_rhs / d_u: device pointers (ints) to complex128 buffers: layout 'rhs' = [nrhs][rows], each right-hand side contiguous;
        layout 'node' = [rows][nrhs], the reference's (N, nrhs) C-order arrays (no transposes inside the direct path).
        support: what rhsSupportFromSparse made of the sparse matrix these right-hand sides were filled from (layout 'node' only; the caller's
        guarantee that they are zero elsewhere; HELM_ND_SUPPORT_CHECK=1 verifies it).
        Returns the per-RHS info list; raises if a right-hand side misses the tolerance.'''
        lib = _lib.load()
        h = self.handle
        rows = int(self.nrow if rows is None else rows)
        if support is not None and layout == 'node':
            _lib.check(lib.helm_set_rhs_support(h, ctypes.c_void_p(support.data_ptr()), rows, int(nrhs)), h)
        info = (_lib.SolveInfo * nrhs)()
        opts = self._solve_opts()
        opts.flags = _lib.HELM_NODE_MAJOR if layout == 'node' else 0
        pm = complex(self.premul)
        rc = lib.helm_solve_device(h, ctypes.c_void_p(d_rhs), ctypes.c_void_p(d_u), int(nrhs), rows, pm.real, pm.imag,
                                   ctypes.byref(opts), info)
        _lib.check(rc, h)
        self.lastInfo = [dict(iterations=i.iterations, status=i.status, restarts=i.restarts, method=i.method,
                              relres=i.relres) for i in info]
        if rc > 0:
            raise ArithmeticError('%d of %d right-hand sides did not reach rtol=%g' % (rc, nrhs, self.rtol))
        return self.lastInfo

    def imagingAccumulateDevice(self, d_uf, d_ub, nsrc, d_scaler, d_g):
        'G += scaler * sum_s UF[s] * UB[s] on the device (zero-lag imaging condition, problem.py:152); all device pointers'
        lib = _lib.load()
        _lib.check(lib.helm_imaging_accumulate_device(self.handle, ctypes.c_void_p(d_uf), ctypes.c_void_p(d_ub), int(nsrc),
                                                      ctypes.c_void_p(d_scaler), ctypes.c_void_p(d_g)), self.handle)

    def rhsFromSparseDevice(self, q, d_rhs, layout='rhs'):
        '''Fill the device buffer d_rhs ([ncols][nrow] complex128; layout 'node': [nrow][ncols]) from the scipy-sparse right-hand-side
        matrix q (nrow x ncols) without densifying it on the host: only the COO triplets cross PCIe.'''
        import torch
        lib = _lib.load()
        if isinstance(q, tuple):                 # (rows, cols, values, shape): triplets the caller has made itself, no two of them at the same place
            r_, c_, v_, shape = q
            nnz = int(len(v_))
        else:
            # duplicates are summed where they can exist at all: a CSR / CSC matrix in canonical form has none, and its COO view needs no sort
            if sp.isspmatrix_csr(q) or sp.isspmatrix_csc(q):
                if not q.has_canonical_format:
                    q = q.copy()
                    q.sum_duplicates()
                coo = q.tocoo(copy=False)
            else:
                coo = sp.coo_matrix(q)
                coo.sum_duplicates()
            r_, c_, v_, shape, nnz = coo.row, coo.col, coo.data, coo.shape, int(coo.nnz)
        dev = torch.device('cuda', self.device)
        row = _lib.to_device(r_, dev, np.int64)
        col = _lib.to_device(c_, dev, np.int32)
        val = _lib.to_device(v_, dev, np.complex128)
        _lib.wait_torch_stream(dev)                         # (the copies of THIS thread: a device-wide synchronisation would wait for the other workers' kernels too)
        _lib.check(lib.helm_rhs_from_coo_device_layout(self.handle, ctypes.c_void_p(row.data_ptr()), ctypes.c_void_p(col.data_ptr()),
                                                       ctypes.c_void_p(val.data_ptr()), nnz, ctypes.c_void_p(d_rhs), int(shape[1]),
                                                       int(shape[0]), _lib.HELM_RHS_NODE_MAJOR if layout == 'node' else 0), self.handle)

    def rhsSupportFromSparse(self, q):
        '''The support of the scipy-sparse right-hand-side matrix q (nrow x ncols, ncols <= 512) as solveDevice(..., support=) takes it: a uint8 device
        tensor, one byte per row, bit b = block b of 64 columns has an entry in that row.  Where a source matrix is nonzero is part of the reference's
        input (scipy-sparse sources, survey.py:86-89,162-188); with it the direct path does not read the dense right-hand sides to find out.'''
        import torch
        lib = _lib.load()
        coo = sp.coo_matrix(q)
        dev = torch.device('cuda', self.device)
        row = _lib.to_device(coo.row, dev, np.int64)
        col = _lib.to_device(coo.col, dev, np.int32)
        bits = torch.zeros(((int(coo.shape[0]) + 3) // 4) * 4, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize(dev)
        _lib.check(lib.helm_rhs_support_from_coo(self.handle, ctypes.c_void_p(row.data_ptr()), ctypes.c_void_p(col.data_ptr()), int(coo.nnz),
                                                 ctypes.c_void_p(bits.data_ptr()), int(coo.shape[0]), int(coo.shape[1])), self.handle)
        return bits

    def sampleDevice(self, d_u, nsrc, csr_dev, d_out):
        'd_out[nrec][nsrc] = R u for the CSR receiver matrix uploaded by the caller: csr_dev = (rowptr, col, val, nrec) device tensors'
        rowptr, col, val, nrec = csr_dev
        _lib.check(_lib.load().helm_sample_device(self.handle, ctypes.c_void_p(d_u), int(nsrc), int(self.nrow), ctypes.c_void_p(rowptr.data_ptr()),
                                                  ctypes.c_void_p(col.data_ptr()), ctypes.c_void_p(val.data_ptr()), int(nrec),
                                                  ctypes.c_void_p(d_out)), self.handle)

    def setProfiling(self, on=True):
        'time every stencil-apply launch of subsequent solves with HIP events on the solver stream'
        _lib.check(_lib.load().helm_set_profiling(self.handle, 1 if on else 0), self.handle)

    def lastTiming(self):
        t = _lib.Timing()
        _lib.check(_lib.load().helm_last_timing(self.handle, ctypes.byref(t)), self.handle)
        return dict(solve_ms=t.solve_ms, apply_ms=t.apply_ms, apply_launches=t.apply_launches, apply_bytes=t.apply_bytes,
                    factor_ms=t.factor_ms, gemm_ms=t.gemm_ms, gemm_launches=t.gemm_launches, gemm_flops=t.gemm_flops,
                    gemm_big_ms=t.gemm_big_ms, gemm_big_launches=t.gemm_big_launches, gemm_big_flops=t.gemm_big_flops,
                    gemm_bytes=t.gemm_bytes, gemm_sol_ms=t.gemm_sol_ms)

    @staticmethod
    def _dense_rhs(rhs):
        if sp.issparse(rhs):
            rhs = rhs.toarray()
        rhs = np.asarray(rhs, dtype=np.complex128)
        onedim = rhs.ndim == 1
        if onedim:
            rhs = rhs.reshape((-1, 1))
        return rhs, onedim

    @staticmethod
    def _as_rhs(rhs):
        'like _dense_rhs, but a scipy-sparse right-hand side stays sparse (it is expanded on the GPU)'
        if sp.issparse(rhs):
            return rhs, False
        return BaseDiscretization._dense_rhs(rhs)

    def __mul__(self, rhs):
        'conj(A^-1 (premul * rhs))  (discretization.py:101-103)'
        rhs, onedim = self._as_rhs(rhs)
        if rhs.shape[0] != self.nrow:
            raise ValueError('dimension mismatch')
        u = self._solve(rhs, self.nrow)
        return u[:, 0] if onedim else u

    def __call__(self, value):
        return self * value


def prefactor_many(ops):
    """Start the factorisations the next solves on `ops` need in the SAME launches and return at once (helm_prefactor_many): operators of one grid on one GPU,
    at most four; the latency-bound top of the elimination tree is then walked once for all of them.  Every operator's factors are bit for bit what
    `op.prefactor()` would have made; operators the library cannot take together are prefactored one by one.  The reference's dispatcher overlaps
    frequencies with a process pool (distributors.py:161-168) -- `zephyr_amd.dispatch` groups its prepare step around this call."""
    ops = [op for op in ops if op is not None]
    if not ops:
        return
    direct = [op for op in ops if getattr(type(op), 'VARIANT', None) in (_lib.HELM_MINIZEPHYR, _lib.HELM_EURUS) and str(getattr(op, 'method', '')).lower() in ('auto', 'direct')]
    for op in ops:
        if op not in direct and hasattr(op, 'prefactor'):
            op.prefactor()
    lib = _lib.load()
    for i in range(0, len(direct), 4):
        part = direct[i:i + 4]
        for op in part:
            lib.helm_set_tolerance_hint(op.handle, float(op.rtol))       # the factors are conditioned for the tolerance the solves will ask for
        hs = (ctypes.c_void_p * len(part))(*[op.handle for op in part])
        _lib.check(lib.helm_prefactor_many(hs, len(part)), part[0].handle)


class DiscretizationWrapper(BaseSCCache):
    """Composite of sub-problems built from per-sub-problem config updates (discretization.py:109-169)."""

    initMap = {
        'Disc':           (True,     None,         None),
        'scaleTerm':      (False,    '_scaleTerm', np.complex128),
    }

    maskKeys = {'scaleTerm'}
    cacheItems = ['_subProblems']

    @property
    def scaleTerm(self):
        return getattr(self, '_scaleTerm', 1.)

    @property
    def _spConfigs(self):
        updates = list(self.spUpdates)
        base = copy.copy(self.systemConfig)
        # one complex128 copy of the velocity model for ALL sub-problems (read-only, so that config.cast_value hands it on as it is): every Disc used to make its
        # own -- 0.25 ms per frequency at 512^2, 1.2 ms at 1024^2, in the calling thread before anything reaches the GPU
        c = base.get('c')
        if isinstance(c, np.ndarray) and c.ndim > 0 and not any('c' in u for u in updates):
            shared = np.array(c, dtype=np.complex128)
            shared.setflags(write=False)
            base['c'] = shared

        def merged(update):
            cfg = copy.copy(base)
            cfg.update(update)
            return cfg
        return (merged(update) for update in updates)

    @property
    def subProblems(self):
        if getattr(self, '_subProblems', None) is None:
            self._subProblems = [self.Disc(cfg) for cfg in self._spConfigs]
        return self._subProblems

    @property
    def factors(self):
        subs = getattr(self, '_subProblems', None)
        return subs is not None and any(sub.factors for sub in subs)

    @factors.deleter
    def factors(self):
        subs = getattr(self, '_subProblems', None)
        if subs is not None:
            for sub in subs:
                del sub.factors

    @property
    def spUpdates(self):
        raise NotImplementedError

    def __mul__(self, rhs):
        raise NotImplementedError
