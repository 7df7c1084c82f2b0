"""Frequency dispatchers (interface of zephyr/backend/distributors.py:26-381).

The reference farms one sub-problem per frequency out to a multiprocessing.Pool
(distributors.py:127-173).  Here every sub-problem owns a device operator; with `parallel` (the
default) the frequencies are dealt over the GPUs this process can see, one solve thread and one
prepare-ahead thread per GPU (`zephyr_amd.dispatch`: the factorisation of the next frequency is
started while the current one is being solved), and when there are fewer frequencies than GPUs the
sources of a frequency are split over the spare ones.  Under a one-process-per-GPU launcher
(`zephyr_amd.parallel`) a process keeps to its own GPU.  No data-path collective either way.  The
result contract is unchanged: an iterable, in `freqs` order, of `scaleTerm * (sub * rhs_i)` arrays
of shape (N, nrhs).
"""
import os
import types
import numpy as np
import scipy.sparse as sp

from .base import BaseModelDependent
from .discretization import DiscretizationWrapper
from . import dispatch


class BaseDist(DiscretizationWrapper):
    """Wrapper base with dispatcher chaining (distributors.py:26-67)."""

    initMap = {
        'Disc':           (True,     '_Disc',      None),
        'parallel':       (False,    '_parallel',  bool),
        'nWorkers':       (False,    '_nWorkers',  np.int64),
        'remDists':       (False,    None,         list),
    }

    maskKeys = {'remDists'}

    @property
    def remDists(self):
        return getattr(self, '_remDists', [])

    @remDists.setter
    def remDists(self, value):
        if value:
            self._DiscOverride = value.pop(0)
        self._remDists = value

    @property
    def Disc(self):
        return getattr(self, '_DiscOverride', self._Disc)

    @property
    def addFields(self):
        return {'remDists': self.remDists}


class BaseMPDist(BaseDist):
    """Per-frequency dispatch (distributors.py:70-193).

    `parallel` (default True): work items (frequency, source batch) are dealt frequency-major over the visible GPUs and run
    by `zephyr_amd.dispatch` -- the counterpart of the reference's `multiprocessing.Pool`; `nWorkers` caps the number of
    GPUs used (reference: pool size).  `parallel=False`: frequencies back to back on the default device."""

    maskKeys = {'parallel'}

    @property
    def parallel(self):
        return bool(getattr(self, '_parallel', True))

    @property
    def devices(self):
        'GPUs of the parallel mode (HELM_DEVICES / all visible ones; a single one under a per-GPU launcher), at most nWorkers of them'
        if not self.parallel:
            return [self.systemConfig['device']] if 'device' in self.systemConfig else dispatch.visible_devices()[:1]
        if 'device' in self.systemConfig:                      # the caller pinned the operators to one GPU
            return [int(self.systemConfig['device'])]
        devs = dispatch.visible_devices()
        cap = int(getattr(self, '_nWorkers', len(devs)))
        return devs[:max(1, cap)]

    @property
    def nWorkers(self):
        return len(self.devices)

    @property
    def subProblems(self):
        'sub-problem i lives on device i mod (number of devices): a GPU keeps the operators (and factors) of its frequencies'
        if getattr(self, '_subProblems', None) is None:
            devs = self.devices
            subs = []
            split = max(1, len(devs) // max(1, len(self.spUpdates)))      # spare GPUs: see __mul__
            for i, cfg in enumerate(self._spConfigs):
                if self.parallel and 'device' not in cfg:
                    cfg['device'] = devs[(i * split) % len(devs)]
                subs.append(self.Disc(cfg))
            self._subProblems = subs
            self._replicas = {}
        return self._subProblems

    def _replica(self, i, r, device):
        'copy r >= 1 of sub-problem i on another GPU (fewer frequencies than GPUs: its sources are split)'
        key = (i, r)
        reps = self.__dict__.setdefault('_replicas', {})
        if key not in reps:
            cfg = list(self._spConfigs)[i]
            cfg['device'] = device
            reps[key] = self.Disc(cfg)
        return reps[key]

    @staticmethod
    def _rhs_getter(rhs):
        'RHS routing of distributors.py:139-159: list -> rhs[i]; generator -> next; single array shared'
        if isinstance(rhs, list):
            def get(i):
                r = rhs[i]
                return r.reshape((r.size, 1)) if getattr(r, 'ndim', 2) < 2 else r
        elif isinstance(rhs, types.GeneratorType):
            def get(i):
                return next(rhs)
        else:
            shared = rhs.reshape((rhs.size, 1)) if rhs.ndim < 2 else rhs

            def get(i):
                return shared
        return get

    def _scaled(self, u):
        'scaleTerm * u without a second copy of a multi-GB wavefield array: in place where the result is this call\'s own complex array'
        st = self.scaleTerm
        if st == 1.:
            return u
        if isinstance(u, np.ndarray) and u.dtype == np.complex128 and u.flags.writeable:
            u *= st
            return u
        return st * u

    @staticmethod
    def _item(sub, r, throttle=None):
        prep = None
        if hasattr(sub, 'prefactor'):
            ncol = int(r.shape[1]) if getattr(r, 'ndim', 1) > 1 else 1

            def prep():
                try:
                    sub.prefactor(ncol)   # builds the handle (assembly on the GPU) and enqueues the factorisation / builds the 3-D hierarchy
                except TypeError:
                    sub.prefactor()

        def solve(_prepared):
            if throttle is not None:
                throttle.acquire()        # not more than one finished result of this worker waiting for the consumer
            return sub * r
        it = dispatch.WorkItem(solve, prep)
        it.owner = sub
        it.nrow = int(r.shape[0])
        it.ncol = int(r.shape[1]) if getattr(r, 'ndim', 1) > 1 else 1
        return it

    def __mul__(self, rhs):
        get = self._rhs_getter(rhs)
        subs = self.subProblems
        if not self.parallel:
            return (self._scaled(sub * get(i)) for i, sub in enumerate(subs))
        # a previous call whose results were not drained must not keep driving the same operator handles (they are not thread-safe): its workers
        # are stopped, and if it has not started any yet its generation token is stale from here on -- its first next() raises instead of
        # starting a second set of pipelines on the handles this call owns
        self._stop_workers()
        gen = self.__dict__['_generation'] = self.__dict__.get('_generation', 0) + 1
        # every right-hand side is taken now, in order, like the reference's apply_async loop (distributors.py:161-166)
        devs = self.devices
        nd = len(devs)
        split = max(1, nd // max(1, len(subs)))                  # GPUs per frequency when there are spare ones
        # results go back to the host here (GBs per frequency over PCIe): two workers per GPU, so that the copy of one frequency's wavefields
        # runs while the other worker's frequency is being solved (HELM_WORKERS_PER_DEVICE overrides)
        wpd = dispatch.workers_per_device(3)
        workers = devs * wpd                                     # worker k drives GPU workers[k]; one solve + one prepare thread each
        queues = [[] for _ in workers]
        # (a result is GBs of pinned memory: HELM_RESULTS_AHEAD waiting per worker, one in the making)
        throttles = [dispatch.Throttle(dispatch.results_ahead(2)) for _ in workers]
        turn = [0] * nd

        def worker_of(slot):                                     # the workers of a GPU take its items in turn
            w = slot + nd * (turn[slot] % wpd)
            turn[slot] += 1
            return w
        parts = []
        for i, sub in enumerate(subs):
            r = get(i)
            ncol = r.shape[1] if getattr(r, 'ndim', 1) > 1 else 1
            k = min(split, ncol) if hasattr(sub, 'prefactor') else 1
            if k <= 1:
                w = worker_of((i * split) % nd)
                it = self._item(sub, r, throttles[w])
                queues[w].append(it)
                parts.append([(it, w)])
                continue
            rc = r.tocsc() if sp.issparse(r) else r
            bounds = [ncol * j // k for j in range(k + 1)]
            row = []
            for j in range(k):
                w = worker_of((i * split + j) % nd)
                owner = sub if j == 0 else self._replica(i, j, workers[w])
                it = self._item(owner, rc[:, bounds[j]:bounds[j + 1]], throttles[w])
                queues[w].append(it)
                row.append((it, w))
            parts.append(row)
        self._throttles = throttles
        strict = any(getattr(s_, 'heavyPrepare', False) for s_ in subs[:1])      # (3-D operators build their preconditioner in the prepare step: strictly one item ahead of the solve)

        def start():
            if self.__dict__.get('_generation') != gen:
                raise RuntimeError('this result was superseded by a later `wrapper * rhs` on the same wrapper before it was iterated: '
                                   'drain (or drop) one product before forming the next -- the operator handles serve one call at a time')
            # what `wpd` concurrent solves on a GPU take from the library's pools is brought into being before the workers start: a
            # hipMalloc issued beside running kernels and copies can take a second (helm_reserve)
            booked = {}
            for w, q in enumerate(queues):
                if q:
                    booked.setdefault(workers[w], []).append(q[0])
            for dev, firsts in booked.items():
                owner = firsts[0].owner
                if hasattr(owner, 'reserve'):
                    owner.reserve(max(it.ncol for w, q in enumerate(queues) if workers[w] == dev for it in q), rows=firsts[0].nrow, concurrent=len(firsts))
            hook = self.__dict__.get('_after_reserve')            # (tests: called once the bookings are made and before any worker starts)
            if hook:
                hook()
            self._pipes = dispatch.dispatch(list(zip(workers, queues)), lookahead=1, strict=strict)

        def results():
            # the workers start with the first result that is asked for (the body of a generator runs from the first next() on): a result object
            # that is dropped without being iterated never starts a thread, one that is dropped half-way closes its throttles in `finally`
            # (which the interpreter runs when the generator is collected)
            try:
                start()
                for row in parts:
                    cols = []
                    for it, w in row:
                        cols.append(it.future.result())
                        it.future = None             # the item must not keep a multi-GB result alive after it has been handed over
                        throttles[w].release()
                    u = cols[0] if len(cols) == 1 else np.hstack(cols)
                    del cols
                    yield self._scaled(u)
                    del u
            finally:
                for t in throttles:                  # an abandoned generator must not leave the workers waiting
                    t.close()
        return results()

    @property
    def factors(self):
        if DiscretizationWrapper.factors.fget(self):
            return True
        return any(rep.factors for rep in self.__dict__.get('_replicas', {}).values())

    def _stop_workers(self):
        for t in self.__dict__.get('_throttles', []):      # (a half-consumed result generator must not keep the workers waiting)
            t.close()
        for p in self.__dict__.get('_pipes', []):
            p.join()
        self._pipes = []
        self._throttles = []

    @factors.deleter
    def factors(self):
        self._stop_workers()
        DiscretizationWrapper.factors.fdel(self)
        for rep in self.__dict__.get('_replicas', {}).values():
            del rep.factors

    def __del__(self):
        try:
            del self.factors
        except Exception:
            pass


class MultiFreq(BaseMPDist):
    """One sub-problem per entry of `freqs` (distributors.py:243-265)."""

    initMap = {
        'freqs':          (True,     None,         list),
    }

    maskKeys = {'freqs'}

    @property
    def spUpdates(self):
        updates = []
        for freq in self.freqs:
            u = {'freq': freq}
            u.update(self.addFields)
            updates.append(u)
        return updates


class SerialMultiFreq(MultiFreq):
    """MultiFreq that never used the pool in the reference (distributors.py:362-381); identical here."""

    @property
    def parallel(self):
        return False

    @property
    def addFields(self):
        return {}


class ViscoMultiFreq(MultiFreq, BaseModelDependent):
    """MultiFreq with a complex velocity from Q and Kolsky-Futterman dispersion per frequency
    (distributors.py:268-359)."""

    initMap = {
        'c':              (True,     None,         np.float64),
        'Q':              (False,    None,         np.float64),
        'freqBase':       (False,    None,         np.float64),
    }

    maskKeys = {'freqs', 'c', 'Q', 'freqBase'}

    @staticmethod
    def _any(criteria):
        if type(criteria) in (bool, np.bool_):
            return criteria
        return np.any(criteria)

    @property
    def freqBase(self):
        return getattr(self, '_freqBase', 0.)

    @freqBase.setter
    def freqBase(self, value):
        assert value >= 0
        self._freqBase = value

    @property
    def Q(self):
        if hasattr(self, '_Q'):
            if not (isinstance(self._Q, np.ndarray) and self._Q.ndim > 0):
                return self._Q * np.ones((self.nz, self.nx), dtype=np.float64)
        else:
            self._Q = np.inf
        return self._Q

    @Q.setter
    def Q(self, value):
        assert not self._any(np.asarray(value) <= 0)
        self._Q = value

    @property
    def disperseFreqs(self):
        return self._any(self.Q != np.inf) and (self.freqBase > 0)

    @property
    def spUpdates(self):
        updates = []
        for freq in self.freqs:
            if self.disperseFreqs:
                fact = 1. + (np.log(freq / self.freqBase) / (np.pi * self.Q))
                assert not self._any(fact < 0.1)
                cR = fact * self.c
                c = cR + (0.5j * cR / self.Q)            # + because of the FT convention
            else:
                c = np.ravel(self.c) + (0.5j * np.ravel(self.c) / np.ravel(self.Q))
            u = {'freq': freq, 'c': c}
            u.update(self.addFields)
            updates.append(u)
        return updates
