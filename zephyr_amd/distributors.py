"""Frequency dispatchers (interface of zephyr/backend/distributors.py:26-381).

The reference farms one sub-problem per frequency out to a multiprocessing.Pool
(distributors.py:127-173).  Here every sub-problem owns a device operator on this process's
GPU and frequencies are solved back to back on it; across GPUs the frequencies (and source
batches) are sharded over ranks, one process per GPU (`zephyr_amd.parallel`), with no
data-path collective.  The result contract is unchanged: an iterable, in `freqs` order, of
`scaleTerm * (sub * rhs_i)` arrays of shape (N, nrhs).
"""
import types
import numpy as np

from .base import BaseModelDependent
from .discretization import DiscretizationWrapper


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
    """Per-frequency dispatch (distributors.py:70-193).  `parallel`/`nWorkers` are accepted for
    compatibility; concurrency comes from the GPU batch (all sources of a frequency iterate
    together) and from rank-level sharding, not from a process pool."""

    maskKeys = {'parallel'}

    @property
    def parallel(self):
        return bool(getattr(self, '_parallel', True))

    @property
    def nWorkers(self):
        return 1

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

    def __mul__(self, rhs):
        get = self._rhs_getter(rhs)
        return (self.scaleTerm * (sub * get(i)) for i, sub in enumerate(self.subProblems))

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
