"""Survey geometry: source / receiver vectors, data projection, residual back-sources.

Interface of zephyr/middleware/survey.py:27-206 (HelmBaseSurvey / Helm2DSurvey) without the SimPEG
base classes: the arithmetic on the hot path (getSources, _lazyProjectFields,
getResidualSources, dpred) is kept, the inversion-framework glue is not.
"""
import numpy as np
import scipy.sparse as sp

from .config import BaseSCCache
from .source import SparseKaiserSource
from . import parallel


class HelmBaseSurvey(BaseSCCache):

    initMap = {
        #   key            required  rename        cast
        'geom':           (True,     None,         dict),
        'freqs':          (True,     None,         tuple),
        'sterms':         (False,    '_sterms',    np.complex128),
    }

    def __init__(self, systemConfig, **kwargs):
        BaseSCCache.__init__(self, systemConfig, **kwargs)
        self.prob = None

    # ---- geometry ------------------------------------------------------------------------------
    @property
    def nfreq(self):
        return len(self.freqs)

    @property
    def geom(self):
        return self._geom

    @geom.setter
    def geom(self, value):
        if value.get('mode', 'fixed') not in {'fixed', 'relative'}:
            raise Exception('%s objects only work with \'fixed\' or \'relative\' receiver arrays' % (self.__class__.__name__,))
        self._geom = value

    @property
    def mode(self):
        return self.geom.get('mode', 'fixed')

    @property
    def sLocs(self):
        return self.geom.get('src')

    @property
    def rLocs(self):
        return self.geom.get('rec')

    @property
    def ssTerms(self):
        return self.geom.get('sterms', np.ones((self.nsrc,), dtype=np.complex128))

    @property
    def srTerms(self):
        return self.geom.get('rterms', np.ones((self.nrec,), dtype=np.complex128))

    @property
    def tsTerms(self):
        return getattr(self, '_sterms', np.ones(self.nfreq, dtype=np.complex128))

    @property
    def nsrc(self):
        return 0 if self.sLocs is None else self.sLocs.shape[0]

    @property
    def nrec(self):
        return 0 if self.rLocs is None else self.rLocs.shape[0]

    @property
    def nD(self):
        return self.nsrc * self.nrec * self.nfreq

    @property
    def RHSGenerator(self):
        if not hasattr(self, '_RHSGenerator'):
            self._RHSGenerator = self.geom.get('GeneratorClass', SparseKaiserSource)
        return self._RHSGenerator

    # ---- source / receiver vectors (survey.py:109-128) ------------------------------------------------
    def sVecs(self):
        if not hasattr(self, '_sVecs'):
            self._sVecs = self.RHSGenerator(self.systemConfig)(self.sLocs) * sp.diags((self.ssTerms,), (0,))
        return self._sVecs

    def rVec(self, isrc):
        if self.mode == 'fixed':
            if not hasattr(self, '_rVecs'):
                self._rVecs = (self.RHSGenerator(self.systemConfig)(self.rLocs) * sp.diags((self.srTerms,), (0,))).T
            return self._rVecs
        if not hasattr(self, '_rVecs'):
            self._rVecs = {}
        if isrc not in self._rVecs:
            self._rVecs[isrc] = (self.RHSGenerator(self.systemConfig)(self.rLocs + self.sLocs[isrc]) * sp.diags((self.srTerms,), (0,))).T
        return self._rVecs[isrc]

    def rVecs(self, ifreq):
        return (self.rVec(i) for i in range(self.nsrc))

    # ---- hot-path pieces ------------------------------------------------------------------------------
    def getSources(self):
        'per-frequency source matrices qf[f] = S diag(ssTerms) conj(tsTerms[f]) (survey.py:162-169)'
        qs = self.sVecs()
        ts = self.tsTerms
        if isinstance(ts, (list, np.ndarray)):
            ts = np.asarray(ts)
            if ts.ndim < 2:
                return [qs * t.conjugate() for t in ts]
            return [qs * sp.diags((t.conjugate(),), (0,)) for t in ts]
        return qs

    def _projectOne(self, uFreq, out):
        'out[:, isrc] = R_isrc uFreq[:, isrc] for one frequency'
        if self.mode == 'fixed':
            out[:, :] = self.rVec(0) * uFreq
        else:
            for isrc in range(self.nsrc):
                out[:, isrc] = self.rVec(isrc) * uFreq[:, isrc]

    def _lazyProjectFields(self, u, owned=None):
        'data[:, isrc, ifreq] = R uF_ifreq[:, isrc] (survey.py:152-160); `owned` lists the frequency indices `u` yields'
        data = np.zeros((self.nrec, self.nsrc, self.nfreq), dtype=np.complex128)
        idx = range(self.nfreq) if owned is None else owned
        for ifreq, uFreq in zip(idx, u):
            self._projectOne(np.asarray(uFreq), data[:, :, ifreq])
        return data

    def getResidualSources(self, resid):
        'back-sources qb[f][:, s] = R_s^T resid[:, s, f] (survey.py:171-188)'
        return [sp.hstack([self.rVec(isrc).T * sp.csc_matrix(resid[:, isrc, ifreq].reshape((self.nrec, 1)))
                           for isrc in range(self.nsrc)])
                for ifreq in range(self.nfreq)]

    def dpred(self, m=None, u=None):
        'predicted data, ravel of (nrec, nsrc, nfreq) in C order (survey.py:190-198)'
        if self.prob is None:
            raise Exception('%s instance is not paired to a problem' % (self.__class__.__name__,))
        if u is None:
            owned = self.prob.ownedFreqs
            if self.mode == 'fixed' and sp.issparse(self.sVecs()) and self.prob._deviceGradientAvailable():
                self.prob.updateModel(m)
                data = self.prob._dpredDevice(owned)          # wavefields never leave HBM
            else:
                data = self._lazyProjectFields(self.prob.lazyFields(m), owned)
            if self.prob._sharded:
                data = parallel.allreduce_sum(data)
            return data.ravel()
        return self._lazyProjectFields(u).ravel()

    @property
    def postProcessors(self):
        return [lambda x: x for _ in self.freqs]

    @property
    def preProcessors(self):
        return [lambda x: x for _ in self.freqs]


class Helm2DSurvey(HelmBaseSurvey):
    pass


class Helm25DSurvey(HelmBaseSurvey):
    'zephyr/middleware/survey.py:343-346'
    pass

