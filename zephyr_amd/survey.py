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
    # What the reference spells out as one property per key (survey.py:52-107) is a table here: attribute -> (key of `geom`, default when the
    # key is absent).  Defaults that depend on the survey are callables of it.
    _GEOM_FIELDS = {
        'mode':    ('mode',   'fixed'),
        'sLocs':   ('src',    None),
        'rLocs':   ('rec',    None),
        'ssTerms': ('sterms', lambda sv: np.ones((sv.nsrc,), dtype=np.complex128)),
        'srTerms': ('rterms', lambda sv: np.ones((sv.nrec,), dtype=np.complex128)),
    }

    def __getattr__(self, name):
        # (only reached for names that are not instance / class attributes)
        fields = type(self)._GEOM_FIELDS
        if name in fields:
            key, default = fields[name]
            geom = self.__dict__.get('_geom')
            if geom is None:
                raise AttributeError(name)
            if key in geom:
                return geom[key]
            return default(self) if callable(default) else default
        raise AttributeError('%s has no attribute %r' % (type(self).__name__, name))

    @property
    def geom(self):
        return self._geom

    @geom.setter
    def geom(self, value):
        if value.get('mode', 'fixed') not in {'fixed', 'relative'}:
            raise Exception('%s objects only work with \'fixed\' or \'relative\' receiver arrays' % (self.__class__.__name__,))
        self._geom = value

    @property
    def nfreq(self):
        return len(self.freqs)

    @property
    def tsTerms(self):
        return self.__dict__.get('_sterms', np.ones(self.nfreq, dtype=np.complex128))

    @staticmethod
    def _rows(locs):
        return 0 if locs is None else locs.shape[0]

    nsrc = property(lambda self: self._rows(self.sLocs))
    nrec = property(lambda self: self._rows(self.rLocs))
    nD = property(lambda self: self.nsrc * self.nrec * self.nfreq)

    @property
    def RHSGenerator(self):
        gen = self.__dict__.get('_RHSGenerator')
        if gen is None:
            gen = self._RHSGenerator = self.geom.get('GeneratorClass', SparseKaiserSource)
        return gen

    # ---- source / receiver vectors (survey.py:109-128) ------------------------------------------------
    def _weightedColumns(self, locs, terms):
        'one column per location from the survey\'s source generator, column j scaled by terms[j]'
        return self.RHSGenerator(self.systemConfig)(locs) * sp.diags((terms,), (0,))

    def sVecs(self):
        'source matrix S diag(ssTerms), (N, nsrc); made once'
        cache = self.__dict__.setdefault('_vecCache', {})
        if 'S' not in cache:
            cache['S'] = self._weightedColumns(self.sLocs, self.ssTerms)
        return cache['S']

    def rVec(self, isrc):
        'receiver sampling matrix of source isrc, (nrec, N): one for all sources with a fixed array, one per source when the array moves with it'
        cache = self.__dict__.setdefault('_vecCache', {})
        moving = self.mode != 'fixed'
        key = ('R', isrc) if moving else 'R'
        if key not in cache:
            where = self.rLocs + self.sLocs[isrc] if moving else self.rLocs
            cache[key] = self._weightedColumns(where, self.srTerms).T
        return cache[key]

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
        """back-sources qb[f][:, s] = R_s^T resid[:, s, f] (survey.py:171-188).  With a fixed receiver array every source shares one R, and the nsrc
        sparse products per frequency of the reference collapse into one, R^T (resid[:, :, f]) -- the same columns, built in one call."""
        if self.mode == 'fixed':
            # R^T is N x nrec with a patch of cells per receiver: only the rows some receiver touches are nonzero.  One sparse x dense product on those
            # rows per frequency (R^T restricted to them stays CSR: ~81 entries per receiver whatever nrec is -- densified it would be O(nrec^2)),
            # handed back as a CSR matrix built from its arrays (sorted rows, every column present): no sparse-sparse product, no sort.
            cache = self.__dict__.setdefault('_vecCache', {})
            if 'RtRows' not in cache:
                Rt = sp.csr_matrix(self.rVec(0).T)
                Rt.sum_duplicates()
                rows = np.flatnonzero(np.diff(Rt.indptr))
                cache['RtRows'] = (rows, sp.csr_matrix(Rt[rows, :]), Rt.shape[0])
            rows, Rsub, N = cache['RtRows']
            ns = resid.shape[1]
            indptr = np.zeros(N + 1, dtype=np.int64)
            indptr[rows + 1] = ns
            np.cumsum(indptr, out=indptr)
            indices = np.tile(np.arange(ns, dtype=np.int32), rows.size)
            out = []
            for ifreq in range(self.nfreq):
                block = np.asarray(Rsub @ np.ascontiguousarray(resid[:, :, ifreq]))      # sparse (rows, nrec) x dense (nrec, nsrc) -> dense (rows, nsrc)
                m = sp.csr_matrix((block.ravel(), indices, indptr), shape=(N, ns))
                m.has_sorted_indices = True
                out.append(m)
            return out
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

