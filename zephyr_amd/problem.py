"""Forward problem and FWI gradient on top of the frequency dispatcher.

Interface of zephyr/middleware/problem.py:17-212 (HelmBaseProblem / Helm2DProblem) without SimPEG:
`lazyFields` (forward wavefields per frequency), `Jtvec` (gradient by the zero-lag imaging
condition; both the "mux" branch that solves forward and back-propagated sources together and the
branch that re-uses given forward fields) and `updateModel`.

Multi-GPU, two ways.  In one process (no process group): the system wrapper's parallel mode deals work items
(frequency, source batch) frequency-major over the visible GPUs -- one solve thread and one prepare-ahead
thread per GPU (`zephyr_amd.dispatch`), sources of a frequency split over spare GPUs when there are fewer
frequencies than GPUs -- and the per-GPU partial gradients / data panels are summed on the host.  One
process per GPU (`shardFreqs`, default on when torch.distributed is initialised): frequencies are sharded
over ranks and the only collective is one all-reduce of the gradient (problem.py:152,162 sum over
frequencies) or of the receiver data.
"""
import numpy as np
import scipy.sparse as sp

from .base import BaseModelDependent
from .config import BaseSCCache
from .distributors import MultiFreq, ViscoMultiFreq
from .survey import HelmBaseSurvey, Helm2DSurvey, Helm25DSurvey
from . import parallel
from . import dispatch
from . import _lib

EPS = 1e-15


def _norm2(d):
    """||d||_2 without BLAS.  np.linalg.norm hands a vector of this size to a threaded dot product, and OpenBLAS's workers (64 of them on a large host) then
    spin for ~100 ms waiting for more work: under a container CPU quota (cgroup cpu.max, 16 CPUs on the GPU boxes of this project) that burns the period's
    budget in 25 ms and the scheduler freezes EVERY thread of the process -- the ones feeding the GPU included -- for the remaining 75 ms.  Round 6: this one
    call cost dpred / Jtvec 60-80 of their 110-140 ms (problem.py:51-66 compares the models the same way)."""
    d = np.asarray(d)
    return float(np.sqrt(np.add.reduce(np.square(d.real)) + (np.add.reduce(np.square(d.imag)) if np.iscomplexobj(d) else 0.0)))


class HelmBaseProblem(BaseModelDependent, BaseSCCache):

    initMap = {
        #   key              required  rename       cast
        'SystemWrapper':    (True,     None,        None),
        'shardFreqs':       (False,    '_shard',    bool),
        'hostGradient':     (False,    '_hostGradient', bool),    # force the numpy imaging condition
    }

    surveyPair = HelmBaseSurvey
    cacheItems = ['_system']

    def __init__(self, systemConfig, *args, **kwargs):
        BaseSCCache.__init__(self, systemConfig, *args, **kwargs)
        self.survey = None

    # ---- pairing (SimPEG's BaseProblem.pair) -----------------------------------------------------------
    def pair(self, survey):
        if not isinstance(survey, self.surveyPair):
            raise TypeError('%s must be paired with a %s' % (self.__class__.__name__, self.surveyPair.__name__))
        self.survey = survey
        survey.prob = self

    @property
    def ispaired(self):
        return self.survey is not None

    # ---- model -----------------------------------------------------------------------------------------
    def updateModel(self, m, loneKey='c'):
        'problem.py:51-66'
        if m is None:
            return
        if isinstance(m, dict):
            self.systemConfig.update(m)
            self.clearCache()
        elif isinstance(m, (np.ndarray, np.inexact, complex, float)):
            m = np.asarray(m)
            old = np.asarray(self.systemConfig.get(loneKey, 0.))
            if old.size != m.size or not _norm2(m.ravel() - old.ravel()) < EPS:
                self.systemConfig[loneKey] = m
                self.clearCache()
        else:
            raise Exception('Class %s doesn\'t know how to update with model of type %s' % (self.__class__.__name__, type(m)))

    def clearCache(self):
        sysw = self.__dict__.get('_system', None)
        if sysw is not None:
            del sysw.factors
        BaseSCCache.clearCache(self)

    @property
    def system(self):
        if getattr(self, '_system', None) is None:
            self._system = self.SystemWrapper(self.systemConfig)
        return self._system

    # ---- sharding --------------------------------------------------------------------------------------
    @property
    def ownedFreqs(self):
        'frequency indices this rank solves'
        nf = self.survey.nfreq
        if getattr(self, '_shard', True):
            return parallel.owned_indices(nf)
        return list(range(nf))

    @property
    def _sharded(self):
        '''True when the frequencies are split over ranks.  The decision is the same on every rank (it must be: it
        guards a collective -- a rank that owns every frequency of a short list still has to enter the all-reduce
        the ranks that own none are waiting in).'''
        return bool(getattr(self, '_shard', True)) and parallel.rank_and_size()[1] > 1

    def _solveOwned(self, rhs_list):
        'generator of (ifreq, scaleTerm * sub * rhs) over the owned frequencies'
        owned = self.ownedFreqs
        sysw = self.system
        if len(owned) == self.survey.nfreq and getattr(sysw, 'parallel', False) and hasattr(sysw, 'devices'):
            # every frequency is this process's: the wrapper's own dispatch (all visible GPUs, prepare-ahead) does the loop
            rl = list(rhs_list) if isinstance(rhs_list, (list, tuple)) else rhs_list
            for ifreq, u in enumerate(sysw * rl):
                yield ifreq, u
            return
        subs = sysw.subProblems
        scale = sysw.scaleTerm
        for ifreq in owned:
            r = rhs_list[ifreq] if isinstance(rhs_list, (list, tuple)) else rhs_list
            yield ifreq, scale * (subs[ifreq] * r)

    # ---- device-resident work items ---------------------------------------------------------------------------
    def _deviceItems(self, owned, ncols):
        """Work items (worker, operator, ifreq, c0, c1) for the owned frequencies: frequency-major over the system wrapper's
        devices (a GPU keeps the operators of its frequencies), and when there are fewer frequencies than GPUs the `ncols`
        source columns of a frequency are split over the spare ones ((frequency, source-batch) items)."""
        sysw = self.system
        subs = sysw.subProblems
        devs = list(sysw.devices) if hasattr(sysw, 'devices') else [subs[0].device]
        nw = len(devs)
        split = max(1, nw // max(1, len(owned))) if hasattr(sysw, '_replica') else 1
        split = min(split, max(1, ncols))
        items = []
        for pos, ifreq in enumerate(owned):
            bounds = [ncols * j // split for j in range(split + 1)]
            for j in range(split):
                if j == 0:          # the frequency's own operator, on the worker of the GPU it lives on
                    op = subs[ifreq]
                    w = devs.index(op.device) if op.device in devs else (pos * split) % nw
                    if split > 1 and devs[(pos * split) % nw] == op.device:
                        w = (pos * split) % nw
                else:               # a copy of it on a spare GPU for another batch of its sources
                    w = (pos * split + j) % nw
                    op = sysw._replica(ifreq, j, devs[w])
                items.append((w, op, ifreq, bounds[j], bounds[j + 1]))
        return devs, items

    def _runOnDevices(self, devs, items, fn):
        """Run fn(state, op, ifreq, c0, c1) for every item on the worker thread of its GPU (`state`: a dict private to that
        worker, for its device buffers), the factorisation of the worker's next item started ahead of time."""
        states = [dict(device=d) for d in devs]
        queues = [[] for _ in devs]
        # the factorisations of a worker's next two operators are enqueued together (discretization.prefactor_many: the fronts of both frequencies in the same
        # batched launches); an item's own prepare step then only builds and assembles its operator
        from .discretization import prefactor_many
        group = 2 if all(getattr(type(op), 'VARIANT', None) in (_lib.HELM_MINIZEPHYR, _lib.HELM_EURUS) for _, op, _, _, _ in items) else 1      # (2-D operators: what helm_prefactor_many takes)
        for w, op, ifreq, c0, c1 in items:
            def solve(_p, w=w, op=op, ifreq=ifreq, c0=c0, c1=c1):
                return fn(states[w], op, ifreq, c0, c1)
            if group > 1:
                prep = (lambda op=op: (op.handle, op)[1])
            else:
                prep = op.prefactor if hasattr(op, 'prefactor') else None
            queues[w].append(dispatch.WorkItem(solve, prep))
        pipes = dispatch.dispatch(list(zip(devs, queues)), lookahead=1, group=group, group_prepare=prefactor_many if group > 1 else None)
        try:
            out = [it.future.result() for q in queues for it in q]
        finally:
            for p in pipes:
                p.join()
        return states, out

    # ---- gradient scalers (problem.py:74-85) --------------------------------------------------------------
    def scaledTerms(self, ifreq):
        omega = 2 * np.pi * self.survey.freqs[ifreq]
        c = self.system.subProblems[ifreq].c
        return omega, c

    def gradientScaler(self, ifreq):
        omega, c = self.scaledTerms(ifreq)
        return self.survey.postProcessors[ifreq](-(omega ** 2 / c ** 3).ravel())

    def sensScaler(self, ifreq):
        'problem.py:83-85'
        omega, c = self.scaledTerms(ifreq)
        return self.survey.postProcessors[ifreq](-(c ** 3 / omega ** 2).ravel())

    # ---- forward -----------------------------------------------------------------------------------------
    def lazyFields(self, m=None):
        'generator of forward wavefields (N, nsrc) for the owned frequencies, in frequency order (problem.py:166-179)'
        if not self.ispaired:
            raise Exception('%s instance is not paired to a survey' % (self.__class__.__name__,))
        self.updateModel(m)
        qf = self.survey.getSources()
        return (u for _, u in self._solveOwned(qf))

    def fields(self, m=None):
        'list of forward wavefields for ALL frequencies on this rank (no sharding)'
        if not self.ispaired:
            raise Exception('%s instance is not paired to a survey' % (self.__class__.__name__,))
        self.updateModel(m)
        qf = self.survey.getSources()
        return list(self.system * qf)

    # ---- sensitivity times vector ------------------------------------------------------------------------
    def Jvec(self, m=None, v=None, u=None):
        """Data perturbation for a model perturbation v (problem.py:87-122): one virtual source
        `v * (-c^3/omega^2)` per frequency is solved, and dpert[:, :, f] is the outer product of the
        receiver and source samplings of that field.  Fixed receiver arrays only: the reference's
        relative-geometry branch multiplies mismatched shapes (problem.py:117-120)."""
        if not self.ispaired:
            raise Exception('%s instance is not paired to a survey' % (self.__class__.__name__,))
        if v is None:
            raise Exception('Actually, Jvec requires a perturbation vector')
        self.updateModel(m)
        sv = self.survey
        if sv.mode != 'fixed':
            raise ValueError('dimension mismatch')
        perturb = np.asarray(v).reshape((self.nz * self.nx, 1))
        qv = [sv.preProcessors[i](perturb * self.sensScaler(i).reshape((self.nz * self.nx, 1))) for i in range(sv.nfreq)]
        qf = sv.getSources()
        owned = self.ownedFreqs
        dpert = np.zeros((sv.nrec, sv.nsrc, sv.nfreq), dtype=np.complex128)
        for ifreq, uFreq in self._solveOwned(qv):
            srcTerms = qf[ifreq].T * uFreq
            recTerms = sv.rVec(0) * uFreq
            dpert[:, :, ifreq] = np.asarray(recTerms).reshape((sv.nrec, 1)) * np.asarray(srcTerms).reshape((1, sv.nsrc))
        if self._sharded:
            dpert = parallel.allreduce_sum(dpert)
        return dpert.ravel()

    # ---- gradient ----------------------------------------------------------------------------------------
    def Jtvec(self, m=None, v=None, u=None):
        """FWI gradient g = sum_f scaler_f sum_s uF (.) uB  (problem.py:124-164).

        u is None: "mux" branch -- forward and back-propagated sources are stacked column-wise and
        solved together per frequency; the result is complex (no .real), as in the reference.
        u given (list of forward fields per frequency): only the back-propagation is solved and
        the real part is returned.
        """
        if not self.ispaired:
            raise Exception('%s instance is not paired to a survey' % (self.__class__.__name__,))
        if v is None:
            raise Exception('Actually, Jtvec requires a residual vector')
        self.updateModel(m)
        sv = self.survey
        nsrc = sv.nsrc
        resid = np.asarray(v).reshape((sv.nrec, sv.nsrc, sv.nfreq))
        qb = sv.getResidualSources(resid)
        owned = self.ownedFreqs
        g = np.zeros(self.nrow, dtype=np.complex128)
        if u is None and self._deviceGradientAvailable():
            return self._JtvecDevice(qb, owned)
        if u is None:
            qf = sv.getSources()
            qm = [sp.hstack((qf[i], qb[i])) if i in owned else None for i in range(sv.nfreq)]
            for ifreq, uMux in self._solveOwned(qm):
                pp = sv.postProcessors[ifreq]
                g += self.gradientScaler(ifreq) * pp((uMux[:, :nsrc] * uMux[:, nsrc:]).sum(axis=1))
        else:
            uF = list(u)
            for ifreq, uB in self._solveOwned(qb):
                pp = sv.postProcessors[ifreq]
                g += self.gradientScaler(ifreq) * (np.asarray(uF[ifreq]) * pp(uB)).sum(axis=1)
        if self._sharded:
            g = parallel.allreduce_sum(g)
        return g if u is None else g.real

    # ---- device-resident gradient (mux branch) -------------------------------------------------------------
    def _deviceGradientAvailable(self):
        'true when the sub-problems are GPU operators and torch can hold the buffers in HBM'
        if getattr(self, '_hostGradient', False):
            return False
        subs = self.system.subProblems
        if not subs or not hasattr(subs[0], 'solveDevice'):
            return False
        from .discretization import DiscretizationWrapper
        if isinstance(subs[0], DiscretizationWrapper):        # composite sub-problems (2.5-D ky sums) own no single device operator
            return False
        try:
            from . import _lib
            if _lib.load().helm_device_count() <= 0:
                return False
            import torch
            return torch.cuda.device_count() > 0
        except Exception:
            return False

    def _JtvecDevice(self, qb, owned):
        '''mux branch with wavefields kept in HBM: per work item (frequency, source batch) upload [qf | qb] of its sources, solve them on
        the item's GPU, accumulate scaler * sum_s uF (.) uB with the imaging kernel into that GPU's partial gradient; partial gradients are
        summed on the host, then ONE all-reduce over ranks when the frequencies are sharded.'''
        import torch
        sv = self.survey
        nsrc, N = sv.nsrc, self.nrow
        scale = complex(self.system.scaleTerm)
        qf = sv.getSources()
        if not owned:
            g = np.zeros(N, dtype=np.complex128)
            return parallel.allreduce_sum(g) if self._sharded else g
        devs, items = self._deviceItems(owned, nsrc)
        # the survey's post-processor is the reference's identity (survey.py:190-196) and the scaler the problem's own: then it can be made where it is used
        plain_scaler = (type(sv).postProcessors is HelmBaseSurvey.postProcessors and type(self).gradientScaler is HelmBaseProblem.gradientScaler
                        and type(self).scaledTerms is HelmBaseProblem.scaledTerms)

        def one(wstate, op, ifreq, c0, c1):
            dev = torch.device('cuda', op.device)
            state = wstate.setdefault(('buffers', op.device), {})         # (a worker's buffers live on the GPU of the operator it is running)
            k = c1 - c0
            if 'G' not in state:
                state['G'] = torch.zeros(N, dtype=torch.complex128, device=dev)
            if state.get('cap', 0) < 2 * k:
                state['U'] = torch.empty((2 * k, N), dtype=torch.complex128, device=dev)
                state['R'] = torch.empty((2 * k, N), dtype=torch.complex128, device=dev)
                state['cap'] = 2 * k
            U, R = state['U'], state['R']
            # [qf | qb] of the item's sources as triplets, made from the two matrices' own arrays (no format conversion, no sort of 10^5..10^6 entries)
            parts = []
            for off, m in ((0, qf[ifreq] if isinstance(qf, (list, tuple)) else qf), (k, qb[ifreq])):
                mc = m if (c0 == 0 and c1 == m.shape[1]) else sp.csc_matrix(m)[:, c0:c1]
                if not (sp.isspmatrix_csr(mc) or sp.isspmatrix_csc(mc)) or not mc.has_canonical_format:
                    mc = sp.csr_matrix(mc)
                    mc.sum_duplicates()
                coo = mc.tocoo(copy=False)
                parts.append((coo.row, coo.col + off, coo.data))
            trip = (np.concatenate([p_[0] for p_ in parts]), np.concatenate([p_[1] for p_ in parts]), np.concatenate([p_[2] for p_ in parts]), (N, 2 * k))
            op.rhsFromSparseDevice(trip, R.data_ptr())      # sparse triplets up, dense on the device
            if plain_scaler:
                # -(omega^2 / c^3) scale^2 on the GPU from one upload of the model per worker: on the host the complex power and division of problem.py:74-81 cost
                # 6 ms per frequency at 512^2 (numpy), in the thread whose only other job is to keep the solve stream fed
                cm = op.c
                inv = state.setdefault('inv_c3', {}).get(id(cm))
                if inv is None:
                    cd = _lib.to_device(np.asarray(cm).ravel(), dev, np.complex128)
                    inv = state['inv_c3'][id(cm)] = 1.0 / (cd * cd * cd)
                    state.setdefault('inv_c3_keep', []).append(cm)          # (the id stays this array's while the worker lives)
                omega = 2 * np.pi * self.survey.freqs[ifreq]
                scaler = inv * complex(-(omega ** 2) * scale * scale)
            else:
                scaler = _lib.to_device(self.gradientScaler(ifreq) * scale * scale, dev, np.complex128)
            _lib.wait_torch_stream(dev)
            op.solveDevice(R.data_ptr(), U.data_ptr(), 2 * k, N)
            op.imagingAccumulateDevice(U.data_ptr(), U.data_ptr() + k * N * 16, k, scaler.data_ptr(), state['G'].data_ptr())
            return None
        states, _ = self._runOnDevices(devs, items, one)
        parts = [b['G'] for st in states for key, b in st.items() if isinstance(key, tuple) and 'G' in b]
        if len(parts) == 1:
            G = parts[0]
            if self._sharded:
                parallel.allreduce_sum_device(G)
            torch.cuda.synchronize(G.device)
            return _lib.from_device(G)
        g = np.zeros(N, dtype=np.complex128)
        for G in parts:                                   # per-GPU partial gradients: 16 B per grid point each
            torch.cuda.synchronize(G.device)
            g += _lib.from_device(G)
        return parallel.allreduce_sum(g) if self._sharded else g

    def _dpredDevice(self, owned):
        '''predicted data with the wavefields kept in HBM (fixed receiver array): per work item (frequency, source batch) the sparse sources are
        expanded on the item's GPU, solved there, and only the receiver samples R u (nrec x sources) come back'''
        import torch
        sv = self.survey
        nsrc, nrec, N = sv.nsrc, sv.nrec, self.nrow
        scale = complex(self.system.scaleTerm)
        data = np.zeros((nrec, nsrc, sv.nfreq), dtype=np.complex128)
        if not owned:
            return data
        Rm = sp.csr_matrix(sv.rVec(0))
        Rm.sum_duplicates()
        qf = sv.getSources()
        devs, items = self._deviceItems(owned, nsrc)

        def one(wstate, op, ifreq, c0, c1):
            dev = torch.device('cuda', op.device)
            state = wstate.setdefault(('buffers', op.device), {})
            k = c1 - c0
            if 'csr' not in state:
                state['csr'] = (_lib.to_device(Rm.indptr, dev, np.int64), _lib.to_device(Rm.indices, dev, np.int64), _lib.to_device(Rm.data, dev, np.complex128), nrec)
            if state.get('cap', 0) < k:
                state['R'] = torch.empty((k, N), dtype=torch.complex128, device=dev)
                state['U'] = torch.empty((k, N), dtype=torch.complex128, device=dev)
                state['out'] = torch.empty((nrec, k), dtype=torch.complex128, device=dev)
                state['cap'] = k
            R, U = state['R'], state['U']
            out = state['out'] if state['cap'] == k else torch.empty((nrec, k), dtype=torch.complex128, device=dev)
            q = qf[ifreq] if isinstance(qf, (list, tuple)) else qf
            op.rhsFromSparseDevice(sp.csc_matrix(q)[:, c0:c1], R.data_ptr())
            op.solveDevice(R.data_ptr(), U.data_ptr(), k, N)
            op.sampleDevice(U.data_ptr(), k, state['csr'], out.data_ptr())      # (returns when the samples are there: helm_sample_device waits for its own stream)
            data[:, c0:c1, ifreq] = scale * _lib.from_device(out)          # (disjoint slices per item: no two workers write the same entries)
            return None
        self._runOnDevices(devs, items, one)
        return data

    @property
    def factors(self):
        return self.system.factors

    @factors.deleter
    def factors(self):
        del self.system.factors


class Helm2DProblem(HelmBaseProblem):

    initMap = {
        'SystemWrapper':    (False,    None,        None),
    }

    surveyPair = Helm2DSurvey
    SystemWrapper = MultiFreq


class Helm2DViscoProblem(Helm2DProblem):

    SystemWrapper = ViscoMultiFreq


class Helm25DProblem(HelmBaseProblem):
    """2.5-D counterpart of Helm2DProblem (zephyr/middleware/problem.py:225-235): same machinery, paired with Helm25DSurvey; the caller's
    systemConfig names `Disc = MiniZephyr25D` (and `nky`), every frequency is a sum over cross-line wavenumbers."""

    initMap = {
        'SystemWrapper':    (False,    None,        None),
    }

    surveyPair = Helm25DSurvey
    SystemWrapper = MultiFreq


class Helm25DViscoProblem(Helm25DProblem):
    'zephyr/middleware/problem.py:236-238'

    SystemWrapper = ViscoMultiFreq

