"""Minimal stand-in for the parts of SimPEG that zephyr.middleware touches at import time and in
dpred / Jtvec (test infrastructure, see ../README.md).  Written from the call sites:
zephyr/middleware/problem.py:17-38,87,124; survey.py:12-49,140,190-191; fields.py:11;
maps.py:9; regularization.py:11-13; optimization.py:8."""
import functools
import numpy as np


class _NS(object):
    pass


Problem, Survey, Mesh, Utils, Fields, Maps, Regularization, Optimize = (_NS() for _ in range(8))


class _BaseProblem(object):
    surveyPair = None

    def __init__(self, mesh, *args, **kwargs):
        self.mesh = mesh
        self._survey = None

    @property
    def survey(self):
        return self._survey

    def pair(self, d):
        self._survey = d
        d._prob = self

    @property
    def ispaired(self):
        return self._survey is not None


class _BaseSurvey(object):
    def __init__(self, **kwargs):
        self._prob = None
        self.srcList = []

    @property
    def prob(self):
        return self._prob

    def pair(self, p):
        p.pair(self)

    @property
    def nSrc(self):
        return len(self.srcList)


class _BaseSrc(object):
    def __init__(self, rxList, **kwargs):
        self.rxList = rxList

    @property
    def nD(self):
        return sum(rx.nD for rx in self.rxList)


class _BaseRx(object):
    def __init__(self, locs, rxType=None, **kwargs):
        self.locs = locs
        self.rxType = rxType

    @property
    def nD(self):
        return self.locs.shape[0]


class _TensorMesh(object):
    def __init__(self, h, x0=None):
        self.h = h
        n = [sum(c[1] for c in hh) for hh in h]
        self.nC = int(np.prod(n))
        self.nN = int(np.prod([k + 1 for k in n]))


def _passthrough(f):
    @functools.wraps(f)
    def wrapper(*a, **k):
        return f(*a, **k)
    return wrapper


def _requires(name):
    def deco(f):
        return f
    return deco


class _Fields(object):
    def __init__(self, mesh, survey, **kwargs):
        self.mesh, self.survey = mesh, survey


class _Empty(object):
    def __init__(self, *a, **k):
        pass


Problem.BaseProblem = _BaseProblem
Survey.BaseSurvey = _BaseSurvey
Survey.BaseSrc = _BaseSrc
Survey.BaseRx = _BaseRx
Mesh.TensorMesh = _TensorMesh
Utils.timeIt = _passthrough
Utils.count = _passthrough
Utils.requires = _requires
Utils.isScalar = lambda v: np.isscalar(v)
Utils.mkvc = lambda v, n=1: np.asarray(v).reshape((-1,) + (1,) * (n - 1))
Fields.Fields = _Fields
Maps.IdentityMap = _Empty
Regularization.BaseRegularization = _Empty
Optimize.Minimize = _Empty
