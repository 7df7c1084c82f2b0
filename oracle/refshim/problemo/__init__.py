"""Stand-in for `problemo.BestSolver` (reference call sites: zephyr/backend/discretization.py:12,83-84,103).

Holds a sparse matrix `.A`, factors it lazily with the configured Solver (default
scipy.sparse.linalg.splu) and solves dense or sparse right-hand sides on `*`.
Test infrastructure only (see README.md).
"""
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla


class BestSolver(object):
    def __init__(self, Solver=None):
        self._Solver = Solver
        self._A = None
        self._fact = None

    @property
    def A(self):
        return self._A

    @A.setter
    def A(self, value):
        self._A = value
        self._fact = None

    def _factor(self):
        if self._fact is None:
            A = self._A.tocsc()
            if self._Solver is None:
                self._fact = spla.splu(A)
            else:
                self._fact = self._Solver(A)
        return self._fact

    def __mul__(self, rhs):
        fact = self._factor()
        if sp.issparse(rhs):
            rhs = rhs.toarray()
        rhs = np.asarray(rhs, dtype=np.complex128)
        if hasattr(fact, 'solve'):
            return fact.solve(rhs)
        return fact * rhs
