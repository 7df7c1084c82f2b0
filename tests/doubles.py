"""Test doubles: product host classes whose arithmetic is done by the CPU oracle, so that the
host-side logic (dispatch, survey, gradient, rank sharding) can be exercised without a GPU."""
import scipy.sparse as sp

from oracle import helm_oracle as ho
import zephyr_amd as za


class OracleMiniZephyr(za.MiniZephyr):
    def __mul__(self, rhs):
        C = ho.minizephyr_coefficients(int(self.nz), int(self.nx), self.c, self.rho, complex(self.freq), dx=self.dx, dz=self.dz,
                                       nPML=int(self.nPML), tau=self.tau, ky=self.ky, freeSurf=self.freeSurf)
        if sp.issparse(rhs):
            rhs = rhs.toarray()
        return ho.DirectOperator(C, premul=self.premul) * rhs


class OracleMiniZephyrHD(za.MiniZephyrHD, OracleMiniZephyr):
    __mul__ = OracleMiniZephyr.__mul__


class OracleMiniZephyr25D(za.MiniZephyr25D):
    'the ky sum with CPU-oracle sub-problems'

    @property
    def Disc(self):
        return OracleMiniZephyr
