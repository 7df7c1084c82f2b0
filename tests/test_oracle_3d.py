"""CPU: the project's own 3-D operator statement against the analytic 3-D Green's function and its own sparse form."""
import numpy as np
from oracle import helm3d_oracle as h3


def test_3d_oracle_apply_equals_sparse_and_green():
    n, dx, f, c0 = 26, 10., 14., 2000.
    C = h3.helm3d_coefficients(n, n, n, c0, 1., f, dx=dx, nPML=6, cPML=300.)
    A = h3.coefficients_to_csr3(C)
    X = np.random.default_rng(0).standard_normal((n ** 3, 2)) + 0j
    assert np.abs(h3.stencil_apply3(C, X) - A @ X).max() < 1e-12 * np.abs(X).max() * np.abs(C).max() * 27
    q = np.zeros(n ** 3, complex); s = n // 2
    q[(s * n + s) * n + s] = 1.
    u = (h3.DirectOperator3(C) * q).reshape((n, n, n))
    iz, iy, ix = np.mgrid[0:n, 0:n, 0:n]
    r = dx * np.sqrt((iz - s) ** 2 + (iy - s) ** 2 + (ix - s) ** 2)
    m = (r > 2.5 * dx) & (iz > 7) & (iz < n - 8) & (iy > 7) & (iy < n - 8) & (ix > 7) & (ix < n - 8)
    g = h3.green3d(2 * np.pi * f / c0, r[m], 1.0, dx ** 3)
    assert np.linalg.norm(u[m] - g) / np.linalg.norm(g) < 8e-2
