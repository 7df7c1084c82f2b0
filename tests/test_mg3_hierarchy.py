"""Host logic of the 3-D layer-preserving multigrid hierarchy (zephyr_amd/csrc/mg3d.hip): axis coarsening, transfer
tables and the 1-D Laplacian factors on the stretched axes, through the host-only diagnostic `helm_mg3_axis` (no GPU needed)."""
import ctypes

import numpy as np

from oracle import helm3d_oracle as h3


def ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def axis(lib, n, npml, h, cpml, om, level):
    nc = lib.helm_mg3_axis(n, npml, h, cpml, om.real, om.imag, level, None, None, None, None, None, None, None, None)
    assert nc > 0
    x = np.zeros(nc); lay = np.zeros(nc, np.int32); lap = np.zeros(6 * nc)
    pc = np.zeros(2 * nc, np.int32); pw = np.zeros(2 * nc); nn = ctypes.c_int(0)
    rf = np.zeros(nc, np.int32); rw = np.zeros(3 * nc)
    assert lib.helm_mg3_axis(n, npml, h, cpml, om.real, om.imag, level, ptr(x), ptr(lay), ptr(lap), ptr(pc), ptr(pw), ctypes.byref(nn), ptr(rf), ptr(rw)) == nc
    ncn = nn.value
    return dict(x=x, lay=lay.astype(bool), lap=(lap[0::2] + 1j * lap[1::2]).reshape(3, nc), pc=pc.reshape(nc, 2), pw=pw.reshape(nc, 2),
                ncn=ncn, rf=rf[:ncn], rw=rw[:3 * ncn].reshape(ncn, 3))


def test_layers_survive_and_the_interior_halves(helm_lib):
    n, npml, h = 256, 10, 10.
    sizes = []
    for level in range(4):
        a = axis(helm_lib, n, npml, h, 300., 2 * np.pi * 5. + 0j, level)
        sizes.append(a['x'].size)
        # every layer node of the fine axis is still there, at its own coordinate
        assert np.array_equal(a['x'][:npml], h * np.arange(npml)) and np.array_equal(a['x'][-npml:], h * np.arange(n - npml, n))
        assert a['lay'][:npml].all() and a['lay'][-npml:].all() and not a['lay'][npml:-npml].any()
        inner = np.diff(a['x'][npml - 1:-npml])                         # from the last left layer node through the interior
        assert np.allclose(inner, h * 2 ** level)                      # uniform interior, spacing doubled per level
        assert 0 < a['x'][-npml] - a['x'][-npml - 1] <= h * 2 ** level  # (the last interior node may sit closer to the right layer)
        assert np.all(np.diff(a['x']) > 0)
    assert sizes == [256, 138, 79, 49]
    assert helm_lib.helm_mg3_axis(8, 5, 1., 1., 1., 0., 0, None, None, None, None, None, None, None, None) < 0


def test_transfer_tables_are_interpolation_and_its_normalised_transpose(helm_lib):
    for n, npml, level in ((256, 10, 0), (256, 10, 2), (61, 6, 0), (47, 8, 1), (30, 6, 0)):
        a = axis(helm_lib, n, npml, 7.5, 300., 2 * np.pi * 4. + 0j, level)
        nc, ncn = a['x'].size, a['ncn']
        b = axis(helm_lib, n, npml, 7.5, 300., 2 * np.pi * 4. + 0j, level + 1)
        assert b['x'].size == ncn
        P = np.zeros((nc, ncn))
        for i in range(nc):
            P[i, a['pc'][i, 0]] += a['pw'][i, 0]
            P[i, a['pc'][i, 1]] += a['pw'][i, 1]
        assert np.allclose(P.sum(axis=1), 1.)
        assert np.allclose(P @ b['x'], a['x'])                          # linear interpolation reproduces the coordinates
        R = np.zeros((ncn, nc))
        for I in range(ncn):
            f = a['rf'][I]
            for d in (-1, 0, 1):
                if a['rw'][I, d + 1] != 0.:
                    R[I, f + d] = a['rw'][I, d + 1]
        assert np.allclose(R.sum(axis=1), 1.)
        Rt = P.T / P.T.sum(axis=1, keepdims=True)
        assert np.allclose(R, Rt)
        assert np.array_equal(a['x'][a['rf']], b['x'])                  # the centre of a coarse node is the node itself


def test_laplacian_factors_match_the_oracle_profile_on_the_uniform_axis(helm_lib):
    n, npml, h, cpml, om = 64, 10, 12.5, 300., 2 * np.pi * 3. - 0.4j
    a = axis(helm_lib, n, npml, h, cpml, om, 0)
    lm, l0, lp = h3._lap_terms(h3.pml_profile(n, npml, h, cpml, om))
    for got, want in zip(a['lap'], (lm, l0, lp)):
        assert np.allclose(got, want / h ** 2, rtol=1e-13, atol=0)
    # coarse axis: the factors are those of the non-uniform three-point formula, second-order exact on quadratics where xi = 1
    c = axis(helm_lib, n, npml, h, 0., om, 1)
    x = c['x']
    u = 0.5 * x ** 2 - 3. * x
    d2 = c['lap'][0][1:-1] * u[:-2] + c['lap'][1][1:-1] * u[1:-1] + c['lap'][2][1:-1] * u[2:]
    assert np.allclose(d2, 1.)
