"""GPU: the treatment of ill-conditioned fronts (NdStable, direct.hip) on models that are NOT the bench model.  VERDICT r3 item 7: the threshold
was a number fitted to the sixteen bench frequencies; it now follows from the requested tolerance (a front is re-eliminated with a pivoted LU
when its condition estimate exceeds rtol / (8 eps)), and what that has to deliver is the same on any model: every wavefield meets rtol in
ONE pass of the factors, i.e. no refinement pass.  Two models, a grid that is not a power of two (other tree shapes), eight frequencies each."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N = 768
FREQS = [2.5, 3.5, 4.5, 5.5, 6.5, 7.5, 8.5, 9.5]


def layered(nz, nx, dx):
    'flat layers with two low-velocity channels and a fast basement -- nothing of the bench generator in it'
    z = (np.arange(nz) * dx)[:, None] * np.ones((1, nx))
    v = 1500. + 0.55 * z
    for top, thick, dv in ((900., 260., -450.), (2600., 380., -700.), (4200., 600., 900.)):
        v = np.where((z >= top) & (z < top + thick), v + dv, v)
    v = np.where(z > 6100., 5200., v)
    return np.clip(v, 1500., 5500.)


def run(c, dx, freqs):
    import zephyr_amd as za
    nz, nx = c.shape
    base = dict(nx=nx, nz=nz, dx=dx, dz=dx, c=c, nPML=10, rtol=1e-10, method='direct', batch=8)
    locs = np.stack([np.linspace(300., dx * nx - 300., 8), np.full(8, 2. * dx)], axis=1)
    out = []
    for f in freqs:
        sc = dict(base, freq=f)
        op = za.Eurus(sc)
        q = za.SparseKaiserSource(sc)(locs)
        u = op * q
        assert np.isfinite(u).all()
        info = op.lastInfo
        out.append((f, max(i['iterations'] for i in info), max(i['relres'] for i in info), sum(i['status'] not in (0, 3) for i in info)))
        del op.factors
    return out


@pytest.mark.parametrize('model', ['marmousi_other_seed', 'layered'])
def test_every_wavefield_in_one_pass_on_other_models(helm_lib, model):
    from zephyr_amd.models import marmousi_like
    dx = 9.0
    c = marmousi_like(N, N, dx, seed=977) if model == 'marmousi_other_seed' else layered(N, N, dx)
    res = run(c, dx, FREQS)
    bad = [r for r in res if r[1] != 1 or r[2] > 1e-10 or r[3]]
    assert not bad, 'frequencies that needed a refinement pass or missed rtol (freq, passes, worst relres, unconverged): %s of %s' % (bad, res)


def test_the_threshold_follows_the_requested_tolerance(helm_lib, monkeypatch, capfd):
    """A looser rtol takes fewer fronts (or none), a tighter one more -- the count printed by HELM_ND_DEBUG is monotone in rtol, and every run
    meets its own tolerance."""
    import re
    import zephyr_amd as za
    from zephyr_amd.models import marmousi_like
    n, dx = 512, 10.0
    c = marmousi_like(n, n, dx)
    monkeypatch.setenv('HELM_ND_DEBUG', '1')
    counts = {}
    for rtol in (1e-8, 1e-10, 1e-12):
        sc = dict(nx=n, nz=n, dx=dx, dz=dx, c=c, nPML=10, rtol=rtol, method='direct', freq=16.0)
        op = za.Eurus(sc)
        op * za.SparseKaiserSource(sc)(np.array([[900., 20.], [3100., 20.]]))
        assert all(i['status'] in (0, 3) for i in op.lastInfo), op.lastInfo
        err = capfd.readouterr().err
        counts[rtol] = sum(int(m) for m in re.findall(r'(\d+) ill-conditioned front\(s\) re-eliminated', err))
        del op.factors
    assert counts[1e-8] <= counts[1e-10] <= counts[1e-12], counts
    assert counts[1e-12] > counts[1e-8], counts


def test_decoupled_rows_are_not_taken_for_ill_conditioning_and_factors_are_reproducible(helm_lib, monkeypatch, capfd):
    """Round 5.  The MiniZephyr system keeps identity rows on the outer boundary (norm 1) beside interior rows of norm 1e-5 (minizephyr.py:246-262 as
    restated by the oracle).  The product of the two plain infinity norms counted that scaling as a condition number of 1e6: every front touching the
    boundary was handed to the pivoted LU -- on a small model all of them, more than the 32 a group treats, and which 32 made it into the list depended on
    the order the flagging threads ran in, so two factorisations of one operator differed in their last bits.  Decoupled (diagonal-only) rows are taken out
    of the norms now and an overflowing list is cut by estimate: (a) a well-conditioned MiniZephyr operator has no front treated; (b) with the threshold
    forced so low that every group overflows, eight factorisations of one operator give bit-identical wavefields."""
    import hashlib
    import re
    import torch
    import zephyr_amd as za
    nz, nx, nrhs = 150, 170, 9
    rng = np.random.default_rng(11)
    c = 2500. + 500. * np.sin(np.arange(nz)[:, None] / 20.) * np.ones((nz, nx))
    cfg = dict(nx=nx, nz=nz, dx=10., dz=10., c=c, freq=8., nPML=8, rtol=1e-10, method='direct', batch=256)
    locs = np.stack([rng.uniform(100., 10. * nx - 100., nrhs), rng.uniform(20., 60., nrhs)], axis=1)
    q = np.ascontiguousarray(za.SparseKaiserSource(cfg)(locs).toarray())

    def solve():
        op = za.MiniZephyr(cfg)
        R = torch.from_numpy(q).cuda()
        U = torch.empty_like(R)
        op.solveDevice(R.data_ptr(), U.data_ptr(), nrhs, nz * nx, layout='node')
        torch.cuda.synchronize()
        info = [dict(i) for i in op.lastInfo]
        del op.factors
        return U.cpu().numpy(), info

    monkeypatch.setenv('HELM_ND_DEBUG', '1')
    u0, info = solve()
    err = capfd.readouterr().err
    treated = sum(int(m) for m in re.findall(r'(\d+) ill-conditioned front\(s\) re-eliminated', err))
    assert treated == 0, 'fronts of a well-conditioned MiniZephyr operator handed to the pivoted LU: %d' % treated
    assert all(i['iterations'] == 1 and i['relres'] <= 1e-12 for i in info), info
    monkeypatch.setenv('HELM_ND_STABLE_THR', '3')
    seen = set()
    for _ in range(8):
        u, info = solve()
        seen.add(hashlib.sha1(u.tobytes()).hexdigest())
        assert all(i['relres'] <= 1e-10 for i in info), info
    err = capfd.readouterr().err
    assert max(int(m) for m in re.findall(r'(\d+) ill-conditioned front\(s\) re-eliminated', err)) == 32      # (groups did overflow)
    assert len(seen) == 1, 'factorisations of one operator gave %d different wavefield arrays' % len(seen)
    assert np.linalg.norm(u - u0) <= 1e-9 * np.linalg.norm(u0)
