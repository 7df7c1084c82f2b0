"""The 2.5-D middleware pairing (zephyr/middleware/problem.py:225-238, survey.py:343-346): Helm25DProblem / Helm25DViscoProblem / Helm25DSurvey.
Golden g11 pins what the pairing computes from the reference's own parts (MiniZephyr25D fields per frequency + the survey's projection);
the reference's own `Helm25DProblem({'Disc': MiniZephyr25D})` raises on the 'Disc' key it hands down (oracle/make_golden.py g11 asserts that)."""
import os

import numpy as np
import pytest

import zephyr_amd as za
from zephyr_amd.problem import Helm25DProblem, Helm25DViscoProblem, Helm2DProblem
from zephyr_amd.survey import Helm25DSurvey, Helm2DSurvey
from zephyr_amd.distributors import MultiFreq, ViscoMultiFreq

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def config(Disc):
    g = np.load(os.path.join(GOLD, 'g11_25d_survey.npz'))
    nz, nx = g['c'].shape
    sc = dict(nx=nx, nz=nz, dx=10., dz=10., c=g['c'], rho=g['rho'], nPML=6, freqs=list(g['freqs']), Disc=Disc, nky=int(g['nky']), parallel=False,
              sterms=g['sterms'], geom=dict(src=g['src'], rec=g['rec'], mode='fixed'))
    return g, sc


def test_class_wiring_matches_the_reference():
    assert Helm25DProblem.surveyPair is Helm25DSurvey and Helm25DProblem.SystemWrapper is MultiFreq
    assert Helm25DViscoProblem.SystemWrapper is ViscoMultiFreq and issubclass(Helm25DViscoProblem, Helm25DProblem)
    assert Helm25DProblem.initMap == Helm2DProblem.initMap
    assert not issubclass(Helm25DSurvey, Helm2DSurvey) and not issubclass(Helm25DProblem, Helm2DProblem)


def test_dpred_of_the_25d_pairing_matches_the_reference_parts_on_the_cpu_double():
    from tests.doubles import OracleMiniZephyr25D
    g, sc = config(OracleMiniZephyr25D)
    prob, surv = Helm25DProblem(sc), Helm25DSurvey(sc)
    prob.pair(surv)
    d = surv.dpred()
    assert d.shape == g['dpred'].shape
    assert np.linalg.norm(d - g['dpred']) <= 1e-9 * np.linalg.norm(g['dpred'])
    # a frequency dispatcher hands its 'Disc' key down: the ky sum must not take itself for its sub-discretisation
    sub = prob.system.subProblems[0]
    assert isinstance(sub, za.MiniZephyr25D) and len(sub.subProblems) == int(g['nky'])
    assert not isinstance(sub.subProblems[0], za.MiniZephyr25D)
    assert za.MiniZephyr25D(dict(sc, freq=7.)).Disc is za.MiniZephyr


@pytest.mark.gpu
def test_dpred_and_gradient_of_the_25d_pairing_on_the_gpu(helm_lib):
    g, sc = config(za.MiniZephyr25D)
    sc['rtol'] = 1e-11
    prob, surv = Helm25DProblem(sc), Helm25DSurvey(sc)
    prob.pair(surv)
    d = surv.dpred()
    assert np.linalg.norm(d - g['dpred']) <= 1e-7 * np.linalg.norm(g['dpred'])
    uF = [np.asarray(u) for u in prob.lazyFields()]
    assert np.linalg.norm(uF[0][:, 2] - g['u_f0_src2']) <= 1e-7 * np.linalg.norm(g['u_f0_src2'])
    # Jtvec (mux branch, problem.py:124-164): g = sum_f gradientScaler_f * sum_s uF (.) uB with uB the fields of the back-propagated residuals
    rng = np.random.default_rng(5)
    resid = (rng.standard_normal(d.shape) + 1j * rng.standard_normal(d.shape)) * np.abs(d).mean()
    grad = prob.Jtvec(None, resid)
    qb = surv.getResidualSources(resid.reshape((surv.nrec, surv.nsrc, surv.nfreq)))
    ref = np.zeros_like(grad)
    for ifreq, sub in enumerate(prob.system.subProblems):
        uB = sub * qb[ifreq]
        ref = ref + prob.gradientScaler(ifreq) * (uF[ifreq] * np.asarray(uB)).sum(axis=1)
    assert np.linalg.norm(grad - ref) <= 1e-9 * np.linalg.norm(ref)
    del prob.factors
