"""CPU: dpred / Jtvec host logic (survey projection, residual back-sources, imaging condition, rank
sharding) against golden vectors from the reference middleware; arithmetic by the oracle double."""
import os
import subprocess
import sys
import numpy as np
import pytest

from tests.doubles import OracleMiniZephyrHD
from zephyr_amd.problem import Helm2DProblem
from zephyr_amd.survey import Helm2DSurvey

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, 'tests', 'golden')


def make(g, Disc=OracleMiniZephyrHD, mode='fixed', **extra):
    nz, nx = g['c'].shape
    rec = g['rec'] if mode == 'fixed' else g['rec_relative']
    sc = dict(nx=nx, nz=nz, dx=10., dz=10., c=g['c'], rho=g['rho'], nPML=6, freqs=list(g['freqs']), Disc=Disc, parallel=False,
              sterms=g['sterms'], geom=dict(src=g['src'], rec=rec, mode=mode))
    sc.update(extra)
    prob, surv = Helm2DProblem(sc), Helm2DSurvey(sc)
    prob.pair(surv)
    return prob, surv


def nrm(a, b):
    return np.linalg.norm(a - b) / np.linalg.norm(b)


def test_dpred_matches_reference():
    g = np.load(os.path.join(GOLD, 'g6_survey.npz'))
    prob, surv = make(g)
    d = surv.dpred()
    assert d.shape == g['dpred'].shape
    assert nrm(d, g['dpred']) < 1e-10
    prob2, surv2 = make(g, mode='relative')
    assert nrm(surv2.dpred(), g['dpred_relative']) < 1e-10


def test_jtvec_both_branches_match_reference():
    g = np.load(os.path.join(GOLD, 'g6_survey.npz'))
    prob, surv = make(g)
    gm = prob.Jtvec(None, g['resid'])
    assert np.iscomplexobj(gm)                         # mux branch stays complex (problem.py:152)
    assert nrm(gm, g['g_mux']) < 1e-10
    uF = prob.fields()
    assert nrm(uF[1][:, 3], g['uF_f1_src3']) < 1e-10
    gu = prob.Jtvec(None, g['resid'], u=uF)
    assert gu.dtype == np.float64                      # .real on the other branch (problem.py:162)
    assert nrm(gu, g['g_u']) < 1e-10


def test_update_model_clears_cache_only_on_change():
    g = np.load(os.path.join(GOLD, 'g6_survey.npz'))
    prob, surv = make(g)
    s0 = prob.system
    prob.updateModel(g['c'].ravel())
    assert prob.system is s0
    prob.updateModel(g['c'].ravel() * 1.01)
    assert prob.system is not s0
    assert np.allclose(prob.system.subProblems[0].c.ravel(), g['c'].ravel() * 1.01)


WORKER = r'''
import os, sys, numpy as np
sys.path.insert(0, %(root)r)
import torch.distributed as dist
dist.init_process_group('gloo', rank=int(os.environ['RANK']), world_size=int(os.environ['WORLD_SIZE']))
from tests.test_survey_gradient import make, nrm, GOLD
from zephyr_amd import parallel
g = np.load(os.path.join(GOLD, 'g6_survey.npz'))
prob, surv = make(g)
assert prob.ownedFreqs == list(range(dist.get_rank(), 3, 2))
d = surv.dpred()
gm = prob.Jtvec(None, g['resid'])
ok = nrm(d, g['dpred']) < 1e-10 and nrm(gm, g['g_mux']) < 1e-10
# every rank holds the same reduced result
chk = parallel.allreduce_sum(np.array([gm.sum()]))
ok = ok and abs(chk[0] - 2 * gm.sum()) < 1e-12 * abs(gm.sum())
print('RANK', dist.get_rank(), 'OK' if ok else 'FAIL', flush=True)
dist.barrier(); dist.destroy_process_group()
sys.exit(0 if ok else 1)
'''


def test_frequency_sharding_world_size_2_gloo(tmp_path):
    """N>1 path: two ranks share the frequencies; ONE all-reduce restores dpred and the gradient."""
    script = tmp_path / 'worker.py'
    script.write_text(WORKER % dict(root=ROOT))
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29617', WORLD_SIZE='2', PYTHONPATH=ROOT)
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(2)]
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and ('RANK %d OK' % r) in o, o


WORKER_ONE_FREQ = r'''
import os, sys, numpy as np
sys.path.insert(0, %(root)r)
import torch.distributed as dist
dist.init_process_group('gloo', rank=int(os.environ['RANK']), world_size=int(os.environ['WORLD_SIZE']))
from tests.test_survey_gradient import make, nrm, GOLD
g = dict(np.load(os.path.join(GOLD, 'g6_survey.npz')))
ref_prob, ref_surv = make(g, shardFreqs=False, freqs=[float(g['freqs'][1])], sterms=g['sterms'][1:2])
d_ref = ref_surv.dpred()
resid = np.ascontiguousarray(g['resid'].reshape((ref_surv.nrec, ref_surv.nsrc, 3))[:, :, 1:2])
g_ref = ref_prob.Jtvec(None, resid)
j_ref = ref_prob.Jvec(None, np.linspace(1., 2., g_ref.size))
prob, surv = make(g, freqs=[float(g['freqs'][1])], sterms=g['sterms'][1:2])
assert prob.ownedFreqs == ([0] if dist.get_rank() == 0 else [])      # rank 1 owns nothing and must still enter the all-reduce
ok = nrm(surv.dpred(), d_ref) < 1e-12 and nrm(prob.Jtvec(None, resid), g_ref) < 1e-12
ok = ok and nrm(prob.Jvec(None, np.linspace(1., 2., g_ref.size)), j_ref) < 1e-12
print('RANK', dist.get_rank(), 'OK' if ok else 'FAIL', flush=True)
dist.barrier(); dist.destroy_process_group()
sys.exit(0 if ok else 1)
'''


def test_single_frequency_on_two_ranks_does_not_hang(tmp_path):
    """nfreq = 1, world_size = 2: rank 0 owns every frequency, rank 1 none.  The all-reduce decision must not depend on the
    rank (`len(owned) != nfreq` let rank 0 skip the collective rank 1 was waiting in)."""
    script = tmp_path / 'worker1.py'
    script.write_text(WORKER_ONE_FREQ % dict(root=ROOT))
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29619', WORLD_SIZE='2', PYTHONPATH=ROOT)
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(2)]
    try:
        outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and ('RANK %d OK' % r) in o, o
