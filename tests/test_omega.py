"""OMEGA project I/O, job profiles, CLI and Jvec against vectors produced by the reference (oracle/make_golden.py g10)."""
import os
import shutil

import numpy as np
import pytest

from zephyr_amd import omega, jobs, cli
from zephyr_amd.problem import Helm2DProblem
from zephyr_amd.survey import Helm2DSurvey
from tests.doubles import OracleMiniZephyrHD

GOLD = os.path.join(os.path.dirname(__file__), 'golden')
FX = os.path.join(GOLD, 'xhlayr')


@pytest.fixture(scope='module')
def g10():
    return np.load(os.path.join(GOLD, 'g10_omega.npz'))


@pytest.fixture()
def project(tmp_path):
    for fn in ('xhlayr.ini', 'xhlayr.vp'):
        shutil.copyfile(os.path.join(FX, fn), tmp_path / fn)
    return str(tmp_path / 'xhlayr')


def test_readini_matches_reference(g10):
    ini = omega.readini(os.path.join(FX, 'xhlayr.ini'))
    checked = 0
    for key in g10.files:
        if not key.startswith('ini_') or key == 'ini_strings':
            continue
        ref = g10[key]
        mine = np.asarray(ini[key[4:]])
        assert mine.shape == ref.shape, key
        assert np.array_equal(mine, ref), key
        checked += 1
    assert checked >= 40
    assert [ini['datain'], ini['dataout'], ini['we']] == list(g10['ini_strings'])
    assert ini['nom'] == 50 and ini['ns'] == 86 and ini['nr'] == 86 and ini['slices' if 'slices' in ini else 'nslices'] == 0


def test_segy_ibm_decoder(g10):
    sf = omega.SEGYFile(os.path.join(FX, 'xhlayr.vp'))
    assert (sf.ntr, sf.ns, sf.fmt) == (100, 200, 1)
    g9 = np.load(os.path.join(GOLD, 'g9_xhlayr.npz'))
    assert np.array_equal(sf[:].T, g9['c'])
    assert np.array_equal(sf[:].T, g10['sc_c'])
    # known IBM words: 0x42640000 = 100.0, 0xC276A000 = -118.625, 0 = 0
    assert np.array_equal(omega.ibm2ieee(np.array([0x42640000, 0xC276A000, 0], dtype=np.uint32)), [100.0, -118.625, 0.0])


def test_segy_ieee_format(tmp_path):
    raw = bytearray(3600)
    raw[3220:3222] = (5).to_bytes(2, 'big')
    raw[3224:3226] = (5).to_bytes(2, 'big')
    tr = (np.arange(15).reshape((3, 5)) * 1.5).astype('>f4')
    for t in range(3):
        raw += bytes(240) + tr[t].tobytes()
    p = tmp_path / 'm.segy'
    p.write_bytes(bytes(raw))
    sf = omega.SEGYFile(str(p))
    assert np.array_equal(sf[:], tr.astype(np.float64)) and len(sf) == 3
    raw[3224:3226] = (3).to_bytes(2, 'big')
    p.write_bytes(bytes(raw))
    with pytest.raises(NotImplementedError):
        omega.SEGYFile(str(p))


def test_datastore_systemconfig(project, g10):
    ds = omega.FullwvDatastore(project)
    assert '.vp' in ds and '.qp' not in ds and len(ds.keys()) == 1
    sc = ds.systemConfig
    assert np.array_equal(sc['c'], g10['sc_c'])
    assert np.array_equal(sc['freqs'], g10['sc_freqs'])
    assert np.array_equal(sc['geom']['src'], g10['sc_src']) and np.array_equal(sc['geom']['rec'], g10['sc_rec'])
    scal = [sc['nx'], sc['nz'], sc['dx'], sc['dz'], sc['xorig'], sc['zorig'], sc['nky'], sc['ireg'], sc['freqBase'], sc['tau']]
    assert np.array_equal(np.array(scal, dtype=np.float64), g10['sc_scalars'])
    assert tuple(sc['freeSurf']) == tuple(bool(v) for v in g10['sc_freeSurf'])
    with pytest.raises(KeyError):
        ds['.rho']
    with pytest.raises(Exception):
        omega.FullwvDatastore(project + '_missing')


def test_utout_writer_bytes(tmp_path, g10):
    freqs = [float(g10['sc_freqs'][i]) for i in g10['job_fid']]
    sc = dict(freqs=freqs, projnm=str(tmp_path / 'p'))
    omega.UtoutWriter(sc)(g10['job_data'])
    assert np.array_equal(np.frombuffer(open(str(tmp_path / 'p.utout'), 'rb').read(), dtype=np.uint8), g10['job_utout'])
    omega.UtoutWriter(dict(sc, tau=0.4))(g10['job_data'], ftype='utdamp')
    assert np.array_equal(np.frombuffer(open(str(tmp_path / 'p.utdamp'), 'rb').read(), dtype=np.uint8), g10['job_utout_tau'])
    om, data = omega.utoutRead(str(tmp_path / 'p.utdamp'), g10['job_data'].shape[0])
    assert np.allclose(om, 2 * np.pi * np.array(freqs) + 1j / 0.4, rtol=1e-6)
    assert np.array_equal(data, g10['job_data'].astype(np.complex64))
    with pytest.raises(Exception):
        omega.UtoutWriter(sc)(g10['job_data'][:, :, 0])


def _restricted(g10):
    fid = [int(i) for i in g10['job_fid']]
    step = int(g10['job_src_step'])
    return dict(freqs=[float(g10['sc_freqs'][i]) for i in fid],
                geom=dict(src=g10['sc_src'][::step], rec=g10['sc_rec'], mode='fixed'))


def test_omega_job_pipeline_host_logic(project, g10):
    'the whole OmegaJob chain with the arithmetic of the discretisation done by the CPU double'
    j = jobs.OmegaJob(project, dict(_restricted(g10), Disc=OracleMiniZephyrHD, parallel=False), verbose=False)
    assert [c.__name__ for c in type(j).__mro__[:3]] == ['OmegaJob', 'IsotropicVisco2DJob', 'Visco2DJob']
    data = j.run()
    ref = g10['job_data']
    assert data.shape == ref.shape
    assert np.linalg.norm(data - ref) / np.linalg.norm(ref) <= 1e-9
    om, got = omega.utoutRead(project + '.utout', ref.shape[0])
    assert np.allclose(got, ref.astype(np.complex64), rtol=1e-5, atol=1e-5 * np.abs(ref).max())


def test_flat_and_pickle_datastores(tmp_path):
    import pickle
    (tmp_path / 'p.py').write_text('systemConfig = {"nx": 3, "nz": 4}\n')
    assert omega.FlatDatastore(str(tmp_path / 'p')).systemConfig == {'nx': 3, 'nz': 4}
    with open(str(tmp_path / 'q.pickle'), 'wb') as fp:
        pickle.dump({'nx': 5}, fp)
    assert omega.PickleDatastore(str(tmp_path / 'q')).systemConfig == {'nx': 5}


def test_cli_dispatch(capsys):
    assert cli.main(['pack', 'foo']) == 0
    assert 'foo' in capsys.readouterr().out
    assert cli.main(['model', 'foo', '--job', 'NoSuchJob']) == 2
    assert cli.main([]) == 2


def test_jvec_matches_reference(g10):
    g6 = np.load(os.path.join(GOLD, 'g6_survey.npz'))
    sc = dict(nx=80, nz=60, dx=10., dz=10., c=g6['c'], rho=g6['rho'], nPML=6, freqs=[6., 9., 14.], Disc=OracleMiniZephyrHD, parallel=False,
              sterms=g6['sterms'], geom=dict(src=g6['src'], rec=g6['rec'], mode='fixed'))
    prob, surv = Helm2DProblem(sc), Helm2DSurvey(sc)
    prob.pair(surv)
    d = prob.Jvec(None, g10['jvec_v'])
    assert np.linalg.norm(d - g10['jvec']) / np.linalg.norm(g10['jvec']) <= 1e-9
    with pytest.raises(Exception):
        prob.Jvec(None, None)


@pytest.mark.gpu
def test_omega_job_on_gpu(project, g10):
    'xhlayr project end to end on the device: ini + SEG-Y in, MiniZephyrHD on the GPU, .utout out'
    j = jobs.OmegaJob(project, _restricted(g10), verbose=False)
    data = j.run()
    ref = g10['job_data']
    assert np.linalg.norm(data - ref) / np.linalg.norm(ref) <= 1e-7
    om, got = omega.utoutRead(project + '.utout', ref.shape[0])
    assert np.allclose(got, ref.astype(np.complex64), rtol=1e-4, atol=1e-5 * np.abs(ref).max())
