"""Test infrastructure: sparse-LU (oracle) side of the full-size parity tests, run in spawned worker processes so that
several frequencies are factored at once on the host cores while the GPU results wait in /dev/shm.

A job is a plain dict (picklable):
    n, dx, nPML, cPML      grid
    model                  ('marmousi', passes)  -> box_smooth(marmousi_like(n, n, dx), passes) (passes = 0: as generated)
    freq                   Hz
    system                 'eurus_m1'  LU of the M1 block (the isotropic Eurus system is block-triangular and its
                                       second-field right-hand side is zero: identical result)
                           'eurus_2n'  the faithful 2N x 2N system the reference factors (eurus.py:430-464,512-533)
                           'minizephyr'
    src                    (nsrc, 2) source locations (x, z); sources are SparseKaiserSource columns
    tti                    optional True: smooth tilted-transverse-isotropy fields (tti_fields) -> the coupled system, eps != delta
    rhsfile                optional .npy with dense right-hand-side columns (N, k) that REPLACE the Kaiser sources of `src` (dense-rhs parity)
    ufile                  .npy with the GPU wavefields (N, ncols), or None
    rec                    optional (nrec, 2): also return the projected data R u_lu
    resid                  optional (nrec, nsrc): back-propagate it and return this frequency's gradient term
                           -(w^2 / c^3) sum_s uF (.) uB   (problem.py:74-81,124-164)
Returns dict(err=[per-column rel-L2 of the GPU field vs LU], data=..., grad=..., seconds=dict(assemble, factor, solve)).
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def model_of(job):
    from zephyr_amd.models import marmousi_like, box_smooth
    kind, passes = job['model']
    assert kind == 'marmousi'
    c = marmousi_like(job['n'], job['n'], job['dx'])
    return box_smooth(c, passes) if passes else c


def tti_fields(n):
    """Smooth theta / eps / delta with eps != delta everywhere (the coupled two-field Eurus system, eurus.py:279-295)."""
    zz, xx = np.mgrid[0:n, 0:n] / float(n)
    return dict(theta=0.3 * np.sin(2 * np.pi * xx) * np.cos(np.pi * zz), eps=0.15 + 0.1 * np.sin(3 * np.pi * zz), delta=0.05 + 0.05 * np.cos(2 * np.pi * xx))


def lu_job(job):
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(limits=1)
    except Exception:
        pass
    import scipy.sparse as sp
    from oracle import helm_oracle as ho
    from zephyr_amd.source import SparseKaiserSource
    n, dx, f = job['n'], job['dx'], float(job['freq'])
    nPML, cPML = job.get('nPML', 10), job.get('cPML', 1e3)
    c = model_of(job)
    cfg = dict(nx=n, nz=n, dx=dx, dz=dx, c=c, nPML=nPML)
    t0 = time.perf_counter()
    rho = ho.gardner_rho(c)
    if job['system'] == 'minizephyr':
        C = ho.minizephyr_coefficients(n, n, c, rho, f, dx=dx, dz=dx, nPML=nPML)
        op = ho.DirectOperator(C)
    else:
        aniso = tti_fields(n) if job.get('tti') else {}
        C4 = ho.eurus_coefficients(n, n, c, rho, f, dx=dx, dz=dx, nPML=nPML, cPML=cPML, **aniso)
        op = ho.DirectOperator(C4, eurus=True) if job['system'] == 'eurus_2n' else ho.DirectOperator(C4[0])
    t1 = time.perf_counter()
    op.factor()
    t2 = time.perf_counter()
    S = SparseKaiserSource(cfg)
    q = np.load(job['rhsfile']) if job.get('rhsfile') else S(np.asarray(job['src']))
    nsrc = q.shape[1]
    out = dict(freq=f)
    rhs = q
    R = None
    if job.get('rec') is not None:
        R = sp.csr_matrix(S(np.asarray(job['rec'])).T)
    if job.get('resid') is not None:
        qb = sp.csc_matrix(R.T @ sp.csc_matrix(np.asarray(job['resid'])))      # (N, nsrc) back-sources (survey.py:171-188)
        rhs = sp.hstack((q, qb))
    u = op * rhs
    t3 = time.perf_counter()
    out['seconds'] = dict(assemble=t1 - t0, factor=t2 - t1, solve=t3 - t2, columns=rhs.shape[1])
    if job.get('ufile'):
        ug = np.load(job['ufile'], mmap_mode='r')
        k = min(ug.shape[1], u.shape[1])
        out['err'] = [float(np.linalg.norm(ug[:, j] - u[:, j]) / np.linalg.norm(u[:, j])) for j in range(k)]
    if R is not None:
        out['data'] = np.asarray(R @ u[:, :nsrc])
    if job.get('resid') is not None:
        omega = 2 * np.pi * f
        out['grad'] = (-(omega ** 2) / c.ravel().astype(complex) ** 3) * (u[:, :nsrc] * u[:, nsrc:]).sum(axis=1)
    return out


def run_jobs(jobs, nproc=None):
    """Run LU jobs in spawned processes (no GPU state is inherited); results in job order."""
    import multiprocessing as mp
    if nproc is None:
        nproc = max(1, min(len(jobs), (os.cpu_count() or 2) // 2, 8))
    keys = ('OMP_NUM_THREADS', 'OPENBLAS_NUM_THREADS', 'MKL_NUM_THREADS')
    saved = {k: os.environ.get(k) for k in keys}
    for k in keys:
        os.environ[k] = '1'
    try:
        if nproc == 1:
            return [lu_job(j) for j in jobs]
        with mp.get_context('spawn').Pool(nproc) as pool:
            return pool.map_async(lu_job, jobs, chunksize=1).get(timeout=3000)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
