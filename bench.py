#!/usr/bin/env python3
"""bench.py -- wavefields/s of the Helmholtz hot path on MI355X (BASELINE.json metric).

Workload (BASELINE.json configs[2] / SURVEY.md 8(d) cfg3): Eurus, isotropic, 1024 x 1024 synthetic
Marmousi-scale model (seed 20240512), dx = dz = 9 m, 16 frequencies linspace(2, 9.5, 16) Hz, 256
Kaiser-windowed-sinc sources at z = 20 m, fp64 complex.  One "step" = one work item of that job:
assemble A(w) for one frequency on the GPU, factor it and solve it for all 256 sources to
a true relative residual <= 1e-10, right-hand sides and wavefields resident in HBM.  Work items are
dealt round-robin over ranks (weak scaling: every rank does `steps` items; no data-path collective).

Prints ONE JSON line (rank 0).  `value` = wavefields completed by all ranks / max-over-ranks time.
Extra passes over the same K items, reported beside it (never as `value`): `unprofiled` (per-launch events off), `every_front_computed` (nothing skipped
on the point sources), `support_declared` (the sparse source matrix's support handed to the solver), `strong_scaling_job` (the 4096-wavefield job once).
"""
import argparse
import json
import os
import sys
import time

# BLAS / OpenMP pools of ONE thread for this process, set before numpy loads its BLAS: the job's host side needs four or five threads (main, prepare, solve, the
# runtime's own), and a threaded BLAS call anywhere near a timed region leaves 64 OpenBLAS workers spinning for ~100 ms -- under the GPU boxes' container CPU
# quota (cgroup cpu.max = 16 CPUs per 100 ms, while os.cpu_count() says 256) that exhausts the period's budget in 25 ms and the scheduler freezes every thread of
# the process, the ones feeding the GPU included, for the remaining 75 ms (round 6: found in config 4, whose updateModel compared two models with np.linalg.norm).
# The CPU-baseline legs are single-thread by definition (`cores`: 1) or run their own processes (cpu_baseline_pool).
for _k in ('OPENBLAS_NUM_THREADS', 'OMP_NUM_THREADS', 'MKL_NUM_THREADS'):
    os.environ.setdefault(_k, '1')
# the library's pool tops its spares up by itself when a device's last operator is destroyed -- which is how every pass of this file ENDS, inside its timing
# window; here the top-ups are asked for between the passes instead (barrier(), helm_pool_spares)
os.environ.setdefault('HELM_POOL_SPARE_AUTO', '0')
# items the device pipeline's prepare thread may be ahead of its solve thread (profiles/r06_lookahead.txt).  Two absorbed the prepare thread's 10-20 ms waits
# behind another operator's solve while those existed (1 -> 14 747 - 15 087 wavefields/s, 2 -> 15 356 - 15 545); since the set-up kernels run on a hardware
# queue of their own, 1, 2 and 3 give the same 15 500 - 15 800.  Two stays: slack against a late hand-over costs one more operator alive (three pool spares).
LOOKAHEAD = int(os.environ.get('HELM_BENCH_LOOKAHEAD', '2'))

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

NFREQ, NSRC = 16, 256
HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable
F64_PEAK_TFLOPS = 78.6    # MI355X fp64 dense peak, vector FMA = MFMA f64 rate (AMD spec)


def cgroup_cpu():
    """CPU quota (cores) and throttling counters of this process's cgroup: (quota_cores or None, throttled periods, throttled microseconds).  cgroup v2 (cpu.max,
    cpu.stat) or v1 (cpu.cfs_quota_us, cpu.stat); zeros where unreadable."""
    quota, nthr, usec = None, 0, 0
    try:
        q, per = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        quota = None if q == 'max' else float(q) / float(per)
        stat = open('/sys/fs/cgroup/cpu.stat').read()
    except Exception:
        try:
            q = float(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read()); per = float(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            quota = None if q <= 0 else q / per
            stat = open('/sys/fs/cgroup/cpu/cpu.stat').read()
        except Exception:
            stat = ''
    for line in stat.splitlines():
        k, _, v = line.partition(' ')
        if k == 'nr_throttled':
            nthr = int(v)
        elif k == 'throttled_usec':
            usec = int(v)
        elif k == 'throttled_time':          # v1: nanoseconds
            usec = int(v) // 1000
    return quota, nthr, usec


def memtrace(what):
    'HELM_BENCH_MEMTRACE=1: free device memory (hipMemGetInfo) at the phases of the run, on stderr'
    if os.environ.get('HELM_BENCH_MEMTRACE'):
        import torch
        free, tot = torch.cuda.mem_get_info()
        sys.stderr.write('[bench mem] %-34s free %6.1f of %6.1f GB\n' % (what, free / 1e9, tot / 1e9))
        sys.stderr.flush()


def build_config(n, dx):
    from zephyr_amd.models import marmousi_like
    # complex128 once, here: the discretisation classes cast `c` to complex128 at construction (discretization.py:23-31), which for a
    # float array is a 16 MB conversion per operator -- a job builds one operator per frequency from the same model
    c = marmousi_like(n, n, dx).astype(np.complex128)
    return dict(nx=n, nz=n, dx=dx, dz=dx, c=c, nPML=10, cPML=1e3, freeSurf=(False, False, False, False))


def source_locations(n, dx, nsrc):
    xs = np.linspace(0.04 * n * dx, 0.96 * n * dx, nsrc)
    return np.stack([xs, np.full(nsrc, 20.0)], axis=1)


def work_item(w, nbatch_per_freq):
    """global work-item index -> (frequency index, source-batch index).  Consecutive items walk the
    frequency list with stride 5 (co-prime with 16) so short runs sample low and high frequencies."""
    return (w * 5) % NFREQ, (w // NFREQ) % nbatch_per_freq


def cpu_baseline(cfg, freqs, q_host, sample_rhs=8):
    """Reference-equivalent CPU path (oracle: numpy assembly + SciPy SuperLU + back-substitution) on a
    bounded sample: ONE frequency of the same 1024^2 model, LU of the isotropic-equivalent M1 block
    (identical result to the reference's 2N system), `sample_rhs` back-substitutions.
    Throughput is quoted for the job's 256 sources per frequency: 256 / (assemble + factor + 256*t_rhs)."""
    from oracle import helm_oracle as ho
    try:
        from threadpoolctl import threadpool_limits
        limiter = threadpool_limits(limits=1)          # one core: SuperLU's BLAS must not fan out over the host
    except Exception:
        limiter = None
    n = cfg['nx']
    f = float(freqs[len(freqs) // 2])
    t0 = time.perf_counter()
    c_real = np.ascontiguousarray(cfg['c'].real)
    rho = ho.gardner_rho(c_real)
    C4 = ho.eurus_coefficients(n, n, c_real, rho, f, dx=cfg['dx'], dz=cfg['dz'], nPML=cfg['nPML'], cPML=cfg['cPML'])
    op = ho.DirectOperator(C4[0])
    t1 = time.perf_counter()
    op.factor()
    t2 = time.perf_counter()
    u = op * q_host[:, :sample_rhs]
    t3 = time.perf_counter()
    t_rhs = (t3 - t2) / sample_rhs
    wps = NSRC / ((t1 - t0) + (t2 - t1) + NSRC * t_rhs)
    return dict(value=wps, unit='wavefields/s', cores=1, kind='port',
                system='M1-only (isotropic-equivalent N x N block; the reference factors the 2N x 2N system, see cpu_baseline_2n)',
                freq_hz=f, assemble_s=t1 - t0, factor_s=t2 - t1, per_rhs_s=t_rhs, sample_rhs=sample_rhs,
                sample='1 of 16 freqs (%.2f Hz) of the 1024^2 job: numpy assembly %.1fs + SciPy SuperLU factor %.1fs + %d back-substitutions %.3fs each, '
                       'extrapolated to 256 sources/frequency; single thread; host has %d logical CPUs, container CPU quota %s'
                       % (f, t1 - t0, t2 - t1, sample_rhs, t_rhs, os.cpu_count(), cgroup_cpu()[0])), u


def cpu_baseline_2n(n=512, dx=10.0, f=6.0, sample_rhs=4):
    """The FAITHFUL reference system: Eurus' 2N x 2N block matrix [[M1, M2], [M3, M4]] handed to the sparse LU
    (eurus.py:430-464,512-533), on the 512^2 configuration (config 2: ~11 GB of factors; the 1024^2 one needs ~45 GB and
    minutes).  Quoted per 64 sources per frequency, config 2's batch."""
    from oracle import helm_oracle as ho
    from zephyr_amd.models import marmousi_like
    try:
        from threadpoolctl import threadpool_limits
        limiter = threadpool_limits(limits=1)
    except Exception:
        limiter = None
    c = marmousi_like(n, n, dx)
    t0 = time.perf_counter()
    C4 = ho.eurus_coefficients(n, n, c, ho.gardner_rho(c), f, dx=dx, dz=dx, nPML=10, cPML=1e3)
    op = ho.DirectOperator(C4, eurus=True)
    t1 = time.perf_counter()
    op.factor()
    t2 = time.perf_counter()
    q = np.zeros((n * n, sample_rhs), complex)
    q[2 * n + n // 2 + np.arange(sample_rhs) * 7, np.arange(sample_rhs)] = 1.
    op * q
    t3 = time.perf_counter()
    t_rhs = (t3 - t2) / sample_rhs
    nsrc = 64
    return dict(value=nsrc / ((t2 - t0) + nsrc * t_rhs), unit='wavefields/s', cores=1, kind='port',
                system='2N x 2N (faithful: the block system the reference factors, eurus.py:430-464)',
                grid=[n, n], freq_hz=f, assemble_s=t1 - t0, factor_s=t2 - t1, per_rhs_s=t_rhs, sample_rhs=sample_rhs,
                sample='1 frequency (%.1f Hz) of the %d^2 config-2 model: assemble %.1fs + SuperLU of the 2N system %.1fs + %d back-substitutions %.3fs each, '
                       'extrapolated to 64 sources/frequency; single thread' % (f, n, t1 - t0, t2 - t1, sample_rhs, t_rhs))


def _pool_worker(job):
    '''one frequency of the CPU path in its own process (spawned: no GPU state is inherited)'''
    n, dx, f, nsolve = job
    import numpy as _np
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(limits=1)
    except Exception:
        pass
    from oracle import helm_oracle as ho
    from zephyr_amd.models import marmousi_like
    t0 = time.perf_counter()
    c = marmousi_like(n, n, dx)
    C4 = ho.eurus_coefficients(n, n, c, ho.gardner_rho(c), f, dx=dx, dz=dx, nPML=10, cPML=1e3)
    op = ho.DirectOperator(C4[0])
    op.factor()
    t1 = time.perf_counter()
    q = _np.zeros((n * n, nsolve), complex)
    q[2 * n + n // 2 + _np.arange(nsolve) * 7, _np.arange(nsolve)] = 1.
    op * q
    t2 = time.perf_counter()
    return (t1 - t0, (t2 - t1) / nsolve)


def cpu_baseline_pool(cfg, freqs, nproc=16, nsolve=4):
    '''The reference's MultiFreq pool mode (distributors.py:92-96,161-168): one process per frequency, all 16
    frequencies of the job factored concurrently, single-thread SuperLU each.'''
    import multiprocessing as mp
    ctx = mp.get_context('spawn')
    jobs = [(cfg['nx'], cfg['dx'], float(f), nsolve) for f in freqs[:nproc]]
    # the children must come up with single-thread BLAS: limiting after scipy's OpenBLAS has spun up its
    # threads in 16 processes at once oversubscribes the host badly enough to look like a hang
    keys = ('OMP_NUM_THREADS', 'OPENBLAS_NUM_THREADS', 'MKL_NUM_THREADS')
    saved = {k: os.environ.get(k) for k in keys}
    for k in keys:
        os.environ[k] = '1'
    t0 = time.perf_counter()
    try:
        with ctx.Pool(len(jobs)) as pool:
            res = pool.map_async(_pool_worker, jobs).get(timeout=600)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    wall = time.perf_counter() - t0
    setup = max(r[0] for r in res); per_rhs = max(r[1] for r in res)
    wps = len(jobs) * NSRC / (setup + NSRC * per_rhs)
    return dict(value=wps, unit='wavefields/s', cores=len(jobs), kind='port', system='M1-only (isotropic-equivalent)',
                slowest_assemble_plus_factor_s=setup, slowest_per_rhs_s=per_rhs,
                sample='%d frequencies of the %d^2 job in %d concurrent single-thread processes (the reference MultiFreq pool mode): slowest assemble+factor %.1fs, '
                       'slowest back-substitution %.3fs per source (%d measured), extrapolated to 256 sources per frequency; wall of the sample %.1fs'
                       % (len(jobs), cfg['nx'], len(jobs), setup, per_rhs, nsolve, wall))


def config5_leg(local, rtol=1e-8, nsrc=16, freqs=(2., 3., 4., 5.), grid=(128, 256, 256)):
    """BASELINE configs[4]: the 3-D 27-point operator on 256 x 256 x 128 (nz = 128), homogeneous c = 2000 m/s, rho = 1, h = 10 m, nPML = 10,
    4 frequencies x 16 point sources, right-hand sides and wavefields resident in HBM; one GPU does the whole job here (under N ranks the
    dispatcher deals (frequency, source-batch) items).  Per frequency: seconds including the preconditioner set-up, seconds of a second
    solve that re-uses it (their difference = the set-up), BiCGSTAB iterations per source; plus the 27-point apply against the HBM roofline."""
    import ctypes
    import torch
    from zephyr_amd import Helm3D, _lib
    lib = _lib.load()
    lib.helm_trim()                                    # the 2-D leg's scratch (tens of GB) goes back first
    nz, ny, nx = grid
    N = nx * ny * nz
    dev = torch.device('cuda', local)
    cfg = dict(nx=nx, ny=ny, nz=nz, dx=10., c=2000., rho=1., freq=freqs[0], nPML=10, rtol=rtol, maxit=60000, batch=nsrc, method='auto', device=local)
    out = {'workload': '3D 27-pt Helmholtz %dx%dx%d (nx, ny, nz) homogeneous c=2000 m/s, rho=1, h=10 m, nPML=10; %d freqs x %d sources, rtol %g, fp64; '
                       'BiCGSTAB right-preconditioned by the layer-preserving 3-D multigrid with a direct coarse solve (column dissection: the 2-D multifrontal solver over z-columns; plane-by-plane elimination where that is cheaper)'
                       % (nx, ny, nz, len(freqs), nsrc, rtol),
           'grid_nz_ny_nx': [nz, ny, nx], 'rtol': rtol, 'per_frequency': [], 'apply': []}
    q = np.zeros((nsrc, N), complex)
    for s_ in range(nsrc):
        q[s_, ((20 + 5 * s_) * ny + ny // 2) * nx + nx // 4 + (30 * s_) % (nx // 2)] = 1.     # all inside the physical domain
    Q = torch.from_numpy(q).to(dev)
    U = torch.empty_like(Q)
    from zephyr_amd import dispatch

    marks = []                           # (what, frequency, seconds since the job started) of the last pipelined job

    def prep(f):
        marks.append(('prepare starts', float(f), time.perf_counter()))
        o = Helm3D(dict(cfg, freq=float(f)))
        o.prefactor(nsrc)
        marks.append(('prepare done', float(f), time.perf_counter()))
        return o

    def solve(o):
        marks.append(('solve starts', float(complex(o.freq).real), time.perf_counter()))
        try:
            o.solveDevice(Q.data_ptr(), U.data_ptr(), nsrc)
            st = 'ok'
        except ArithmeticError as e:
            st = str(e)
        its_ = [i['iterations'] for i in o.lastInfo]
        del o.factors
        marks.append(('solve done', float(complex(o.freq).real), time.perf_counter()))
        return st, its_

    def pipelined_job():
        return list(dispatch.pipelined([dispatch.WorkItem(solve, (lambda f=f: prep(f))) for f in freqs], device=local,
                                       lookahead=int(os.environ.get('HELM_C5_LOOKAHEAD', '1')), strict=os.environ.get('HELM_C5_STRICT', '1') != '0'))
    # untimed warm-up (the W of this leg): the job once through the pipeline brings the Krylov workspaces (23 GB per operator in flight), the factor
    # storage of the directly solved levels and the once-per-process calibrations of the depth model into being; everything goes back to the
    # library's pools before the clock starts
    pipelined_job()
    torch.cuda.synchronize()
    total = 0.0
    for f in freqs:
        cfg['freq'] = float(f)
        op = Helm3D(cfg)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        status = 'ok'
        try:
            op.solveDevice(Q.data_ptr(), U.data_ptr(), nsrc)       # builds the operator and the preconditioner of this frequency, then solves
        except ArithmeticError as e:
            status = str(e)
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        its = [i['iterations'] for i in op.lastInfo]
        worst = max(i['relres'] for i in op.lastInfo)
        t0 = time.perf_counter()
        try:
            op.solveDevice(Q.data_ptr(), U.data_ptr(), nsrc)       # the same solves with the preconditioner already built
        except ArithmeticError:
            pass
        torch.cuda.synchronize()
        t_re = time.perf_counter() - t0
        total += t_all
        out['per_frequency'].append({'freq_hz': float(f), 'seconds': t_all, 'seconds_reusing_setup': t_re, 'setup_seconds': max(0.0, t_all - t_re),
                                     'setup_share': max(0.0, t_all - t_re) / t_all, 'iterations': its, 'worst_relres': worst, 'status': status})
        del op.factors
    out['job_seconds_one_after_the_other'] = total
    # the job as the dispatcher runs it (zephyr_amd.dispatch, what MultiFreq's parallel mode uses): the prepare thread builds the operator AND the
    # preconditioner of frequency k+1 (helm_prefactor_n: multigrid hierarchy + factorisation of the directly solved level) while frequency k iterates
    torch.cuda.synchronize()
    del marks[:]
    t0 = time.perf_counter()
    res = pipelined_job()
    torch.cuda.synchronize()
    tp = time.perf_counter() - t0
    out['job_seconds'] = tp
    out['job_seconds_what'] = ('4 frequencies x %d sources through the device pipeline: set-up of frequency k+1 beside the iterations of frequency k; '
                               'job_seconds_one_after_the_other = the sum of the per-frequency seconds above' % nsrc)
    out['pipelined'] = [{'freq_hz': float(f), 'status': r[0], 'iterations': r[1]} for f, r in zip(freqs, res)]
    out['pipelined_timeline_ms'] = [(w, f, round(1e3 * (t - t0), 1)) for w, f, t in sorted(marks, key=lambda m: m[2])]
    out['wavefields_per_s'] = len(freqs) * nsrc / tp
    if rtol > 1e-10:
        # the same job at the tolerance the 2-D path is held to (solver stop at ||r|| / ||q|| <= 1e-10)
        cfg['rtol'] = 1e-10
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res10 = pipelined_job()
        torch.cuda.synchronize()
        out['job_seconds_rtol1e10'] = time.perf_counter() - t0
        out['pipelined_rtol1e10'] = [{'freq_hz': float(f), 'status': r[0], 'iterations': r[1]} for f, r in zip(freqs, res10)]
        cfg['rtol'] = rtol
    del Q, U
    # 27-point apply (k_stencil3): algorithmic bytes N*(32*B + 432), HIP events on the solver stream
    op = Helm3D(cfg)
    op.setProfiling(True)
    for Bm in (1, 4, 8, 16):
        X = torch.randn((Bm, N), dtype=torch.complex128, device=dev)
        Y = torch.empty_like(X)
        torch.cuda.synchronize()
        ms = by = 0.0
        for rep in range(6):
            _lib.check(lib.helm_apply_device(op.handle, 0, 0, ctypes.c_void_p(X.data_ptr()), ctypes.c_void_p(Y.data_ptr()), Bm), op.handle)
            tm = op.lastTiming()
            if rep:
                ms += tm['apply_ms']; by += tm['apply_bytes']
        out['apply'].append({'B': Bm, 'us': 1e3 * ms / 5, 'GBps': by / (ms * 1e-3) / 1e9, 'frac_of_peak': by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS})
        del X, Y
    del op.factors
    lib.helm_trim()
    return out


def config2_leg(local, rtol=1e-10, method='auto'):
    """BASELINE configs[1]: Eurus isotropic 512 x 512 synthetic Marmousi slice (dx = dz = 10 m), 8 frequencies linspace(3, 10, 8) Hz x 64
    Kaiser-windowed sources at z = 20 m, fp64, one GPU.  Three figures: the job with right-hand sides and wavefields resident in HBM through the device pipeline
    (the headline's call shape at config 2's sizes), the same job through the reference's call shape `MultiFreq(cfg) * q` (scipy-sparse in, numpy out over PCIe),
    and the products' roofline fraction in a serial pass with nothing skipped."""
    import torch
    from zephyr_amd import Eurus, MultiFreq, SparseKaiserSource, dispatch, _lib
    _lib.load().helm_trim()
    n, dx, nf, ns = 512, 10.0, 8, 64
    cfg = build_config(n, dx)
    freqs = np.linspace(3.0, 10.0, nf)
    locs = np.stack([np.linspace(200.0, 4920.0, ns), np.full(ns, 20.0)], axis=1)
    qs = SparseKaiserSource(cfg)(locs)
    N = n * n
    dev = torch.device('cuda', local)
    d_rhs = torch.from_numpy(np.ascontiguousarray(qs.toarray())).to(dev)
    d_u = torch.empty((N, ns), dtype=torch.complex128, device=dev)

    def prep(f, profile):
        op = Eurus(dict(cfg, freq=float(f), rtol=rtol, maxit=400000, method=method, batch=ns, device=local))
        op.setProfiling(profile)
        if not profile:
            op.prefactor()
        return op

    def solve(op):
        info = op.solveDevice(d_rhs.data_ptr(), d_u.data_ptr(), ns, N, layout='node')
        t = op.lastTiming()
        del op.factors
        return info, t

    def job(profile=False):
        items = [dispatch.WorkItem(solve, (lambda f=f: prep(f, profile))) for f in freqs]
        if profile:
            return [solve(prep(f, True)) for f in freqs]
        return list(dispatch.pipelined(items, device=local, lookahead=1))
    job()                                                     # untimed: pools, plan, pinned records
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 3
    res = None
    for _ in range(reps):
        res = job()
    torch.cuda.synchronize()
    tj = (time.perf_counter() - t0) / reps
    worst = max(i['relres'] for info, _ in res for i in info)
    passes = max(i['iterations'] for info, _ in res for i in info)
    os.environ['HELM_ND_SPARSE_RHS'] = '0'
    try:
        prof = job(profile=True)
    finally:
        os.environ.pop('HELM_ND_SPARSE_RHS', None)
    gms = sum(t['gemm_ms'] for _, t in prof); gfl = sum(t['gemm_flops'] for _, t in prof)
    out = {'workload': 'Eurus 2D isotropic 512x512 synthetic-Marmousi slice (dx=10 m), 8 freqs 3-10 Hz x 64 Kaiser sources at z=20 m, fp64, 1 GPU',
           'wavefields': nf * ns, 'device_job_seconds': tj, 'device_wfs': nf * ns / tj, 'device_ms_per_item': 1e3 * tj / nf,
           'worst_relres': worst, 'max_passes': passes,
           'gemm_tflops_serial': gfl / (gms * 1e-3) / 1e12 if gms > 0 else None, 'gemm_frac_serial': gfl / (gms * 1e-3) / 1e12 / F64_PEAK_TFLOPS if gms > 0 else None}
    del d_rhs, d_u
    try:
        sch = dict(cfg, freqs=[float(f) for f in freqs], Disc=Eurus, rtol=rtol, maxit=400000, method=method, batch=ns, device=local)
        mfw = MultiFreq(dict(sch))
        for u in mfw * qs:
            del u
        del mfw.factors
        ths = []
        for _ in range(3):                                    # (median of three: the leg is 50-80 ms of PCIe copies and thread hand-overs, single runs scatter by 40 %)
            mf = MultiFreq(sch)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            chk = 0.0
            for u in mf * qs:
                chk += float(abs(u[N // 2 + 7, 0]))
                del u
            ths.append(time.perf_counter() - t0)
            del mf.factors
        th = float(np.median(ths))
        out['host_api_seconds'] = th
        out['host_api_seconds_all'] = ths
        out['host_api_wfs'] = nf * ns / th
    except Exception as exc:
        out['host_api_wfs'] = None
        out['host_api_error'] = str(exc)
    return out


def config4_leg(local, world, rank, dist, backend):
    """BASELINE configs[3]: one FWI gradient on the 512 x 512 model -- `dpred` of the 25-point-smoothed "current" model, the residual against
    the "true" model's data on 128 receivers at z = 20 m, `Jtvec` in the mux form (forward and back-propagated sources solved together, imaging condition on the
    device: zephyr/middleware/problem.py:124-164).  8 frequencies x 64 sources.  Under N ranks the frequencies are sharded over the ranks (zephyr_amd.parallel)
    and the gradient is summed with ONE all-reduce (problem.py:152,162); the seconds are the slowest rank's."""
    import torch
    import zephyr_amd as za
    from zephyr_amd import parallel, _lib
    from zephyr_amd.models import marmousi_like, box_smooth
    from zephyr_amd.problem import Helm2DProblem
    from zephyr_amd.survey import Helm2DSurvey
    _lib.load().helm_trim()
    n, dx, nf, ns, nr = 512, 10.0, 8, 64, 128
    ctrue = marmousi_like(n, n, dx)
    ccur = box_smooth(ctrue, 12)                              # 25-point box: 3-pt -> 25-pt smoothed "current" model
    freqs = list(np.linspace(3.0, 10.0, nf))
    src = np.stack([np.linspace(200.0, 4920.0, ns), np.full(ns, 20.0)], axis=1)
    rec = np.stack([np.linspace(100.0, dx * n - 100.0, nr), np.full(nr, 20.0)], axis=1)
    base = dict(nx=n, nz=n, dx=dx, dz=dx, freqs=freqs, Disc=za.Eurus, geom=dict(src=src, rec=rec, mode='fixed'), batch=ns, device=local)
    dev = torch.device('cuda', local)

    def pair(c):
        sc = dict(base, c=c)
        p, sv = Helm2DProblem(sc), Helm2DSurvey(sc)
        p.pair(sv)
        return p, sv

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    def slowest(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=dev if backend == 'nccl' else 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())
    # ONE problem / survey pair, the model handed in like an inversion does it (`dpred(m)`, `Jtvec(m, v)`: problem.py:51-66 updateModel): the survey's source and
    # receiver matrices are built once, each model gets its own operators and factorisations
    import gc
    gc.collect()                                               # (arrays of the legs before this one that only the collector can free go now, not inside a timed call:
                                                               #  unmapping host memory the runtime has registered stalls the process's queues)
    prob, surv = pair(ctrue)
    dobs = surv.dpred()                                        # (also the warm-up of the pools for the timed calls below)
    mcur, mtrue = ccur.ravel(), ctrue.ravel()
    dcur = surv.dpred(mcur)
    resid = dcur - dobs
    g0 = prob.Jtvec(mcur, resid)                               # (first call: the 2 x 64-column buffers come into being)

    per_op = []

    def device_ms():
        'sum over the model\'s eight operators of the device spans of their last factorisation and solve call (HIP events on their own streams)'
        tot = 0.0
        per_op.append([])
        for op in prob.system.subProblems:
            if op.factors:
                t = op.lastTiming()
                tot += t['solve_ms'] + t['factor_ms']
                per_op[-1].append((round(t['factor_ms'], 2), round(t['solve_ms'], 2)))
        return tot
    # five repetitions of each call, every one on a model that differs from the one before it (an inversion step: the call builds, assembles and factors its
    # eight operators); the figure quoted is the median, the spread (max - min) / median beside it
    reps = 5
    t_fwd, t_grad, gpu_fwd, gpu_grad, cg0 = [], [], [], [], cgroup_cpu()
    rt_fwd, rt_grad = [], []                                   # what each timed call made the HIP runtime and the interpreter's collector do (diagnostics)

    def rt_of(c0):
        r = _lib.runtime_stats()
        return {'dev_allocs': r['dev_allocs'], 'dev_alloc_ms': round(r['dev_alloc_ms'], 3), 'dev_frees': r['dev_frees'], 'pinned_allocs': r['host_allocs'],
                'first_launches': r['first_launches'], 'slow_syncs': r['slow_syncs'], 'worst_sync_ms': round(r['worst_sync_ms'], 3),
                'gc_collections': [a['collections'] - b for a, b in zip(gc.get_stats(), c0)]}
    g = g0
    for _ in range(reps):
        prob.updateModel(mtrue)
        sync(); _lib.runtime_stats(reset=True); c0 = [a['collections'] for a in gc.get_stats()]; t0 = time.perf_counter()
        surv.dpred(mcur)
        sync(); t_fwd.append(slowest(time.perf_counter() - t0)); gpu_fwd.append(device_ms()); rt_fwd.append(rt_of(c0))
        prob.updateModel(mtrue)
        sync(); _lib.runtime_stats(reset=True); c0 = [a['collections'] for a in gc.get_stats()]; t0 = time.perf_counter()
        g = prob.Jtvec(mcur, resid)
        sync(); t_grad.append(slowest(time.perf_counter() - t0)); gpu_grad.append(device_ms()); rt_grad.append(rt_of(c0))
    cg1 = cgroup_cpu()
    del prob.factors
    med = lambda v: float(np.median(v))
    spread = lambda v: float((max(v) - min(v)) / np.median(v))
    out = {'workload': 'FWI gradient step on 512x512 (true: synthetic Marmousi slice, current: its 25-pt box smooth), 8 freqs 3-10 Hz x 64 sources, 128 receivers at z=20 m; '
                       'dpred(m) + Jtvec(m, v) (mux form, device imaging), each including the construction, assembly and factorisation of the 8 operators of the model handed in; '
                       'frequencies sharded over %d rank(s), one all-reduce of the gradient; median of %d repetitions' % (world, reps),
           'dpred_seconds': med(t_fwd), 'jtvec_seconds': med(t_grad), 'dpred_seconds_all': t_fwd, 'jtvec_seconds_all': t_grad, 'dpred_runtime_all': rt_fwd, 'jtvec_runtime_all': rt_grad,
           'dpred_spread': spread(t_fwd), 'jtvec_spread': spread(t_grad),
           'gpu_ms_dpred_all': gpu_fwd, 'gpu_ms_jtvec_all': gpu_grad, 'gpu_ms_per_operator_factor_solve': per_op[-2 * reps:], 'gpu_ms_dpred': med(gpu_fwd), 'gpu_ms_jtvec': med(gpu_grad), 'gpu_ms': med(gpu_fwd) + med(gpu_grad),
           'gpu_ms_what': 'sum over the eight operators of a call of the device spans (HIP events) of factorisation and solve; the two run on different streams and overlap between operators',
           'cpu_throttled_ms': (cg1[2] - cg0[2]) / 1e3,
           'wavefields_per_s_forward': nf * ns / med(t_fwd), 'wavefields_per_s_gradient': 2 * nf * ns / med(t_grad),
           'gradient_norm': float(np.linalg.norm(g)), 'gradient_repeatable_rel': float(np.linalg.norm(g - g0) / max(np.linalg.norm(g), 1e-300)),
           'residual_norm': float(np.linalg.norm(resid))}
    # the collective on its own: N x 16 B complex128 at 512^2 and 1024^2, in place on the device (what _JtvecDevice issues once per gradient)
    for side in (512, 1024):
        key = 'gradient_allreduce_ms_%d' % side
        if world == 1:
            out[key] = 0.0
            continue
        G = torch.ones(side * side, dtype=torch.complex128, device=dev)
        parallel.allreduce_sum_device(G); sync()
        t0 = time.perf_counter()
        for _ in range(5):
            parallel.allreduce_sum_device(G)
        sync()
        out[key] = slowest(1e3 * (time.perf_counter() - t0) / 5)
    return out


def spawn_ranks(ngpus, argv):
    """`python bench.py --gpus N` without a launcher: start N rank processes (one per GPU, RCCL rendezvous on 127.0.0.1) BEFORE
    anything in this process touches the GPU, relay rank 0's JSON line, fail if any rank fails.  (The driver's
    `torch.distributed.run` launch sets WORLD_SIZE and never comes here.)"""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0))
        port = so.getsockname()[1]
    base = dict(os.environ, WORLD_SIZE=str(ngpus), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    procs = []
    for r in range(ngpus):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0 = procs[0].communicate()[0].decode()
    codes = [p.wait() for p in procs]
    sys.stdout.write(out0)
    sys.stdout.flush()
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if bad:
        sys.stderr.write('bench.py: ranks failed (rank, exit code): %s\n' % bad)
        return 1
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=8)
    ap.add_argument('--warmup', type=int, default=4)       # (three operators are alive at a time in the pipeline: fewer warm-up items leave allocations of GBs -- 70-100 ms each beside running kernels -- to the timed region)
    ap.add_argument('--grid', '--n', dest='n', type=int, default=1024, help='grid side (1024 = BASELINE workload)')
    ap.add_argument('--dx', type=float, default=9.0)
    ap.add_argument('--batch', type=int, default=256, help='sources per work item (256 = all sources of a frequency)')
    ap.add_argument('--rtol', type=float, default=1e-10)
    ap.add_argument('--no-cpu', action='store_true', help='skip the CPU baseline leg')
    ap.add_argument('--cpu-pool', action='store_true', help='(default now) kept for compatibility')
    ap.add_argument('--no-cpu-pool', action='store_true', help='skip the 16-process CPU pool baseline (it adds ~35 s)')
    ap.add_argument('--method', default='auto')
    ap.add_argument('--no-plain-pass', action='store_true', help='skip the extra un-profiled pass over the same work items')
    ap.add_argument('--no-roofline-pass', action='store_true', help='skip the separate serial pass the kernel-level roofline is measured in (profile collections that want only the timed region in the trace)')
    ap.add_argument('--no-cpu-2n', action='store_true', help='skip the faithful 2N x 2N Eurus LU baseline at 512^2 (~1 min, ~11 GB)')
    ap.add_argument('--layout', choices=('node', 'rhs'), default='node',
                    help="device buffers of the timed region: 'node' = the reference's (N, nsrc) C-order arrays (default), 'rhs' = one right-hand side per row")
    ap.add_argument('--no-host-api', action='store_true', help='skip the host-array leg (MultiFreq * q with numpy / scipy-sparse in, numpy out; ~5 s)')
    ap.add_argument('--no-pipeline', dest='pipeline', action='store_false',
                    help='work items strictly one after the other (no prepare-ahead thread, no helm_prefactor)')
    ap.add_argument('--scaling', choices=('weak', 'strong'), default='weak',
                    help="weak (default, the driver's contract): every rank does --steps work items; strong: the 16-frequency job itself (16 items of 256 sources) "
                         "split over the ranks, frequency-major, one number for the whole job")
    ap.add_argument('--no-config5', action='store_true', help='skip the 3-D leg (BASELINE configs[4]: 256x256x128, 4 freqs x 16 sources; ~15 s)')
    ap.add_argument('--config5-rtol', type=float, default=1e-8)
    ap.add_argument('--no-config2', action='store_true', help='skip the config-2 leg (512^2, 8 freqs x 64 sources; ~10 s)')
    ap.add_argument('--no-config4', action='store_true', help='skip the config-4 leg (FWI gradient step at 512^2; ~15 s)')
    ap.add_argument('--group', type=int, default=int(os.environ.get('HELM_BENCH_GROUP', '2')),
                    help='work items whose factorisations the device pipeline enqueues together (helm_prefactor_many: the fronts of `group` frequencies in the same '
                         'batched launches; 1: every frequency factored by itself, rounds 1-5)')
    ap.add_argument('--streams', type=int, default=1, help='work items in flight per GPU (host threads, one operator handle / HIP stream each)')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))

    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

    # N rank processes on one node share the container's CPU quota: a rank of this job keeps ~2.5 CPUs busy (two threads that wait for the GPU in the runtime's
    # spinning wait).  Where the quota does not cover that, the library's waits sleep between polls instead (helm_tuning.sync_sleep_us): exhausting the quota would
    # freeze every thread of every rank for the rest of each 100-ms scheduler period.
    quota_cores = cgroup_cpu()[0]
    if quota_cores is not None and quota_cores / max(1, world) < 3.0 and 'HELM_SYNC_SLEEP_US' not in os.environ:
        os.environ['HELM_SYNC_SLEEP_US'] = '40'
    import torch
    import torch.distributed as dist
    backend = os.environ.get('HELM_BENCH_BACKEND', 'nccl')      # 'gloo' only for single-GPU dry runs of the N>1 logic
    ndev = max(1, torch.cuda.device_count())               # (counting devices does not initialise the GPU)
    if world > 1 and backend == 'nccl' and int(os.environ.get('LOCAL_WORLD_SIZE', world)) > ndev:
        # more ranks than GPUs on this node (a dry run of the N > 1 logic on a one-GPU box): RCCL refuses two ranks on one device
        sys.stderr.write('[bench] %d ranks on %d GPU(s): collectives through gloo (RCCL wants one device per rank)\n' % (world, ndev))
        backend = 'gloo'
    local = local % ndev
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        if backend == 'nccl':
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device('cuda', local))      # (the rank's own GPU, said explicitly)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)

    import __graft_entry__ as g
    if world > 1:                      # one rank checks / rebuilds the library, the others wait for it
        if rank == 0:
            g.build()
        dist.barrier()
    g.build()
    from zephyr_amd import Eurus, SparseKaiserSource

    n, dx = args.n, args.dx
    cfg = build_config(n, dx)
    freqs = np.linspace(2.0, 9.5, NFREQ)
    B = args.batch
    nb = NSRC // B
    src = SparseKaiserSource(cfg)
    locs = source_locations(n, dx, NSRC)
    N = n * n

    # all right-hand sides of the job are staged in HBM before the timed region
    q_sparse = src(locs)                                     # (N, 256) scipy-sparse: what the surveys hand to `system * q`
    q_all = q_sparse.toarray()                               # (N, 256) complex
    node = args.layout == 'node' and B == NSRC
    if node:            # the reference's own array layout: (N, nsrc) C-order in, (N, nsrc) C-order out (discretization.py:101-103)
        d_rhs = torch.from_numpy(np.ascontiguousarray(q_all)).to(dev)        # [N][256]
        d_u = torch.empty((N, B), dtype=torch.complex128, device=dev)
    else:               # one right-hand side per row (the C ABI's default layout)
        d_rhs = torch.from_numpy(np.ascontiguousarray(q_all.T)).to(dev)      # [256][N]
        d_u = torch.empty((B, N), dtype=torch.complex128, device=dev)

    # the support of the sparse source matrix, as Discretization.rhsSupportFromSparse makes it (one byte per cell, bit b = block b of 64 sources has an entry there):
    # used by the extra `support_declared` pass only -- the headline lets the library find the nonzeros of the dense right-hand sides itself
    use_support = [False]
    d_support = None
    if node:
        coo_ = q_sparse.tocoo()
        bits_ = np.zeros(((N + 3) // 4) * 4, np.uint8)
        np.bitwise_or.at(bits_, coo_.row, (1 << (coo_.col >> 6)).astype(np.uint8))
        d_support = torch.from_numpy(bits_).to(dev)

    ops = {}

    stamps = []                          # completion time of every work item (diagnostics: `item_done_ms` of the timed region)
    tl = []                              # (what, item, time) marks of the prepare / solve steps (`first_items_timeline_ms`)

    def prepare_item(w, profile):
        'create the operator of work item w, assemble it on the GPU (inside the timed region) and start its factorisation'
        fi, bi = work_item(w, nb)
        tl.append(('prepare starts', w, time.perf_counter()))
        sc = dict(cfg)
        sc.update(freq=float(freqs[fi]), rtol=args.rtol, maxit=400000, method=args.method, batch=B, device=local)
        op = Eurus(sc)
        op.setProfiling(profile and os.environ.get('HELM_BENCH_NOPROFILE', '0') != '1')
        if args.pipeline and args.group <= 1:
            op.prefactor()               # launches only: the factorisation runs beside the solves of the previous item
        # (--group G > 1: the dispatcher calls prefactor_many on G prepared operators at once, zephyr_amd.dispatch)
        tl.append(('prepare done', w, time.perf_counter()))
        return op

    nsolvers = max(1, int(os.environ.get('HELM_BENCH_SOLVERS', '1')))       # solve threads of the device pipeline (each writes its own wavefield array)
    u_bufs = {}
    import threading
    u_lock = threading.Lock()

    def solve_item(w, op, ubuf=None):
        if ubuf is None and nsolvers > 1:
            key = threading.get_ident()
            with u_lock:                 # (two solve threads must not both take d_u)
                if key not in u_bufs:
                    u_bufs[key] = d_u if not u_bufs else torch.empty_like(d_u)
                ubuf = u_bufs[key]
        ubuf = d_u if ubuf is None else ubuf
        fi, bi = work_item(w, nb)
        tl.append(('solve starts', w, time.perf_counter()))
        rhs_ptr = d_rhs.data_ptr() + (0 if node else bi * B * N * 16)
        info = op.solveDevice(rhs_ptr, ubuf.data_ptr(), B, N, layout='node' if node else 'rhs', support=d_support if (use_support[0] and node) else None)
        t = op.lastTiming()
        del op.factors                   # nothing is carried over between steps
        stamps.append(time.perf_counter())
        return fi, info, t

    def run_item(w, profile, ubuf=None):
        return solve_item(w, prepare_item(w, profile), ubuf)

    def run_items(ws, profile):
        """the K work items of a timed region.  Default: through the dispatcher's device pipeline (zephyr_amd.dispatch, what
        MultiFreq's parallel mode uses): a prepare thread builds / assembles / pre-factors item k+1 while item k is being solved."""
        if not args.pipeline:
            return [run_item(w, profile) for w in ws]
        from zephyr_amd import dispatch
        items = [dispatch.WorkItem((lambda op, w=w: solve_item(w, op)), (lambda w=w: prepare_item(w, profile))) for w in ws]
        from zephyr_amd import prefactor_many
        npipes = max(1, int(os.environ.get('HELM_BENCH_PIPES', '1')))

        def one_pipe(its):
            return list(dispatch.pipelined(its, device=local, lookahead=LOOKAHEAD, strict=os.environ.get('HELM_BENCH_STRICT', '0') == '1',
                                           solvers=nsolvers, group=args.group, group_prepare=prefactor_many if args.group > 1 else None))
        if npipes <= 1:
            return one_pipe(items)
        # (experiment) several independent pipelines on the one GPU, items dealt round-robin, each pipeline with a wavefield array of its own
        import threading
        outs = [None] * npipes
        bufs = [d_u] + [torch.empty_like(d_u) for _ in range(npipes - 1)]

        def run(k):
            sub = [dispatch.WorkItem((lambda op, w=w, k=k: solve_item(w, op, bufs[k])), (lambda w=w: prepare_item(w, profile))) for w in ws[k::npipes]]
            outs[k] = one_pipe(sub)
        ths = [threading.Thread(target=run, args=(k,)) for k in range(npipes)]
        for t_ in ths: t_.start()
        for t_ in ths: t_.join()
        res = [None] * len(ws)
        for k in range(npipes):
            res[k::npipes] = outs[k]
        return res

    def barrier(spares=False):
        if spares:                       # (the barrier that OPENS a pass: the pool's spares are topped up here, between passes, never inside one)
            try:
                from zephyr_amd import _lib as _zlb
                _zlb.load().helm_pool_spares(local, int(os.environ.get('HELM_POOL_SPARE', '3')))
            except Exception:
                pass
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.scaling == 'strong':
        # the job: work items 0 .. 15 (all 16 frequencies, 256 sources each), dealt round-robin over the ranks; a rank with nothing to do just waits
        timed_items = [w for w in range(NFREQ * nb) if w % world == rank]
        warm_items = timed_items[:args.warmup]
    else:
        timed_items = [rank + world * (args.warmup + k) for k in range(args.steps)]
        warm_items = [rank + world * k for k in range(args.warmup)]
    nsteps = max(1, len(timed_items)) if args.scaling == 'strong' else args.steps
    run_items(warm_items, True)          # (events on, like the timed region: the per-process event pool comes into being here, not inside it)

    # the interpreter's cyclic collector: everything alive now (torch, scipy, the model, the source matrices: ~10^6 objects) moves to the permanent generation,
    # so that a full collection triggered inside a timed region walks the job's own few objects, not the process (50-100 ms with torch imported).  The
    # collector stays ON; long-running dispatch loops do the same after their set-up (gc.freeze is the documented tool for it).
    import gc
    gc.collect()
    gc.freeze()
    barrier(True)
    del stamps[:]
    del tl[:]
    from zephyr_amd import _lib as _zl0
    if os.environ.get('HELM_ALLOC_TRACE'):
        sys.stderr.write('[bench] timed region starts\n'); sys.stderr.flush()
    cg0 = cgroup_cpu()
    cpu0 = time.process_time()
    _zl0.runtime_stats(reset=True)       # what the timed region makes the HIP runtime create (allocations, events, first launches) is counted from here
    t0 = time.perf_counter()
    results = [None] * len(timed_items)
    if args.streams <= 1:
        results = run_items(timed_items, True)
    else:
        # several work items in flight: each host thread drives its own operator handle (own HIP stream); ctypes
        # releases the GIL, so the latency-bound coarse multigrid levels of one item overlap the fine levels of another
        import threading
        bufs = [torch.empty((B, N), dtype=torch.complex128, device=dev) for _ in range(args.streams)]
        torch.cuda.synchronize()

        def worker(tid):
            for k in range(tid, len(timed_items), args.streams):
                results[k] = run_item(timed_items[k], True, bufs[tid])
        threads = [threading.Thread(target=worker, args=(t,)) for t in range(args.streams)]
        for th in threads:
            th.start()
        for th in threads:
            th.join()
    def aggregate(res):
        a = dict(iters=[], freqs=[], apply_ms=0.0, apply_launches=0, apply_bytes=0.0, solve_ms=0.0, gemm_ms=0.0, gemm_launches=0, gemm_flops=0.0,
                 factor_ms=0.0, methods=set(), big_ms=0.0, big_launches=0, big_flops=0.0, gemm_bytes=0.0, gemm_sol_ms=0.0)
        for fi, info, t in res:
            a['iters'] += [i['iterations'] for i in info]
            a['freqs'].append(float(freqs[fi]))
            a['methods'].update(i['method'] for i in info)
            for k_, t_ in (('apply_ms', 'apply_ms'), ('apply_launches', 'apply_launches'), ('apply_bytes', 'apply_bytes'), ('solve_ms', 'solve_ms'),
                           ('gemm_ms', 'gemm_ms'), ('gemm_launches', 'gemm_launches'), ('gemm_flops', 'gemm_flops'), ('factor_ms', 'factor_ms'),
                           ('big_ms', 'gemm_big_ms'), ('big_launches', 'gemm_big_launches'), ('big_flops', 'gemm_big_flops'),
                           ('gemm_bytes', 'gemm_bytes'), ('gemm_sol_ms', 'gemm_sol_ms')):
                a[k_] += t[t_]
        return a
    agg = aggregate(results)
    barrier()
    elapsed = time.perf_counter() - t0
    memtrace('after the timed region')
    rt_timed = _zl0.runtime_stats()
    cg1 = cgroup_cpu()
    rt_timed['cpu_cores_used'] = (time.process_time() - cpu0) / max(elapsed, 1e-9)      # CPU seconds of this process (all threads) per second of the timed region
    rt_timed['cpu_quota_cores'] = cg1[0]
    rt_timed['cpu_throttled_periods'] = cg1[1] - cg0[1]
    rt_timed['cpu_throttled_ms'] = (cg1[2] - cg0[2]) / 1e3
    if os.environ.get('HELM_ALLOC_TRACE'):
        sys.stderr.write('[bench] timed region ends\n'); sys.stderr.flush()
    elapsed_local = elapsed
    item_done_ms = [round(1e3 * (x - t0), 2) for x in stamps]
    first_items = [(a, int(w_), round(1e3 * (t_ - t0), 2)) for a, w_, t_ in sorted(tl, key=lambda m: m[2])][:12]

    def max_over_ranks(x):
        if world == 1:
            return x
        tt = torch.tensor([x], dtype=torch.float64, device=dev if backend == 'nccl' else 'cpu')
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())
    elapsed = max_over_ranks(elapsed)

    # Kernel-level roofline: in the pipelined region the factorisation of item k+1 shares the CUs with the solves of item k, so the
    # per-launch durations there measure the sharing (both kernels stretch).  The same K items therefore run once more strictly one after
    # the other with the events on; `roofline` quotes that pass, `roofline.in_pipeline` the stretched figures of the timed region itself.
    # r4: this pass runs with HELM_ND_SPARSE_RHS=0, i.e. every front of the forward pass and every row of the leaf back substitution is
    # multiplied whatever the right-hand sides hold -- gemm() books 8 M N K per launch, and a launch that skips the zero rows of a survey's point
    # sources would be credited with flops it did not do.  (The pipelined figures `in_pipeline` come from the timed region, where the skipping
    # is on: their TFLOP/s are marked `counts_skipped_flops`.)
    agg_k = agg
    if args.streams <= 1 and not args.no_roofline_pass:
        barrier()
        os.environ['HELM_ND_SPARSE_RHS'] = '0'
        if args.pipeline and args.group > 1:
            # the production launches: the factorisations of `group` items in the same batched launches (helm_prefactor_many), here with nothing beside them --
            # the set is factored, the GPU drained, then its items are solved one after the other
            from zephyr_amd import prefactor_many
            res_k = []
            for i0 in range(0, len(timed_items), args.group):
                ws_ = timed_items[i0:i0 + args.group]
                ops_ = [prepare_item(w, True) for w in ws_]
                prefactor_many(ops_)
                torch.cuda.synchronize()
                res_k += [solve_item(w, op_) for w, op_ in zip(ws_, ops_)]
            agg_k = aggregate(res_k)
        else:
            agg_k = aggregate([run_item(w, True) for w in timed_items])
        os.environ.pop('HELM_ND_SPARSE_RHS', None)
        barrier()

    # the same K work items once more with the per-launch HIP events off: what the event traffic of the roofline measurement costs
    elapsed_plain = None
    if args.streams <= 1 and not args.no_plain_pass:
        barrier(True)
        t1 = time.perf_counter()
        run_items(timed_items, False)
        barrier()
        elapsed_plain = max_over_ranks(time.perf_counter() - t1)
    # ... and once more with the forward pass computing every front whatever the right-hand sides hold (HELM_ND_SPARSE_RHS=0): what the job costs
    # when the sources are NOT the point sources of a survey (a dense right-hand side sets every flag itself and lands here too)
    elapsed_dense = None
    if args.streams <= 1 and not args.no_plain_pass and args.method in ('auto', 'direct'):
        os.environ['HELM_ND_SPARSE_RHS'] = '0'
        barrier(True)
        t1 = time.perf_counter()
        run_items(timed_items, False)
        barrier()
        elapsed_dense = max_over_ranks(time.perf_counter() - t1)
        os.environ.pop('HELM_ND_SPARSE_RHS', None)

    # ... and with the support of the sparse source matrix declared to the solver (helm_set_rhs_support; the reference's sources ARE scipy-sparse matrices): the leaf
    # level of the forward pass then does not read the dense right-hand sides to look for their nonzeros
    elapsed_support = None
    if args.streams <= 1 and not args.no_plain_pass and args.method in ('auto', 'direct') and d_support is not None:
        use_support[0] = True
        barrier(True)
        t1 = time.perf_counter()
        run_items(timed_items, False)
        barrier()
        elapsed_support = max_over_ranks(time.perf_counter() - t1)
        use_support[0] = False

    # ... and, under the default weak-scaling run, the job north_star names once through: all 16 frequencies x 256 sources = 4096 wavefields, the
    # 16 work items dealt round-robin over the ranks (events off) -- the strong-scaling figure of the same launch, so that a driver that only ever
    # calls `bench.py --gpus N` gets both curves
    strong_job = None
    if args.scaling == 'weak' and args.streams <= 1 and not args.no_plain_pass:
        job_items = [w for w in range(NFREQ * nb) if w % world == rank]
        barrier(True)
        t1 = time.perf_counter()
        run_items(job_items, False)
        barrier()
        tj = max_over_ranks(time.perf_counter() - t1)
        strong_job = {'wavefields': NFREQ * nb * B, 'seconds': tj, 'value': NFREQ * nb * B / tj, 'unit': 'wavefields/s',
                      'what': 'the whole 16-frequency job (every frequency x %d sources), its work items round-robin over the %d rank(s); max over ranks' % (B * nb, world)}
    memtrace('after the extra passes')
    wavefields = (NFREQ * nb * B) if args.scaling == 'strong' else world * args.steps * B
    value = wavefields / elapsed

    # multi-GPU readiness, verifiable from the line of an N-rank run: how many ranks the collective library really joined (an all-reduce of ones),
    # every rank's own seconds per step (the headline takes the slowest), and -- in config4_leg -- the gradient all-reduce on its own
    ranks_seen, rank_ms = 1, [1e3 * elapsed_local / nsteps]
    if world > 1:
        one = torch.ones(1, dtype=torch.float64, device=dev if backend == 'nccl' else 'cpu')
        dist.all_reduce(one, op=dist.ReduceOp.SUM)
        ranks_seen = int(round(float(one.item())))
        mine = torch.tensor([1e3 * elapsed_local / nsteps], dtype=torch.float64, device=dev if backend == 'nccl' else 'cpu')
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        rank_ms = [float(t.item()) for t in every]
    # config 4 on every rank (its frequencies are sharded over the ranks and the gradient crosses the all-reduce)
    c4 = None
    if not args.no_config4 and args.method in ('auto', 'direct'):
        try:
            c4 = config4_leg(local, world, rank, dist, backend)
        except Exception as exc:
            if world > 1:
                raise                     # (a rank that drops out of a collective leg must not leave the others waiting silently)
            c4 = 'failed: %s' % exc

    out = None
    if rank == 0:
        direct = agg['methods'] == {4}
        how = ('sparse direct: nested-dissection multifrontal factorisation of A(f) on the GPU, kept for all sources of the frequency, '
               'triangular solves as batched complex GEMMs + iterative refinement with the stencil kernel' if direct else
               'BiCGSTAB right-preconditioned by shifted-Laplacian multigrid with damped-Jacobi smoothing + PML line relaxation')
        where = (('a pass over the same K work items strictly one after the other (no other kernel on the GPU; the factorisations of %d items per set of launches as in the timed '
                  'region), events on, HELM_ND_SPARSE_RHS=0 (every booked flop and byte is executed); in_pipeline = the same quantities from the timed region'
                  % (args.group if args.pipeline else 1)) if args.streams <= 1 else 'the timed region')

        def stencil_block(a):
            ach = (a['apply_bytes'] / (a['apply_ms'] * 1e-3)) / 1e9 if a['apply_ms'] > 0 else 0.0
            return {'achieved': ach, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': ach / HBM_PEAK_GBS, 'launches_timed': int(a['apply_launches']),
                    'avg_launch_us': 1e3 * a['apply_ms'] / a['apply_launches'] if a['apply_launches'] else None,
                    'bytes_per_launch_algorithmic': a['apply_bytes'] / a['apply_launches'] if a['apply_launches'] else None,
                    'apply_share_of_solve_time': a['apply_ms'] / a['solve_ms'] if a['solve_ms'] > 0 else None}

        def gemm_block(a):
            tf = a['gemm_flops'] / (a['gemm_ms'] * 1e-3) / 1e12 if a['gemm_ms'] > 0 else 0.0
            return {'achieved': tf, 'peak': F64_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': tf / F64_PEAK_TFLOPS,
                    'launches_timed': int(a['gemm_launches']), 'avg_launch_us': 1e3 * a['gemm_ms'] / a['gemm_launches'] if a['gemm_launches'] else None,
                    'flops_per_launch_algorithmic': a['gemm_flops'] / a['gemm_launches'] if a['gemm_launches'] else None,
                    'gemm_share_of_solve_time': a['gemm_ms'] / a['solve_ms'] if a['solve_ms'] > 0 else None,
                    # the launches are not all bound by the same roof: the thin fronts low in the tree are HBM-bound products.  Per launch
                    # max(flops / fp64 peak, operand bytes / 8 TB/s) with every operand counted once; their sum against the measured time
                    'two_roofs': {'frac': a['gemm_sol_ms'] / a['gemm_ms'] if a['gemm_ms'] > 0 else None,
                                  'roofline_ms_per_item': a['gemm_sol_ms'] / max(1, len(a['freqs'])), 'measured_ms_per_item': a['gemm_ms'] / max(1, len(a['freqs'])),
                                  'operand_GB_per_item': a['gemm_bytes'] / max(1, len(a['freqs'])) / 1e9,
                                  'flop_per_byte': a['gemm_flops'] / a['gemm_bytes'] if a['gemm_bytes'] > 0 else None,
                                  'what': 'sum over launches of max(8MNK / 78.6 TFLOP/s, 16 (MK + KN + MN (1 or 2)) / 8 TB/s) divided by their measured time'},
                    'launches_of_at_least_1_GFLOP': {'launches': int(a['big_launches']), 'share_of_gemm_time': a['big_ms'] / a['gemm_ms'] if a['gemm_ms'] > 0 else None,
                                                     'achieved': a['big_flops'] / (a['big_ms'] * 1e-3) / 1e12 if a['big_ms'] > 0 else None, 'unit': 'TFLOP/s',
                                                     'frac': a['big_flops'] / (a['big_ms'] * 1e-3) / 1e12 / F64_PEAK_TFLOPS if a['big_ms'] > 0 else None}}
        stencil = {'bound': 'hbm',
                   'kernel': ('k_resid_nm (9-pt complex128 stencil apply in the node-major layout of the direct path with the residual q - A x and its norms fused), '
                              'launches inside the solves; apply_microbench = the rhs-major batched apply k_stencil' if direct else
                              'k_stencil (batched 9-pt complex128 apply with fused dot-product / residual epilogue), launches inside the timed solves'),
                   'bytes_formula': ('N*(32*B + 144) per norm-only launch (x and q in, nine coefficients once, nothing out), + N*16*B when r is stored' if direct else
                                     'N*(32*B_active + 144) for the apply + N*16*B_active for the epilogue operand it must read'),
                   'measured_on': where, 'traffic': None}
        stencil.update(stencil_block(agg_k))
        if agg_k is not agg:
            stencil['in_pipeline'] = stencil_block(agg)
        iters = agg['iters']
        out = {
            'metric': 'wavefields/sec (freq x source solves/s) on 1024^2 grid',
            'value': value, 'unit': 'wavefields/s', 'n_gpus': world, 'steps': nsteps, 'warmup': args.warmup,
            'ms_per_step': 1e3 * elapsed / nsteps, 'higher_is_better': True, 'scaling': args.scaling,
            'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'timed_region': 'K work items through the device pipeline with per-launch HIP events on; `unprofiled` repeats the same K items with the events off',
            'pipeline': (('device pipeline of zephyr_amd.dispatch (MultiFreq parallel mode): items k+1 .. k+%d are created and assembled, their factorisations enqueued TOGETHER '
                          '(helm_prefactor_many: the fronts of %d frequencies in the same batched launches, high-priority stream) while the items before them are being solved'
                          % (args.group, args.group)) if args.pipeline and args.group > 1 else
                         'device pipeline of zephyr_amd.dispatch (MultiFreq parallel mode): item k+1 is created, assembled and its factorisation enqueued '
                         '(helm_prefactor, high-priority stream) while item k is being solved' if args.pipeline else 'off: items strictly one after the other'),
            'factorisations_per_launch_set': args.group if args.pipeline else 1,
            'item_done_ms': item_done_ms, 'first_items_timeline_ms': first_items,
            'unprofiled': None if elapsed_plain is None else {'value': wavefields / elapsed_plain, 'ms_per_step': 1e3 * elapsed_plain / nsteps},
            'strong_scaling_job': strong_job,
            'support_declared': None if elapsed_support is None else {
                'value': wavefields / elapsed_support, 'ms_per_step': 1e3 * elapsed_support / nsteps,
                'what': 'the same K items (events off) with the support of the scipy-sparse source matrix handed to the solver (solveDevice(..., support=), '
                        'helm_set_rhs_support): no scan of the dense right-hand sides at the leaf level of the forward pass. NOT the headline: `value` lets the library find the nonzeros itself'},
            'every_front_computed': None if elapsed_dense is None else {
                'value': wavefields / elapsed_dense, 'ms_per_step': 1e3 * elapsed_dense / nsteps,
                'what': 'the same K items (events off) with HELM_ND_SPARSE_RHS=0: the forward pass visits every front; `value` lets it skip the fronts '
                        'whose right-hand-side rows (81 nonzeros per Kaiser source at the surface) and whose children are all zero in a block of 64 columns'},
            # the driver's record keeps the first 24 scalar keys of `config` and 120 characters of a string: the flat keys are written at the end of main(),
            # decisive ones first; prose and lists live in `detail`
            'config': {'workload': ('Eurus 2D iso %dx%d synth-Marmousi dx=%gm, 16 freqs 2-9.5Hz x 256 src; step=assemble+factor 1 freq+solve %d src, relres<=%g'
                                    % (n, n, dx, B, args.rtol))[:120]},
            'detail': {'workload': 'Eurus 2D isotropic %dx%d synthetic-Marmousi (seed 20240512, dx=%g m), 16 freqs 2-9.5 Hz x 256 Kaiser sources; '
                                   'step = create + assemble 1 frequency + solve %d sources to true relres<=%g (method=%s: %s)' % (n, n, dx, B, args.rtol, args.method, how),
                       'buffers': ("node-major: right-hand sides and wavefields in the reference's own (N, nsrc) C-order arrays, resident in HBM" if node else
                                   'rhs-major: one right-hand side / wavefield per row, resident in HBM'),
                       'grid': [n, n], 'sources_per_step': B, 'work_items_in_flight': args.streams, 'freqs_hz_this_run': agg['freqs'], 'sharding': ('strong: the 16 work items of the job (one frequency x 256 sources each) round-robin over ranks; value = 4096 wavefields / slowest rank' if args.scaling == 'strong' else 'work items (freq, source batch) round-robin over ranks'),
                       'solves_or_iterations_per_rhs_mean': float(np.mean(iters)) if iters else None,
                       'solves_or_iterations_per_rhs_max': int(np.max(iters)) if iters else None,
                       'runtime_objects_created_in_timed_region': rt_timed,
                       'device_ms_per_step': {'solve_call': agg['solve_ms'] / nsteps, 'of_which_factorisation': agg['factor_ms'] / nsteps,
                                              'note': 'pipelined: solve_call covers the triangular solves + residual checks of an item, the factorisation (of_which_factorisation: its span on its own '
                                                      'stream, beside the previous item) is no longer inside it' if args.pipeline else 'serial: the factorisation is inside solve_call'}},
        }
        if direct:
            out['roofline'] = {'bound': 'mfma',
                               'kernel': ('k_zgemm3 (strided-batched complex128 GEMM of the multifrontal factorisation and triangular solves on the matrix cores: four real '
                                          'v_mfma_f64_16x16x4_f64 per complex 16x16x4 block; peak = the dense fp64 MFMA rate of MI355X, 78.6 TFLOP/s, which '
                                          'tools/fp64_clock.hip reaches to 99 % from two waves per SIMD up)'),
                               'flops_formula': '8*M*N*K per batch item (4 real multiply-adds per complex one)', 'measured_on': where, 'traffic': None}
            out['roofline'].update(gemm_block(agg_k))
            if agg_k is not agg:
                out['roofline']['in_pipeline'] = gemm_block(agg)
                out['roofline']['in_pipeline']['counts_skipped_flops'] = True
                out['roofline']['serial_pass_env'] = 'HELM_ND_SPARSE_RHS=0: every booked flop is executed'
            out['stencil_roofline'] = stencil
        else:
            out['roofline'] = stencil
        out['roofline_northstar'] = stencil        # the kernel BASELINE.json's north_star names: the 9-point stencil apply against the HBM roofline
        micro_target = out['stencil_roofline'] if direct else out['roofline']
        # stencil-apply microbenchmark of SURVEY.md 8(d) (outside the timed region): Y = A X on random X, B right-hand sides,
        # algorithmic bytes N*(32*B + 144), HIP events on the solver stream
        if world == 1:
            try:
                import ctypes
                from zephyr_amd import _lib
                sc = dict(cfg); sc.update(freq=float(freqs[8]), device=local)
                opm = Eurus(sc)
                opm.setProfiling(True)
                gen = torch.Generator(device=dev); gen.manual_seed(1234)
                micro = []
                for Bm in (1, 8, 32, 64):
                    X = torch.randn((Bm, N, 2), dtype=torch.float64, device=dev, generator=gen)
                    Y = torch.empty_like(X)
                    torch.cuda.synchronize()
                    ms = by = 0.0
                    for rep in range(6):
                        _lib.check(_lib.load().helm_apply_device(opm.handle, 0, 0, ctypes.c_void_p(X.data_ptr()), ctypes.c_void_p(Y.data_ptr()), Bm), opm.handle)
                        tm = opm.lastTiming()
                        if rep:
                            ms += tm['apply_ms']; by += tm['apply_bytes']
                    row = {'B': Bm, 'us': 1e3 * ms / 5, 'GBps': by / (ms * 1e-3) / 1e9, 'frac_of_peak': by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
                    if (2 * Bm * N * 16 + 9 * N * 16) < 256 * 2 ** 20:
                        # in + out + coefficient planes fit the 256 MiB Infinity Cache and the launches run back to back: this row measures the cache, not HBM
                        row['note'] = 'Infinity-Cache resident (footprint %.0f MB < 256 MiB, back-to-back launches): not HBM evidence; quote B >= 8' % ((2 * Bm * N * 16 + 9 * N * 16) / 1e6)
                    micro.append(row)
                    del X, Y
                micro_target['apply_microbench'] = micro
                del opm.factors
            except Exception as exc:          # never let the extra measurement break the bench line
                micro_target['apply_microbench'] = 'failed: %s' % exc
        # HBM traffic per launch from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of this same command in two separate
        # runs, tools/run_profiles_r3.sh; bench.py cannot run the profiler on itself) -- only when they were collected for this workload
        def pmc(name):
            # the newest round's file first (profiles/rNN_<name>.json)
            import glob
            for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]_%s.json' % name)), reverse=True):
                try:
                    d_ = json.load(open(path))
                    if d_.get('batch') == B and d_.get('grid') == [n, n]:
                        return d_, 'profiles/' + os.path.basename(path)
                except Exception:
                    pass
            return None, None
        try:
            if direct:
                pz, src_z = pmc('pmc_traffic_zgemm')
                if pz:
                    out['roofline']['traffic'] = pz['traffic_bytes_per_launch']
                    out['roofline']['traffic_source'] = src_z + ' (HBM bytes per GEMM launch, FETCH_SIZE x2 + WRITE_SIZE, separate PMC passes)'
                    try:      # the same per WORK ITEM (the PMC run's own item count: one residual launch each) against the operand bytes the library books per item here
                        tr = out['roofline']
                        pr0, _ = pmc('pmc_traffic_resid_nm')
                        items_pmc = float(pr0['launches_fetch_pass']) if pr0 and pr0.get('launches_fetch_pass') else 2.0
                        per_item = pz['traffic_bytes_per_launch'] * pz['launches_fetch_pass'] / items_pmc / 1e9
                        tr['traffic_GB_per_item'] = per_item
                        tr['traffic_over_operand_bytes'] = per_item / tr['two_roofs']['operand_GB_per_item']
                    except Exception:
                        pass
                pr, src_r = pmc('pmc_traffic_resid_nm')
                if pr:
                    out['stencil_roofline']['traffic'] = pr['traffic_bytes_per_launch']
                    out['stencil_roofline']['traffic_source'] = src_r + ' (HBM bytes per residual-kernel launch)'
            else:
                pk = json.load(open(os.path.join(ROOT, 'profiles', 'r01_krylov_pmc_traffic.json')))
                if pk.get('batch') == B and pk.get('grid') == [n, n]:
                    out['roofline']['traffic'] = pk['traffic_bytes_per_launch_outer_applies']
                    out['roofline']['traffic_source'] = 'profiles/r01_krylov_pmc_traffic.json'
        except Exception:
            pass
        if world == 1 and not args.no_host_api and direct:
            # The reference's actual call shape (discretization.py:101-103, distributors.py:127-173): `MultiFreq(systemConfig) * q` with a
            # scipy-sparse source matrix in and one (N, nsrc) numpy array per frequency out -- all 16 frequencies of the job, every
            # wavefield copied back to the host (4.3 GB per frequency over PCIe into pinned memory).  PCIe-inclusive: never `value`.
            try:
                from zephyr_amd import MultiFreq
                del d_u
                torch.cuda.empty_cache()
                from zephyr_amd import _lib as _zl1
                _zl1.load().helm_trim()          # (every leg starts from empty pools and warms its own: what the passes above left idle -- tens of GB -- goes back to the device)
                memtrace('before the host-API leg')
                best = None
                for wpd in (1, 3):
                    os.environ['HELM_WORKERS_PER_DEVICE'] = str(wpd)
                    sch = dict(cfg); sch.update(freqs=[float(f) for f in freqs], Disc=Eurus, rtol=args.rtol, maxit=400000, method=args.method, batch=NSRC,
                                               device=local)        # this leg is a one-GPU figure like `value`, also on a node with eight
                    # untimed warm-up (like the W warm-up steps): the pinned result buffers and device pools come into being here
                    from zephyr_amd import _lib as _zl
                    from zephyr_amd import dispatch as _zd
                    _zl.pinned_reserve((N, NSRC), wpd * (_zd.results_ahead() + 1) + 1)
                    mfw = MultiFreq(dict(sch))            # (the whole job once: every operator's device buffers then come from the library's pool)
                    for u in mfw * q_sparse:
                        del u
                    del mfw.factors
                    mf = MultiFreq(sch)
                    torch.cuda.synchronize()
                    th0 = time.perf_counter()
                    chk = 0.0
                    arrivals = []
                    for u in mf * q_sparse:
                        chk += float(abs(u[N // 2 + 7, 0]))       # touch the result; the array goes back to the pinned pool when dropped
                        arrivals.append(int(1e3 * (time.perf_counter() - th0)))
                        del u
                    th = time.perf_counter() - th0
                    del mf.factors
                    rec = {'value': NFREQ * NSRC / th, 'unit': 'wavefields/s', 'seconds': th, 'workers_per_device': wpd, 'arrivals_ms': arrivals}
                    if best is None or rec['value'] > best['value']:
                        best = rec
                    out.setdefault('value_host_api_runs', []).append(rec)
                os.environ.pop('HELM_WORKERS_PER_DEVICE', None)
                best = dict(best)
                best['what'] = ('MultiFreq(Disc=Eurus, 16 freqs) * q, q = scipy-sparse (N, 256) Kaiser sources, results = 16 numpy arrays (N, 256) complex128 in pinned host '
                                'memory; includes operator construction, assembly, factorisation, the solves and the device-to-host copy of every wavefield')
                out['value_host_api'] = best
            except Exception as exc:
                out['value_host_api'] = 'failed: %s' % exc
            try:
                from zephyr_amd import _lib as _zl2
                _zl2.load().helm_trim(); _zl2.load().helm_host_trim()
            except Exception:
                pass
            memtrace('after the host-API leg')
            d_u = torch.empty((N, B) if node else (B, N), dtype=torch.complex128, device=dev)
        out['config4'] = c4
        if world == 1 and not args.no_config2 and direct:
            try:
                del d_u
                torch.cuda.empty_cache()
                out['config2'] = config2_leg(local, rtol=args.rtol, method=args.method)
            except Exception as exc:
                out['config2'] = 'failed: %s' % exc
            d_u = torch.empty((N, B) if node else (B, N), dtype=torch.complex128, device=dev)
        if world == 1 and not args.no_config5 and n == 1024:
            try:
                del d_u
                torch.cuda.empty_cache()
                out['config5'] = config5_leg(local, rtol=args.config5_rtol)
            except Exception as exc:
                out['config5'] = 'failed: %s' % exc
            d_u = torch.empty((N, B) if node else (B, N), dtype=torch.complex128, device=dev)
        if world == 1 and not args.no_cpu:
            cb, u_lu = cpu_baseline(cfg, freqs, q_all)
            out['cpu_baseline'] = cb
            # parity gate of the bench itself: the LU wavefields the CPU leg just computed against the GPU path on the same frequency / sources
            try:
                ns = min(u_lu.shape[1], B)
                u_lu = u_lu[:, :ns]
                sc = dict(cfg); sc.update(freq=cb['freq_hz'], rtol=args.rtol, maxit=400000, method=args.method, batch=B, device=local)
                opp = Eurus(sc)
                d_rp = torch.from_numpy(np.ascontiguousarray(q_all[:, :ns].T)).to(dev)
                d_up = torch.empty((ns, N), dtype=torch.complex128, device=dev)
                opp.solveDevice(d_rp.data_ptr(), d_up.data_ptr(), ns, N)
                u_gpu = d_up.cpu().numpy().T
                rel = np.linalg.norm(u_gpu - u_lu, axis=0) / np.linalg.norm(u_lu, axis=0)
                out['parity_vs_lu_max_rel'] = float(rel.max())
                out['parity_vs_lu'] = {'freq_hz': cb['freq_hz'], 'sources': int(ns), 'grid': [n, n], 'tolerance': 1e-7, 'ok': bool(rel.max() <= 1e-7),
                                       'what': 'max over sources of ||u_gpu - u_lu|| / ||u_lu||, u_lu = SciPy SuperLU solve of the oracle matrix (the cpu_baseline leg)'}
                del opp.factors
            except Exception as exc:
                out['parity_vs_lu_max_rel'] = None
                out['parity_vs_lu'] = 'failed: %s' % exc
            if not args.no_cpu_2n and n == 1024:
                try:
                    out['cpu_baseline_2n'] = cpu_baseline_2n()
                except Exception as exc:
                    out['cpu_baseline_2n'] = 'failed: %s' % exc
            if not args.no_cpu_pool and (os.cpu_count() or 1) >= 32:
                try:
                    out['cpu_baseline_pool'] = cpu_baseline_pool(cfg, freqs)
                except Exception as exc:
                    out['cpu_baseline_pool'] = 'failed: %s' % exc
        # The driver's record keeps the first 24 SCALAR entries of `config` and 120 characters of a string (round 5's 40-odd keys lost every config-4 / config-5 /
        # parity / multi-GPU figure): exactly 24 flat keys, decisive ones first; everything else is in `detail`, `config2/4/5`, `roofline`.
        try:
            cfgd = out['config']
            det = out['detail']
            dms = det.get('device_ms_per_step') or {}
            c2 = out.get('config2') if isinstance(out.get('config2'), dict) else {}
            c4d = c4 if isinstance(c4, dict) else {}
            c5 = out.get('config5') if isinstance(out.get('config5'), dict) else {}
            rl = out['roofline']
            gaps = np.diff(np.array([0.0] + item_done_ms)) if item_done_ms else np.zeros(1)
            unprof = (out.get('unprofiled') or {}).get('value')
            strong = (out.get('strong_scaling_job') or {}).get('value')
            flat = [
                ('unprofiled_wfs', unprof),
                ('timed_max_item_gap_ms', float(gaps.max())), ('timed_p50_item_ms', float(np.median(gaps))),
                ('timed_cpu_throttled_ms', rt_timed['cpu_throttled_ms']), ('cpu_quota_cores', rt_timed['cpu_quota_cores']),
                ('timed_dev_allocs', rt_timed['dev_allocs']), ('timed_dev_alloc_ms', rt_timed['dev_alloc_ms']), ('timed_first_launches', rt_timed['first_launches']),
                ('roofline_frac_serial', rl.get('frac')), ('two_roofs_frac', (rl.get('two_roofs') or {}).get('frac')),
                ('stencil_frac', (out.get('stencil_roofline') or rl).get('frac')),
                ('c4_dpred_s', c4d.get('dpred_seconds')), ('c4_jtvec_s', c4d.get('jtvec_seconds')), ('c4_gpu_ms', c4d.get('gpu_ms')),
                ('c5_job_s', c5.get('job_seconds')),
                ('c5_apply_frac_B16', next((a_.get('frac_of_peak') for a_ in (c5.get('apply') or []) if a_.get('B') == 16), None)),
                ('parity_vs_lu_max_rel', out.get('parity_vs_lu_max_rel')),
                ('cpu_baseline_wfs', out['cpu_baseline'].get('value') if isinstance(out.get('cpu_baseline'), dict) else None),
                ('rccl_ranks_seen', ranks_seen), ('gradient_allreduce_ms', c4d.get('gradient_allreduce_ms_512')),
                ('ms_per_step_slowest_rank', max(rank_ms)),
                ('strong_job_wfs', strong), ('c2_wfs_device', c2.get('device_wfs')),
            ]
            assert len(flat) + len(cfgd) <= 24
            for k, v in flat:
                cfgd[k] = float('%.6g' % v) if isinstance(v, float) else v
            more = {
                'grid_n': n, 'freqs_this_run': len(agg['freqs']),
                'dense_rhs_wfs': (out.get('every_front_computed') or {}).get('value'),
                'timed_first_launch_ms': rt_timed['first_launch_ms'], 'timed_dev_alloc_mb': rt_timed['dev_alloc_bytes'] / 1e6,
                'timed_pinned_allocs': rt_timed['host_allocs'], 'timed_cpu_throttled_periods': rt_timed['cpu_throttled_periods'], 'timed_cpu_cores_used': rt_timed['cpu_cores_used'], 'wait_sleep_us': int(os.environ.get('HELM_SYNC_SLEEP_US', '0') or 0),
                'timed_slow_syncs': rt_timed['slow_syncs'], 'timed_worst_sync_ms': rt_timed['worst_sync_ms'],
                'timed_events_created': rt_timed['events_created'], 'timed_streams_created': rt_timed['streams_created'],
                'kernels_registered': rt_timed['kernels_registered'], 'kernels_resolved_by_warm': rt_timed['kernels_resolved'], 'warm_ms': rt_timed['warm_ms'],
                'device_ms_solve_call': dms.get('solve_call'), 'device_ms_factorisation': dms.get('of_which_factorisation'),
                'in_pipeline_frac': (rl.get('in_pipeline') or {}).get('frac'), 'gemm_avg_launch_us': rl.get('avg_launch_us'),
                'support_declared_wfs': (out.get('support_declared') or {}).get('value'), 'strong_job_s': (out.get('strong_scaling_job') or {}).get('seconds'),
                'host_api_wfs': out['value_host_api'].get('value') if isinstance(out.get('value_host_api'), dict) else None,
                'c2_ms_per_item': c2.get('device_ms_per_item'), 'c2_wfs_host_api': c2.get('host_api_wfs'),
                'c2_gemm_frac': c2.get('gemm_frac_serial'), 'c2_worst_relres': c2.get('worst_relres'),
                'c4_allreduce_ms_512': c4d.get('gradient_allreduce_ms_512'), 'c4_allreduce_ms_1024': c4d.get('gradient_allreduce_ms_1024'),
                'c5_rtol': c5.get('rtol'), 'c5_job_s_rtol1e10': c5.get('job_seconds_rtol1e10'),
                'c5_apply_us_B16': next((a_.get('us') for a_ in (c5.get('apply') or []) if a_.get('B') == 16), None),
                'collective_backend': backend if world > 1 else 'none', 'ms_per_step_fastest_rank': min(rank_ms),
                # (N = 1) the weak-scaling headline and the whole 16-frequency job are the same pipeline over 20 and 16 items: they must agree
                'weak_over_strong': (value / strong) if strong else None,
                'weak_equals_strong_within_5pct': (abs(value / strong - 1.0) <= 0.05) if (strong and world == 1) else None,
                'headline_over_unprofiled': (value / unprof) if unprof else None,
            }
            for r_, ms_ in enumerate(rank_ms):
                more['ms_per_step_rank%d' % r_] = ms_
            det['flat'] = {k: (float('%.6g' % v) if isinstance(v, float) else v) for k, v in more.items()}
            if more['weak_equals_strong_within_5pct'] is False:
                sys.stderr.write('bench.py: WARNING weak-scaling value %.0f and strong_job_wfs %.0f differ by more than 5 %%\n' % (value, strong))
        except Exception as exc:
            out['config']['flat_keys_error'] = str(exc)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
