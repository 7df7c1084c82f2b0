"""Run-to-run reproducibility probe of the 2-D direct path (GPU box): which step makes two factorisations of the same system differ in their last bits?
Prints, per switch setting, how many DISTINCT wavefield arrays `reps` fresh factorisations of one system produce (1 = reproducible)."""
import os, sys, hashlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import zephyr_amd as za

KNOBS = ('HELM_ND_POISON', 'HELM_ND_STABLE', 'HELM_ND_SPARSE_RHS', 'HELM_ND_DIRECT_OUT', 'HELM_ND_GJSTEP', 'HELM_ND_FUSEDLEAF', 'HELM_ND_OVERLAP_NM', 'HELM_ND_XCDMAP', 'HELM_ND_LEAF')


def device_solve(op, q):
    R = torch.from_numpy(np.ascontiguousarray(q)).cuda()
    U = torch.empty_like(R)
    op.solveDevice(R.data_ptr(), U.data_ptr(), q.shape[1], q.shape[0], layout='node')
    torch.cuda.synchronize()
    return U.cpu().numpy()


def case(name, cls_name, rough, nrhs, env, reps=8, nz=150, nx=170):
    for k in KNOBS:
        os.environ.pop(k, None)
    os.environ.update(env)
    rng = np.random.default_rng(11)
    c = 1800. + 2000. * rng.random((nz, nx)) if rough else 2500. + 500. * np.sin(np.arange(nz)[:, None] / 20.) * np.ones((nz, nx))
    cfg = dict(nx=nx, nz=nz, dx=10., dz=10., c=c, freq=8., nPML=8, rtol=1e-10, method='direct', batch=256)
    locs = np.stack([rng.uniform(100., 10. * nx - 100., nrhs), rng.uniform(20., 60., nrhs)], axis=1)
    q = za.SparseKaiserSource(cfg)(locs).toarray()
    seen = []
    for _ in range(reps):
        op = getattr(za, cls_name)(cfg)
        seen.append(hashlib.sha1(device_solve(op, q).tobytes()).hexdigest()[:6])
        del op.factors
    print('%-40s distinct %d of %d   %s' % (name, len(set(seen)), reps, ' '.join(seen)), flush=True)


if __name__ == '__main__':
    case('MiniZephyr smooth', 'MiniZephyr', 0, 9, {})
    case('MiniZephyr rough', 'MiniZephyr', 1, 9, {})
    case('Eurus smooth', 'Eurus', 0, 9, {})
    case('Eurus rough', 'Eurus', 1, 9, {})
    for knob, val in (('HELM_ND_STABLE', '0'), ('HELM_ND_GJSTEP', '0'), ('HELM_ND_FUSEDLEAF', '0'), ('HELM_ND_OVERLAP_NM', '0'), ('HELM_ND_XCDMAP', '0'), ('HELM_ND_SPARSE_RHS', '0'),
                      ('HELM_ND_LEAF', '4'), ('HELM_ND_LEAF', '16')):
        case('MiniZephyr smooth %s=%s' % (knob, val), 'MiniZephyr', 0, 9, {knob: val})
    case('MiniZephyr smooth 64x64', 'MiniZephyr', 0, 9, {}, nz=64, nx=64)
    case('MiniZephyr smooth 300x300', 'MiniZephyr', 0, 9, {}, nz=300, nx=300)
