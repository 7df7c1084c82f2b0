#!/usr/bin/env python3
"""Config 4 probe: one FWI gradient (forward + back-propagation + imaging condition) on the 512 x 512 model,
8 frequencies x 64 sources, 128 receivers, device-resident wavefields; compares with the host imaging path."""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
g.build()
import zephyr_amd as za
from zephyr_amd.models import marmousi_like, box_smooth
from zephyr_amd.problem import Helm2DProblem
from zephyr_amd.survey import Helm2DSurvey

ap = argparse.ArgumentParser()
ap.add_argument('--n', type=int, default=512); ap.add_argument('--nfreq', type=int, default=8); ap.add_argument('--nsrc', type=int, default=64)
ap.add_argument('--check-host', action='store_true')
a = ap.parse_args()
n, dx = a.n, 10.
ctrue = marmousi_like(n, n, dx)
ccur = box_smooth(ctrue, 12)
freqs = list(np.linspace(3., 10., a.nfreq))
src = np.stack([np.linspace(200., dx * n - 200., a.nsrc), np.full(a.nsrc, 20.)], 1)
rec = np.stack([np.linspace(100., dx * n - 100., 128), np.full(128, 20.)], 1)
base = dict(nx=n, nz=n, dx=dx, dz=dx, freqs=freqs, Disc=za.Eurus, geom=dict(src=src, rec=rec, mode='fixed'), batch=64)

def make(c, **kw):
    sc = dict(base, c=c, **kw)
    p, s = Helm2DProblem(sc), Helm2DSurvey(sc)
    p.pair(s)
    return p, s

t0 = time.time(); ptrue, strue = make(ctrue); dobs = strue.dpred(); t_true = time.time() - t0
del ptrue.factors
pcur, scur = make(ccur)
t0 = time.time(); dcur = scur.dpred(); t_fwd = time.time() - t0
resid = dcur - dobs
t0 = time.time(); gdev = pcur.Jtvec(None, resid); t_grad = time.time() - t0
out = dict(grid=[n, n], nfreq=a.nfreq, nsrc=a.nsrc, nrec=128, dpred_seconds=t_fwd, wavefields_per_s_forward=a.nfreq * a.nsrc / t_fwd,
           gradient_seconds=t_grad, wavefields_per_s_gradient=2 * a.nfreq * a.nsrc / t_grad, grad_norm=float(np.linalg.norm(gdev)))
if a.check_host:
    ph, sh = make(ccur, hostGradient=True)
    t0 = time.time(); gh = ph.Jtvec(None, resid); out['host_gradient_seconds'] = time.time() - t0
    out['device_vs_host_rel'] = float(np.linalg.norm(gdev - gh) / np.linalg.norm(gh))
print(json.dumps(out))
