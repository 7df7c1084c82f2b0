"""Deterministic synthetic velocity models for benchmarks and parity tests.

No Marmousi file exists in the reference tree or in this image (SURVEY.md §7 hard part 7),
so the BASELINE configs use a seeded "Marmousi-scale" generator: a water layer over dipping,
faulted sedimentary layers with a velocity gradient, lateral variation and a few high-velocity
bodies, 1500-5500 m/s, smoothed with a 3-point box filter.
"""
import numpy as np

MARMOUSI_SEED = 20240512


def _box3(a):
    p = np.pad(a, 1, mode='edge')
    out = np.zeros_like(a)
    for dz in range(3):
        for dx in range(3):
            out += p[dz:dz + a.shape[0], dx:dx + a.shape[1]]
    return out / 9.0


def box_smooth(a, passes=1):
    """Repeated 3x3 box smoothing (passes=1 -> '3-pt'; 12 passes ~ a 25-pt kernel)."""
    a = np.asarray(a, dtype=np.float64)
    for _ in range(passes):
        a = _box3(a)
    return a


def marmousi_like(nz, nx, dx=10.0, dz=None, seed=MARMOUSI_SEED, vmin=1500.0, vmax=5500.0):
    """Seeded layered + dipping + faulted velocity model, (nz,nx) float64, m/s.

    Geometry is defined in physical units so that the same seed gives the same geology
    at different grid sizes (512^2 @10 m and 1024^2 @9 m cover 5.1 km and 9.2 km).
    """
    dz = dx if dz is None else dz
    rng = np.random.default_rng(seed)
    x = (np.arange(nx) * dx)[None, :]
    z = (np.arange(nz) * dz)[:, None]
    Lx = nx * dx
    Lz = nz * dz

    nlay = 28
    # layer base depths (fraction of Lz) and layer velocities increasing with depth + jitter
    tops = np.sort(rng.uniform(0.06, 1.0, nlay))
    vlay = vmin + (vmax - vmin) * (np.linspace(0.08, 0.95, nlay) ** 1.1) + rng.normal(0, 180.0, nlay)
    vlay = np.clip(vlay, vmin + 100.0, vmax)
    # a few velocity inversions (low-velocity layers)
    for k in rng.choice(np.arange(4, nlay - 2), size=4, replace=False):
        vlay[k] -= rng.uniform(300.0, 700.0)

    # structural deformation: regional dip + two anticlines + three listric-ish faults
    dip = rng.uniform(-0.12, 0.12)
    shift = dip * (x - Lx / 2)
    for _ in range(2):
        xc = rng.uniform(0.2, 0.8) * Lx
        w = rng.uniform(0.12, 0.3) * Lx
        amp = rng.uniform(0.04, 0.10) * Lz
        shift = shift - amp * np.exp(-((x - xc) / w) ** 2)
    zdef = z - shift
    for _ in range(3):
        xf = rng.uniform(0.15, 0.85) * Lx
        slope = rng.uniform(0.35, 0.9) * rng.choice([-1.0, 1.0])
        throw = rng.uniform(0.015, 0.045) * Lz
        side = (x - xf - slope * z) > 0
        zdef = zdef + throw * side

    frac = zdef / Lz
    idx = np.searchsorted(tops, frac.ravel()).reshape(frac.shape)
    idx = np.clip(idx, 0, nlay - 1)
    v = vlay[idx]
    # compaction gradient inside layers and mild lateral variation
    v = v + 350.0 * (frac - tops[idx]) + 60.0 * np.sin(2 * np.pi * x / (0.37 * Lx) + 1.3) * (frac > 0.1)
    # high-velocity bodies (salt/carbonate lenses)
    for _ in range(2):
        xc, zc = rng.uniform(0.2, 0.8) * Lx, rng.uniform(0.45, 0.8) * Lz
        ax, az = rng.uniform(0.05, 0.12) * Lx, rng.uniform(0.02, 0.05) * Lz
        inside = ((x - xc) / ax) ** 2 + ((z - zc) / az) ** 2 < 1.0
        v = np.where(inside, np.maximum(v, rng.uniform(4600.0, 5400.0)), v)
    # water layer
    v = np.where(frac < 0.05, vmin, v)
    v = np.clip(v, vmin, vmax)
    return box_smooth(v, 1)
