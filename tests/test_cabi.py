"""CPU: the C-ABI shared library builds for gfx950, loads, and exports every symbol include/helm.h
declares.  No compute calls (there is no GPU here); the product path must fail loudly without one."""
import os
import re
import ctypes
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, 'include', 'helm.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(helm_[a-z_0-9]+)\s*\(', text)))


def test_library_exports_every_declared_symbol(helm_lib):
    from zephyr_amd import _lib
    names = header_symbols()
    assert len(names) >= 15
    for n in names:
        assert hasattr(helm_lib, n), 'libhelm.so does not export %s' % n
    assert sorted(_lib.exported_symbols()) == names           # the ctypes table covers the header exactly
    assert b'gfx950' in helm_lib.helm_version()


def test_struct_layouts_match_header():
    from zephyr_amd import _lib
    assert ctypes.sizeof(_lib.SolveOpts) == 32
    assert ctypes.sizeof(_lib.SolveInfo) == 24
    assert ctypes.sizeof(_lib.Timing) == 104      # 11 + 2 (gemm_bytes, gemm_sol_ms) eight-byte fields
    text = open(os.path.join(ROOT, 'include', 'helm.h')).read()
    body = re.sub(r'/\*.*?\*/', '', text[text.index('typedef struct helm_runtime_stats {'):text.index('} helm_runtime_stats;')], flags=re.S)
    names = [n.strip() for decl in re.findall(r'(?:long long|double)\s+([^;]+);', body) for n in decl.split(',')]
    assert names == [n for n, _ in _lib.RuntimeStats._fields_] and ctypes.sizeof(_lib.RuntimeStats) == 8 * len(names)


def test_every_kernel_registers_at_load(helm_lib):
    """Every kernel instantiation with a launch site registers its host handle while the library is loaded (no GPU needed): helm_warm() resolves
    that list on a device, so no kernel meets the runtime's lazy symbol lookup inside a solve."""
    from zephyr_amd import _lib
    st = _lib.runtime_stats()
    assert st['kernels_registered'] >= 200
    assert st['first_launches'] == 0 and st['dev_allocs'] == 0
    src = os.path.join(ROOT, 'zephyr_amd', 'csrc')
    for f in os.listdir(src):                       # no launch site bypasses the registry
        if f.endswith(('.hip', '.hpp')):
            text = open(os.path.join(src, f)).read()
            stray = [l for l in text.splitlines() if re.search(r'\bhipLaunchKernelGGL\(|<<<', l) and 'define' not in l and not l.lstrip().startswith(('//', 'else hipLaunchKernelGGL', 'if (tl_ev0)'))]
            assert not stray, (f, stray[:2])


def test_tuning_struct_round_trip(helm_lib, monkeypatch):
    """helm_tuning (include/helm.h) <-> _lib.Tuning: the same field order, every field reachable through its environment variable (read when
    used), helm_set_tuning replaces the lot and NULL restores defaults + environment.  Host-only: no GPU call."""
    from zephyr_amd import _lib
    text = open(os.path.join(ROOT, 'include', 'helm.h')).read()
    body = text[text.index('typedef struct helm_tuning {'):text.index('} helm_tuning;')]
    fields = re.findall(r'^\s*(int|double)\s+(\w+);', body, flags=re.M)
    assert [n for _, n in fields] == [n for n, _ in _lib.Tuning._fields_]
    assert [t for t, _ in fields] == ['int' if c is ctypes.c_int else 'double' for _, c in _lib.Tuning._fields_]
    for k in list(os.environ):
        if k.startswith('HELM_'):
            monkeypatch.delenv(k)
    t = _lib.tuning()
    assert (t.nd_leaf, t.nd_ws_gb, t.nd_sparse_rhs, t.nd_gjstep_min, t.nd_plans, t.ws_slots, t.mg3_keep_levels) == (8, 32.0, 1, 128, 6, 3, -1)
    assert t.mg3_omega == 0.9 and t.nd_stable_safety == 8.0 and t.sync_spin_ms == 0.0      # first and last doubles: the layouts agree end to end
    monkeypatch.setenv('HELM_ND_LEAF', '6')
    monkeypatch.setenv('HELM_MG3_OMEGA', '0.7')
    t = _lib.tuning()
    assert t.nd_leaf == 6 and t.mg3_omega == 0.7
    t.nd_leaf = 5; t.nd_sparse_rhs = 0
    _lib.set_tuning(t)
    monkeypatch.setenv('HELM_ND_LEAF', '7')                                  # a set structure wins over the environment
    u = _lib.tuning()
    assert u.nd_leaf == 5 and u.nd_sparse_rhs == 0 and u.mg3_omega == 0.7
    _lib.set_tuning(None)
    assert _lib.tuning().nd_leaf == 7
    # a C caller that zero-initialises the structure and sets one field gets the same limits the environment path applies
    z = _lib.Tuning()
    z.nd_sparse_rhs = 1
    _lib.set_tuning(z)
    u = _lib.tuning()
    assert (u.nd_leaf, u.nd_plans, u.ws_slots, u.pf_prio) == (2, 1, 1, 0) and u.nd_ws_gb == 32.0 and u.nd_stable_safety == 1.0 and u.mg3_omega == 0.9
    _lib.set_tuning(None)


def test_code_object_is_gfx950():
    so = os.path.join(ROOT, 'zephyr_amd', 'libhelm.so')
    import __graft_entry__ as g
    g.build()
    blob = open(so, 'rb').read()
    assert b'gfx950' in blob and b'k_stencil' in blob


def gpu_present(lib):
    return lib.helm_device_count() > 0


def test_fails_loudly_without_gpu(helm_lib):
    import zephyr_amd as za
    if gpu_present(helm_lib):
        pytest.skip('a GPU is present')
    op = za.MiniZephyr(dict(nx=16, nz=16, c=2000., freq=5.))
    with pytest.raises(Exception) as ei:
        op * np.ones(256, complex)
    assert 'no CPU' in str(ei.value) or 'HELM_ERR_DEVICE' in str(ei.value)


def test_product_never_imports_oracle():
    """The product path may cite the oracle in comments, but must never import, include, open or execute it."""
    pkg = os.path.join(ROOT, 'zephyr_amd')
    bad_py = re.compile(r'^\s*(from\s+oracle|import\s+oracle|from\s+\.+\s*oracle)|__import__\(.oracle|importlib.*oracle|open\([^)]*oracle|sys\.path.*oracle', re.M)
    bad_c = re.compile(r'#\s*include[^\n]*oracle|dlopen[^\n]*oracle|fopen[^\n]*oracle')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            path = os.path.join(dirpath, f)
            if f.endswith('.py'):
                assert not bad_py.search(open(path).read()), '%s imports the oracle' % f
            elif f.endswith(('.hip', '.hpp', '.cpp', '.h')):
                assert not bad_c.search(open(path).read()), '%s includes the oracle' % f
    # bench.py may use the oracle only inside its cpu_baseline leg
    text = open(os.path.join(ROOT, 'bench.py')).read()
    allowed = {'cpu_baseline', 'cpu_baseline_2n', 'cpu_baseline_pool', '_pool_worker'}
    uses = [m.start() for m in re.finditer(r'from oracle|import oracle', text)]
    assert uses
    for pos in uses:
        defs = list(re.finditer(r'^def (\w+)', text[:pos], re.M))
        assert defs and defs[-1].group(1) in allowed, 'bench.py uses the oracle outside the CPU baseline leg (in %s)' % (defs[-1].group(1) if defs else '?')
