#!/usr/bin/env python3
"""Copy a finished gpurun_out/<dir> of tools/run_profiles_r5.sh into profiles/ (r05_* names, the git hash of the collection stamped into every JSON / text
file) and regenerate profiles/README.md from the numbers in those files.

    python tools/update_profiles_r5.py gpurun_out/r5p
"""
import csv, glob, json, os, re, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
S = sys.argv[1]
P = os.path.join(ROOT, 'profiles') + os.sep
HASH = open(os.path.join(S, 'githash.txt')).read().strip() if os.path.exists(os.path.join(S, 'githash.txt')) else 'unknown'


def last_json(path):
    return json.loads(open(path).read().strip().splitlines()[-1])


def put_json(obj, name):
    obj = dict(obj); obj['collected_at_git'] = HASH
    json.dump(obj, open(P + name, 'w'), indent=1)


def find(pattern):
    fs = glob.glob(os.path.join(S, pattern), recursive=True)
    return fs[0] if fs else None


def put_text(src, name, header):
    open(P + name, 'w').write('# %s   (collected at git %s)\n' % (header, HASH) + open(os.path.join(S, src)).read())


def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('HIP_vector_type<double, 2u>', 'cplx')
    m = re.match(r'(void )?([\w:]+(<[^(]*>)?)', n)
    return m.group(2) if m else n[:40]


def per_item(path):
    """kernel family -> (launches per item, ms per item) of a --kernel-trace --stats csv; an item = one residual launch"""
    rows = list(csv.DictReader(open(path)))
    items = sum(int(r['Calls']) for r in rows if 'k_resid_nm' in r['Name']) or 1
    fam = {}
    for r in rows:
        n = short(r['Name'])
        key = 'k_zgemm3*' if n.startswith('k_zgemm3') else n.split('<')[0]
        c, t = fam.get(key, (0, 0.0))
        fam[key] = (c + int(r['Calls']), t + float(r['TotalDurationNs']) / 1e6)
    return items, {k: (c / items, t / items) for k, (c, t) in fam.items()}, rows


d = last_json(os.path.join(S, 'bench_n1.json')); put_json(d, 'r05_bench_n1.json')
dd = last_json(os.path.join(S, 'bench_driver.json')); put_json(dd, 'r05_bench_driver_cmd.json')
dp = last_json(os.path.join(S, 'bench_pipelined_under_rocprof.json')); put_json(dp, 'r05_bench_pipelined_under_rocprofv3.json')
ds = last_json(os.path.join(S, 'bench_serial_under_rocprof.json')); put_json(ds, 'r05_bench_serial_under_rocprofv3.json')
for src, dst in (('stats_pipe/**/s_kernel_stats.csv', 'r05_bench_pipelined_rocprofv3_kernel_stats.csv'), ('stats_serial/**/s_kernel_stats.csv', 'r05_bench_serial_rocprofv3_kernel_stats.csv'),
                 ('stats_serial_sparse/**/s_kernel_stats.csv', 'r05_bench_serial_sparse_rhs_rocprofv3_kernel_stats.csv'), ('stats3d/**/s_kernel_stats.csv', 'r05_config5_rocprofv3_kernel_stats.csv')):
    f = find(src)
    if f:
        shutil.copyfile(f, P + dst)
for a, b in (('pmc_traffic_zgemm.json', 'r05_pmc_traffic_zgemm.json'), ('pmc_traffic_resid.json', 'r05_pmc_traffic_resid_nm.json'),
             ('pmc_traffic_stencil_micro.json', 'r05_pmc_traffic_stencil_apply.json'), ('pmc_traffic_stencil3.json', 'r05_pmc_traffic_stencil3_apply.json')):
    if os.path.exists(os.path.join(S, a)):
        put_json(json.load(open(os.path.join(S, a))), b)
for a, b in (('apply3d_B16.json', 'r05_apply3d_B16_on_the_fly.json'), ('apply3d_B16_planes.json', 'r05_apply3d_B16_stored_planes.json')):
    if os.path.exists(os.path.join(S, a)) and open(os.path.join(S, a)).read().strip():
        put_json(last_json(os.path.join(S, a)), b)
open(P + 'r05_direct_per_level_trace.txt', 'w').write('# HELM_ND_TRACE=1 python tools/bench_direct.py --freqs 5.5   (collected at git %s)\n' % HASH +
                                                       ''.join(l for l in open(os.path.join(S, 'trace.txt')) if l.startswith('[nd trace]')) +
                                                       '# the same with HELM_ND_SPARSE_RHS=0 (every front of the forward pass, every row of the leaf back substitution)\n' +
                                                       ''.join(l for l in open(os.path.join(S, 'trace_every_front.txt')) if l.startswith('[nd trace]')))
put_text('gemm_log.txt', 'r05_gemm_log_by_shape.txt', 'HELM_GEMM_LOG=1 python tools/bench_direct.py --freqs 5.5 | tools/gemm_log.py: every product of one factorisation + three passes by shape')
put_text('zgemm_lab.txt', 'r05_zgemm_lab.txt', 'python tools/zgemm_lab.py -1,160: the tile the library chooses against the 128 x 64 tile (measured and not adopted)')
open(P + 'r05_githash.txt', 'w').write(HASH + '\n')
if os.path.exists(os.path.join(S, 'bench3d_under_rocprof.txt')):
    try:
        put_json(last_json(os.path.join(S, 'bench3d_under_rocprof.txt')), 'r05_config5_5hz_under_rocprofv3.json')
    except Exception:
        pass

pz = json.load(open(P + 'r05_pmc_traffic_zgemm.json')); pr = json.load(open(P + 'r05_pmc_traffic_resid_nm.json'))
p3 = json.load(open(P + 'r05_pmc_traffic_stencil3_apply.json')) if os.path.exists(P + 'r05_pmc_traffic_stencil3_apply.json') else None
a3 = json.load(open(P + 'r05_apply3d_B16_on_the_fly.json')) if os.path.exists(P + 'r05_apply3d_B16_on_the_fly.json') else None
a3p = json.load(open(P + 'r05_apply3d_B16_stored_planes.json')) if os.path.exists(P + 'r05_apply3d_B16_stored_planes.json') else None
it4, f4, _ = per_item(P + 'r04_bench_serial_sparse_rhs_rocprofv3_kernel_stats.csv')
it5, f5, rows_ss = per_item(P + 'r05_bench_serial_sparse_rhs_rocprofv3_kernel_stats.csv')
_, f5s, rows_s = per_item(P + 'r05_bench_serial_rocprofv3_kernel_stats.csv')
_, f5p, rows_p = per_item(P + 'r05_bench_pipelined_rocprofv3_kernel_stats.csv')
rows3 = list(csv.DictReader(open(P + 'r05_config5_rocprofv3_kernel_stats.csv'))) if os.path.exists(P + 'r05_config5_rocprofv3_kernel_stats.csv') else []
keys = sorted(set(f4) | set(f5), key=lambda k: -(f5.get(k, (0, 0))[1] + f4.get(k, (0, 0))[1]))
keys = [k for k in keys if max(f4.get(k, (0, 0))[1], f5.get(k, (0, 0))[1]) >= 0.03][:22]
tab = '\n'.join('| `%s` | %.1f | %.2f | %.1f | %.2f |' % (k, f4.get(k, (0, 0))[0], f4.get(k, (0, 0))[1], f5.get(k, (0, 0))[0], f5.get(k, (0, 0))[1]) for k in keys)
tot4 = sum(t for _, t in f4.values()); tot5 = sum(t for _, t in f5.values())


def table_of(rs, k):
    tot = sum(int(r['TotalDurationNs']) for r in rs)
    return '\n'.join('| `%s` | %s | %.1f | %.1f |' % (short(r['Name']), r['Calls'], float(r['AverageNs']) / 1e3, 100.0 * int(r['TotalDurationNs']) / tot) for r in rs[:k])


gm = [r for r in rows_s if 'k_zgemm3' in r['Name'] or 'k_gj_step' in r['Name']]
gcalls = sum(int(r['Calls']) for r in gm); gms = sum(float(r['TotalDurationNs']) for r in gm) / 1e6
Rd = dd['roofline']; Rs = ds['roofline']; R = d['roofline']; St = d.get('stencil_roofline', {})
cfg = d['config']; cfd = dd['config']
c5 = d.get('config5') if isinstance(d.get('config5'), dict) else {}
cb = d.get('cpu_baseline') if isinstance(d.get('cpu_baseline'), dict) else {}
pmc_items = 2.0
traffic_item = pz['traffic_bytes_per_launch'] * pz['launches_fetch_pass'] / 1e9 / pmc_items
oper_item = Rd['two_roofs']['operand_GB_per_item']
apply_line = ''
if a3:
    apply_line = ('27-point apply, 256 x 256 x 128, B = 16 (`tools/apply3d_micro.py`): coefficients on the fly **%.0f us** = %.3f of 8 TB/s by SURVEY 8(d)\'s N (32 B + 432)'
                  % (a3['us'], a3['frac_of_8TBps']))
    if a3p:
        apply_line += '; the same kernel reading its 27 stored planes: %.0f us = %.3f' % (a3p['us'], a3p['frac_of_8TBps'])
    if p3:
        apply_line += ('; PMC: %.2f GB per launch moved (FETCH_SIZE x 2 + WRITE_SIZE, two separate passes) = %.2f x the formula\'s %.2f GB (what the on-the-fly launch has to move, N (32 B + 24), is %.2f GB)'
                       % (p3['traffic_bytes_per_launch'] / 1e9, p3['traffic_bytes_per_launch'] / a3['algorithmic_bytes_per_launch'], a3['algorithmic_bytes_per_launch'] / 1e9,
                          8388608 * (32 * 16 + 24) / 1e9))
text = f'''# profiles/ -- round 5 (MI355X, 1 GPU; collected at git `{HASH}`)

Collected by `tools/run_profiles_r5.sh` on the GPU box (one `gpurun` call) and summarised by `tools/update_profiles_r5.py`, which stamps the git hash of the
collection into every JSON / text file (`collected_at_git`, `r05_githash.txt`).  Earlier rounds' files (`r04_*` ... `r01_*`) are kept for the before / after
comparison; their descriptions are in the git history of this file.

## The bench job: 1024 x 1024 Eurus, 16 frequencies x 256 sources (work item = create + assemble + factor one frequency + solve 256 sources to relres <= 1e-10)

| file | what |
|---|---|
| `r05_bench_driver_cmd.json` | the driver's command line, `python bench.py --steps 20 --warmup 5 --no-cpu`: **{dd['value']:.0f} wavefields/s**, {dd['ms_per_step']:.2f} ms per item (round 4: 13 911 / 18.40; {cfd.get('unprofiled_wfs', 0):.0f} with the per-launch events off; `dense_rhs_wfs` {cfd.get('dense_rhs_wfs', 0):.0f}: nothing skipped on the point sources; `support_declared_wfs` {cfd.get('support_declared_wfs', 0):.0f}; `strong_job_wfs` {cfd.get('strong_job_wfs', 0):.0f}: the whole 4096-wavefield job once in {cfd.get('strong_job_s', 0):.3f} s).  `roofline` (separate serial pass, every booked flop executed): all `k_zgemm3` + `k_gj_step` launches {Rd['achieved']:.1f} TFLOP/s = **{Rd['frac']:.3f}** of 78.6 ({Rd['launches_timed']} launches, avg {Rd['avg_launch_us']:.0f} us); against both roofs per launch {Rd['two_roofs']['frac']:.3f}; residual kernel `stencil_frac` {cfd.get('stencil_frac', 0):.3f} of 8 TB/s (norm-only launch on the caller's wavefield array) |
| `r05_bench_n1.json` | `python bench.py` (default: {d['steps']} timed items after {d['warmup']} warm-up items, all legs): {d['value']:.0f} wavefields/s, {d['ms_per_step']:.2f} ms per item; `parity_vs_lu_max_rel` = {d.get('parity_vs_lu_max_rel') or float('nan'):.2e}; `host_api_wfs` {cfg.get('host_api_wfs') or float('nan'):.0f} (MultiFreq * q over PCIe).  **config 2** (512^2, 8 x 64): device-resident {cfg.get('c2_wfs_device') or float('nan'):.0f} wavefields/s ({cfg.get('c2_ms_per_item') or float('nan'):.2f} ms per frequency), host API {cfg.get('c2_wfs_host_api') or float('nan'):.0f}, products {cfg.get('c2_gemm_frac') or float('nan'):.3f} of the MFMA peak.  **config 4** (gradient step at 512^2, 8 x 64, 128 receivers): `dpred(m)` {cfg.get('c4_dpred_s') or float('nan'):.3f} s, `Jtvec(m, v)` {cfg.get('c4_jtvec_s') or float('nan'):.3f} s (round 1, the only earlier figures: 0.42 / 0.80 s).  **config 5**: {c5.get('job_seconds', float('nan')):.2f} s at rtol 1e-8 through the device pipeline, {c5.get('job_seconds_rtol1e10', float('nan')):.2f} s at 1e-10 (round 4: 2.15 / 2.98); 27-point apply inside that leg {', '.join('%.0f us = %.2f' % (a['us'], a['frac_of_peak']) for a in c5.get('apply', []))} at B = 1 / 4 / 8 / 16.  CPU leg on the GPU box's own host: 1 core, M1-only LU {cb.get('value', float('nan')):.2f} wavefields/s |
| `r05_bench_serial_rocprofv3_kernel_stats.csv`, `r05_bench_serial_under_rocprofv3.json` | `HELM_ND_SPARSE_RHS=0 rocprofv3 --kernel-trace --stats -- python3 bench.py ... --no-pipeline --steps 8 --warmup 3 --no-plain-pass`: the kernels with nothing else on the GPU and nothing skipped -- the run `roofline` must agree with.  `r05_bench_serial_sparse_rhs_rocprofv3_kernel_stats.csv`: the same with the skipping on (what a production item costs; the per-item table below) |
| `r05_bench_pipelined_rocprofv3_kernel_stats.csv`, `r05_bench_pipelined_under_rocprofv3.json` | the pipelined timed region under the profiler ({dp['value']:.0f} wavefields/s): durations stretched by the sharing |
| `r05_pmc_traffic_zgemm.json`, `r05_pmc_traffic_resid_nm.json`, `r05_pmc_traffic_stencil_apply.json`, `r05_pmc_traffic_stencil3_apply.json` | `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (two separate passes; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950), reduced per kernel by `tools/pmc_reduce.py`: `k_zgemm3` + `k_gj_step` {pz['traffic_bytes_per_launch'] / 1e6:.0f} MB per launch over {pz['launches_fetch_pass']} launches of 2 work items = {traffic_item:.1f} GB per item against {oper_item:.1f} GB of necessary operand bytes = **{traffic_item / oper_item:.2f} x**; the residual kernel {pr['traffic_bytes_per_launch'] / 1e9:.2f} GB per launch against {St.get('bytes_per_launch_algorithmic', 0) / 1e9:.2f} GB algorithmic (round 4, with the wavefield store: 13.95 against 13.04) |
| `r05_apply3d_B16_on_the_fly.json`, `r05_apply3d_B16_stored_planes.json` | {apply_line} |
| `r05_direct_per_level_trace.txt` | `HELM_ND_TRACE=1 python tools/bench_direct.py --freqs 5.5`: device milliseconds per tree level of the factorisation and of the forward / backward sweeps, with and without the sparse-right-hand-side skipping |
| `r05_gemm_log_by_shape.txt` | every product of one factorisation + three passes aggregated by shape and addressing mode: microseconds, TFLOP/s, operand GB/s, roofline microseconds |
| `r05_zgemm_lab.txt` | the tile-kernel lab on the shapes the 1024^2 plan issues: the tile the library chooses against the 128 x 64 tile (round 5: measured, not adopted) |
| `r05_pipeline_overlap.txt` | `tools/trace_overlap.py` on a kernel trace of the pipelined region: share of the wall time with no kernel / one stream / both streams busy, idle gaps |

Agreement check (serial run, nothing skipped): the profiler's total over all `k_zgemm3<...>` / `k_zgemm3_la` / `k_gj_step` launches is {gcalls} launches, {gms:.1f} ms, **{1e3 * gms / max(gcalls, 1):.1f} us** on average;
bench.py's HIP-event average in that run is {Rs['avg_launch_us']:.1f} us ({Rd['avg_launch_us']:.1f} us in `r05_bench_driver_cmd.json`).

## One work item, kernel by kernel: round 4 against round 5 (serial, sparse-right-hand-side skipping on: what a production item costs)

From `r04_bench_serial_sparse_rhs_rocprofv3_kernel_stats.csv` ({it4} items) and `r05_bench_serial_sparse_rhs_rocprofv3_kernel_stats.csv` ({it5} items); an item = one residual launch.

| kernel family | r4 launches / item | r4 ms / item | r5 launches / item | r5 ms / item |
|---|---|---|---|---|
{tab}
| **all kernels** | | **{tot4:.2f}** | | **{tot5:.2f}** |

What moved: the residual launch no longer stores the wavefield (the back substitution writes the caller's array: `k_resid_nm_lds`); fronts of 128 and 256 separator
unknowns take the one-launch block step (`k_gj_panel` -> `k_gj_step`); M2 .. M4 of the isotropic Eurus operator are not assembled for N-row right-hand sides
(`k_assemble_eurus`); `k_lu_solve` keeps four factor rows in flight.

Kernel time of the serial profiled run (nothing skipped), top rows:

| kernel | calls | avg us | % of GPU time |
|---|---|---|---|
{table_of(rows_s, 18)}

The pipelined run (the factorisation of item k+1 beside the solve of item k; skipping on):

| kernel | calls | avg us | % of GPU time |
|---|---|---|---|
{table_of(rows_p, 12)}

## Config 5: 3-D 27-point, 256 x 256 x 128, 5 Hz x 16 sources under the profiler (`r05_config5_rocprofv3_kernel_stats.csv`)

| kernel | calls | avg us | % of GPU time |
|---|---|---|---|
{table_of(rows3, 10)}
'''
open(P + 'README.md', 'w').write(text)
print('profiles/README.md regenerated for round 5 at git', HASH)
