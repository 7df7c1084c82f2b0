#!/usr/bin/env python3
"""Copy a finished gpurun_out/<dir> of tools/run_profiles_r6.sh into profiles/ (r06_* names, the git hash of the collection stamped into every JSON / text
file) and regenerate profiles/README.md from the numbers in those files.

    python tools/update_profiles_r6.py gpurun_out/r6prof [gpurun_out/<other lease>/run1.json ...]      (further arguments: headline runs of other leases)
"""
import csv, glob, json, os, re, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
S = sys.argv[1]
OTHER = sys.argv[2:]
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
    if os.path.exists(os.path.join(S, src)):
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


d = last_json(os.path.join(S, 'bench_n1.json')); put_json(d, 'r06_bench_n1.json')
dd = last_json(os.path.join(S, 'bench_driver.json')); put_json(dd, 'r06_bench_driver_cmd.json')
dp = last_json(os.path.join(S, 'bench_pipelined_under_rocprof.json')); put_json(dp, 'r06_bench_pipelined_under_rocprofv3.json')
ds = last_json(os.path.join(S, 'bench_serial_under_rocprof.json')); put_json(ds, 'r06_bench_serial_under_rocprofv3.json')
dset = last_json(os.path.join(S, 'bench_sets_under_rocprof.json')); put_json(dset, 'r06_bench_sets_under_rocprofv3.json')
for src, dst in (('stats_pipe/**/s_kernel_stats.csv', 'r06_bench_pipelined_rocprofv3_kernel_stats.csv'), ('stats_serial/**/s_kernel_stats.csv', 'r06_bench_serial_rocprofv3_kernel_stats.csv'),
                 ('stats_serial_sparse/**/s_kernel_stats.csv', 'r06_bench_serial_sparse_rhs_rocprofv3_kernel_stats.csv'), ('stats_sets/**/s_kernel_stats.csv', 'r06_bench_sets_rocprofv3_kernel_stats.csv'),
                 ('stats3d/**/s_kernel_stats.csv', 'r06_config5_rocprofv3_kernel_stats.csv')):
    f = find(src)
    if f:
        shutil.copyfile(f, P + dst)
for a, b in (('pmc_traffic_zgemm.json', 'r06_pmc_traffic_zgemm.json'), ('pmc_traffic_resid.json', 'r06_pmc_traffic_resid_nm.json'),
             ('pmc_traffic_stencil_micro.json', 'r06_pmc_traffic_stencil_apply.json'), ('pmc_traffic_stencil3.json', 'r06_pmc_traffic_stencil3_apply.json')):
    if os.path.exists(os.path.join(S, a)):
        put_json(json.load(open(os.path.join(S, a))), b)
if os.path.exists(os.path.join(S, 'apply3d_B16.json')) and open(os.path.join(S, 'apply3d_B16.json')).read().strip():
    put_json(last_json(os.path.join(S, 'apply3d_B16.json')), 'r06_apply3d_B16_on_the_fly.json')
# per-level traces: WARM (third factorisation of the process) for one operator and for sets of two and four in the same launches; then one warm operator's passes
raw = open(os.path.join(S, 'factor_trace_raw.txt')).read() if os.path.exists(os.path.join(S, 'factor_trace_raw.txt')) else ''
warm = raw[raw.index('=== round 1'):] if '=== round 1' in raw else raw
tr = open(os.path.join(S, 'trace.txt')).read() if os.path.exists(os.path.join(S, 'trace.txt')) else ''
open(P + 'r06_direct_per_level_trace.txt', 'w').write(
    '# WARM per-level device time of the factorisation at 1024^2 (tools/factor_trace.py, HELM_ND_TRACE=1; second round of the process: every kernel resolved, pools filled):\n'
    '# one operator, then two and four operators in the same launches (helm_prefactor_many).   (collected at git %s)\n' % HASH +
    ''.join(l + '\n' for l in warm.splitlines() if l.startswith('===') or l.startswith('[nd trace]')) +
    '# HELM_ND_TRACE=1 python tools/bench_direct.py --freqs 5.5: the FIRST operator of a process (COLD: its factorisation includes first-use costs) and its passes\n' +
    ''.join(l for l in tr.splitlines(True) if l.startswith('[nd trace]')))
put_text('gemm_log.txt', 'r06_gemm_log_by_shape.txt', 'HELM_GEMM_LOG=1 python tools/bench_direct.py --freqs 5.5 | tools/gemm_log.py: every product of one factorisation + three passes by shape')
put_text('pipeline_overlap.txt', 'r06_pipeline_overlap.txt', 'tools/trace_overlap.py on the kernel trace of the pipelined timed region (sets of two factorisations)')
put_text('factor_many.txt', 'r06_factor_many.txt', 'tools/factor_many_probe.py 1024 5: milliseconds of 1 / 2 / 4 factorisations, one after the other against in the same launches')
put_text('c4_repeat.txt', 'r06_config4_repeat.txt', 'tools/c4_repeat.py 8 alternate: dpred(m) / Jtvec(m, v) of config 4 called back to back on alternating models, milliseconds per call')
open(P + 'r06_githash.txt', 'w').write(HASH + '\n')
if os.path.exists(os.path.join(S, 'bench3d_under_rocprof.txt')):
    try:
        put_json(last_json(os.path.join(S, 'bench3d_under_rocprof.txt')), 'r06_config5_5hz_under_rocprofv3.json')
    except Exception:
        pass

# headline reproducibility: the driver's command in fresh processes on fresh leases
runs = []
for tag, path in [('this lease, first GPU process of the box (all legs)', os.path.join(S, 'bench_driver.json'))] + [('this lease, fresh process %d' % i, os.path.join(S, 'headline_%d.json' % i)) for i in (1, 2, 3)] + \
        [('separate gpurun call: ' + os.path.relpath(p, ROOT), p) for p in OTHER]:
    if os.path.exists(path) and open(path).read().strip():
        try:
            r = last_json(path)
            c = r.get('config', {})
            runs.append({'where': tag, 'value': r['value'], 'ms_per_step': r['ms_per_step'], 'unprofiled_wfs': c.get('unprofiled_wfs'), 'timed_max_item_gap_ms': c.get('timed_max_item_gap_ms'),
                         'timed_dev_allocs': c.get('timed_dev_allocs'), 'timed_first_launches': c.get('timed_first_launches'), 'timed_cpu_throttled_ms': c.get('timed_cpu_throttled_ms')})
        except Exception:
            pass
vals = sorted(r['value'] for r in runs)
rep = {'command': 'python3 bench.py --gpus 1 --steps 20 --warmup 5 (other legs switched off in the fresh-process repeats)', 'runs': runs, 'n': len(vals),
       'min': vals[0] if vals else None, 'median': vals[len(vals) // 2] if vals else None, 'max': vals[-1] if vals else None,
       'spread_max_over_min': (vals[-1] / vals[0]) if vals else None}
put_json(rep, 'r06_headline_repro.json')

pz = json.load(open(P + 'r06_pmc_traffic_zgemm.json')); pr = json.load(open(P + 'r06_pmc_traffic_resid_nm.json'))
p3 = json.load(open(P + 'r06_pmc_traffic_stencil3_apply.json')) if os.path.exists(P + 'r06_pmc_traffic_stencil3_apply.json') else None
a3 = json.load(open(P + 'r06_apply3d_B16_on_the_fly.json')) if os.path.exists(P + 'r06_apply3d_B16_on_the_fly.json') else None
it5, f5, _ = per_item(P + 'r05_bench_serial_sparse_rhs_rocprofv3_kernel_stats.csv')
it6, f6, rows_ss = per_item(P + 'r06_bench_serial_sparse_rhs_rocprofv3_kernel_stats.csv')
_, f6s, rows_s = per_item(P + 'r06_bench_serial_rocprofv3_kernel_stats.csv')
_, f6p, rows_p = per_item(P + 'r06_bench_pipelined_rocprofv3_kernel_stats.csv')
its, f6t, rows_t = per_item(P + 'r06_bench_sets_rocprofv3_kernel_stats.csv')
rows3 = list(csv.DictReader(open(P + 'r06_config5_rocprofv3_kernel_stats.csv'))) if os.path.exists(P + 'r06_config5_rocprofv3_kernel_stats.csv') else []
keys = sorted(set(f5) | set(f6), key=lambda k: -(f6.get(k, (0, 0))[1] + f5.get(k, (0, 0))[1]))
keys = [k for k in keys if max(f5.get(k, (0, 0))[1], f6.get(k, (0, 0))[1]) >= 0.03][:24]
tab = '\n'.join('| `%s` | %.1f | %.2f | %.1f | %.2f | %.1f | %.2f |' % (k, f5.get(k, (0, 0))[0], f5.get(k, (0, 0))[1], f6.get(k, (0, 0))[0], f6.get(k, (0, 0))[1], f6t.get(k, (0, 0))[0], f6t.get(k, (0, 0))[1]) for k in keys)
tot5 = sum(t for _, t in f5.values()); tot6 = sum(t for _, t in f6.values()); tot6t = sum(t for _, t in f6t.values())


def table_of(rs, k):
    tot = sum(int(r['TotalDurationNs']) for r in rs)
    return '\n'.join('| `%s` | %s | %.1f | %.1f |' % (short(r['Name']), r['Calls'], float(r['AverageNs']) / 1e3, 100.0 * int(r['TotalDurationNs']) / tot) for r in rs[:k])


def gemm_of(rows):
    gm = [r for r in rows if 'k_zgemm3' in r['Name'] or 'k_gj_step' in r['Name'] or 'k_sep_bwd_small' in r['Name']]
    return sum(int(r['Calls']) for r in gm), sum(float(r['TotalDurationNs']) for r in gm) / 1e6


gcalls, gms = gemm_of(rows_s)
tcalls, tms = gemm_of(rows_t)
Rd = dd['roofline']; Rs = ds['roofline']; Rt = dset['roofline']; St = dd.get('stencil_roofline', {})
cfd = dd['config']; fl = dd.get('detail', {}).get('flat', {})
c5 = dd.get('config5') if isinstance(dd.get('config5'), dict) else {}
c4 = dd.get('config4') if isinstance(dd.get('config4'), dict) else {}
cb = dd.get('cpu_baseline') if isinstance(dd.get('cpu_baseline'), dict) else {}
pmc_items = 2.0
traffic_item = pz['traffic_bytes_per_launch'] * pz['launches_fetch_pass'] / 1e9 / pmc_items
oper_item = Rs['two_roofs']['operand_GB_per_item']
apply_line = ''
if a3:
    apply_line = ('27-point apply, 256 x 256 x 128, B = 16 (`tools/apply3d_micro.py`): coefficients on the fly **%.0f us** = %.3f of 8 TB/s by SURVEY 8(d)\'s N (32 B + 432)' % (a3['us'], a3['frac_of_8TBps']))
    if p3:
        apply_line += ('; PMC: %.2f GB per launch moved (FETCH_SIZE x 2 + WRITE_SIZE, two separate passes) against %.2f GB by the formula and %.2f GB an on-the-fly launch has to move'
                       % (p3['traffic_bytes_per_launch'] / 1e9, a3['algorithmic_bytes_per_launch'] / 1e9, 8388608 * (32 * 16 + 24) / 1e9))
rr = '\n'.join('| %s | %.0f | %.2f | %s | %s | %s | %s |' % (r['where'], r['value'], r['ms_per_step'], r.get('unprofiled_wfs'), r.get('timed_max_item_gap_ms'), r.get('timed_dev_allocs'), r.get('timed_cpu_throttled_ms')) for r in runs)
text = f'''# profiles/ -- round 6 (MI355X, 1 GPU; collected at git `{HASH}`)

Collected by `tools/run_profiles_r6.sh` on the GPU box (one `gpurun` call) and summarised by `tools/update_profiles_r6.py`, which stamps the git hash of the
collection into every JSON / text file (`collected_at_git`, `r06_githash.txt`).  Earlier rounds' files (`r05_*` ... `r01_*`) are kept for the before / after
comparison; their descriptions are in the git history of this file.  `r06_cpu_quota_stall.txt` and `r06_build_front_bisect.txt` are the evidence of the two
findings of the round (written from the `gpurun` outputs of the day, commands inside).

## Is the headline reproducible?  (`r06_headline_repro.json`)

The driver's command in fresh processes, on this lease and in separate gpurun calls of the day: **min {rep['min']:.0f} / median {rep['median']:.0f} / max {rep['max']:.0f} wavefields/s**
over {rep['n']} runs (max / min = {rep['spread_max_over_min']:.3f}).  Round 5's driver run measured 10 914 for a command that gave 13 950-14 630 here.

| run | wavefields/s | ms per step | `unprofiled_wfs` | longest item gap, ms | device allocations in the timed region | cgroup throttling in the timed region, ms |
|---|---|---|---|---|---|---|
{rr}

## The bench job: 1024 x 1024 Eurus, 16 frequencies x 256 sources (work item = create + assemble + factor one frequency + solve 256 sources to relres <= 1e-10)

| file | what |
|---|---|
| `r06_bench_driver_cmd.json` | the driver's exact command, `python3 bench.py --gpus 1 --steps 20 --warmup 5`, as the first GPU process of the box: **{dd['value']:.0f} wavefields/s**, {dd['ms_per_step']:.2f} ms per item (round 5's collection: 14 627 / 17.50; {cfd.get('unprofiled_wfs', 0):.0f} with the per-launch events off; `strong_job_wfs` {cfd.get('strong_job_wfs', 0):.0f}; `dense_rhs_wfs` {fl.get('dense_rhs_wfs') or 0:.0f}; `support_declared_wfs` {fl.get('support_declared_wfs') or 0:.0f}); timed region: longest item gap {cfd.get('timed_max_item_gap_ms')} ms (the first item: the pipeline fills), {cfd.get('timed_dev_allocs')} device allocations, {cfd.get('timed_first_launches')} kernels launched for the first time, {cfd.get('timed_cpu_throttled_ms')} ms of cgroup throttling (quota {cfd.get('cpu_quota_cores')} CPUs).  `roofline` (production launch sets with nothing beside them, every booked flop executed): `k_zgemm3` + `k_gj_step` {Rd['achieved']:.1f} TFLOP/s = **{Rd['frac']:.3f}** of 78.6 ({Rd['launches_timed']} launches, avg {Rd['avg_launch_us']:.0f} us); against both roofs per launch {Rd['two_roofs']['frac']:.3f}; residual kernel `stencil_frac` {cfd.get('stencil_frac', 0):.3f} of 8 TB/s.  **config 2** device-resident {cfd.get('c2_wfs_device') or 0:.0f} wavefields/s; **config 4** `dpred(m)` {c4.get('dpred_seconds', float('nan')):.3f} s, `Jtvec(m, v)` {c4.get('jtvec_seconds', float('nan')):.3f} s (medians of five, spreads {c4.get('dpred_spread', float('nan')):.2f} / {c4.get('jtvec_spread', float('nan')):.2f}; device spans {c4.get('gpu_ms', float('nan')):.0f} ms; round 5: 0.126-0.160 / 0.183-0.198 s); **config 5** {c5.get('job_seconds', float('nan')):.2f} s at rtol 1e-8, {c5.get('job_seconds_rtol1e10', float('nan')):.2f} s at 1e-10; `parity_vs_lu_max_rel` {dd.get('parity_vs_lu_max_rel') or float('nan'):.2e}; CPU leg on the GPU box's own host (1 core, M1-only LU): {cb.get('value', float('nan')):.2f} wavefields/s |
| `r06_bench_n1.json` | `python bench.py` (default: {d['steps']} timed items after {d['warmup']} warm-up items, all legs): {d['value']:.0f} wavefields/s, {d['ms_per_step']:.2f} ms per item |
| `r06_bench_sets_rocprofv3_kernel_stats.csv`, `r06_bench_sets_under_rocprofv3.json` | `HELM_ND_SPARSE_RHS=0 rocprofv3 --kernel-trace --stats -- python3 bench.py ... --steps 8 --warmup 4 --no-plain-pass`: includes the pass `roofline` quotes -- the production launches (the factorisations of two items in the same batched launches, helm_prefactor_many) with nothing beside them and nothing skipped |
| `r06_bench_serial_rocprofv3_kernel_stats.csv`, `r06_bench_serial_under_rocprofv3.json` | the same with `--no-pipeline`: every item factored by itself (rounds 1-5's launches), strictly one after the other.  `r06_bench_serial_sparse_rhs_rocprofv3_kernel_stats.csv`: that with the skipping on (what an item costs kernel by kernel; the table below) |
| `r06_bench_pipelined_rocprofv3_kernel_stats.csv`, `r06_bench_pipelined_under_rocprofv3.json`, `r06_pipeline_overlap.txt` | the pipelined timed region under the profiler ({dp['value']:.0f} wavefields/s): durations stretched by the sharing; share of the wall time with no kernel / one stream / both streams busy |
| `r06_pmc_traffic_zgemm.json`, `r06_pmc_traffic_resid_nm.json`, `r06_pmc_traffic_stencil_apply.json`, `r06_pmc_traffic_stencil3_apply.json` | `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (two separate passes; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950), reduced per kernel by `tools/pmc_reduce.py`: `k_zgemm3` + `k_gj_step` {pz['traffic_bytes_per_launch'] / 1e6:.0f} MB per launch over {pz['launches_fetch_pass']} launches of 2 work items = {traffic_item:.1f} GB per item against {oper_item:.1f} GB of necessary operand bytes = **{traffic_item / oper_item:.2f} x**; the residual kernel {pr['traffic_bytes_per_launch'] / 1e9:.2f} GB per launch against {St.get('bytes_per_launch_algorithmic', 0) / 1e9:.2f} GB algorithmic.  `bench.py` reads the newest round's files for `roofline.traffic` |
| `r06_direct_per_level_trace.txt` | WARM per-level device time of the factorisation (one operator; two and four in the same launches), then the cold first operator of a process with its passes -- labelled as such |
| `r06_factor_many.txt` | one / two / four factorisations: one after the other against in the same launches |
| `r06_gemm_log_by_shape.txt` | every product of one factorisation + three passes aggregated by shape and addressing mode |
| `r06_config4_repeat.txt` | config 4's `dpred(m)` / `Jtvec(m, v)` called back to back on alternating models |
| `r06_apply3d_B16_on_the_fly.json` | {apply_line} |

Agreement check: the profiler's total over all `k_zgemm3<...>` / `k_gj_step` / `k_sep_bwd_small` launches of the serial run (nothing skipped, every item by itself) is {gcalls} launches, {gms:.1f} ms, **{1e3 * gms / max(gcalls, 1):.1f} us** on average -- bench.py's HIP-event average in that run is {Rs['avg_launch_us']:.1f} us (`roofline.frac` {Rs['frac']:.3f});
in the run with the production launch sets: {tcalls} launches, {tms:.1f} ms, {1e3 * tms / max(tcalls, 1):.1f} us (timed region and roofline pass together; bench.py's average over the roofline pass {Rt['avg_launch_us']:.1f} us, `roofline.frac` {Rt['frac']:.3f}).

## One work item, kernel by kernel: round 5 against round 6 (serial, sparse-right-hand-side skipping on: what a production item costs)

From `r05_bench_serial_sparse_rhs_rocprofv3_kernel_stats.csv` ({it5} items), `r06_bench_serial_sparse_rhs_rocprofv3_kernel_stats.csv` ({it6} items; every item factored by itself) and
`r06_bench_sets_rocprofv3_kernel_stats.csv` ({its} items; two factorisations per set of launches, nothing skipped in its roofline pass); an item = one residual launch.

| kernel family | r5 launches / item | r5 ms / item | r6 launches / item | r6 ms / item | r6 (sets of two) launches / item | ms / item |
|---|---|---|---|---|---|---|
{tab}
| **all kernels** | | **{tot5:.2f}** | | **{tot6:.2f}** | | **{tot6t:.2f}** |

Kernel time of the serial profiled run (nothing skipped), top rows:

| kernel | calls | avg us | % of GPU time |
|---|---|---|---|
{table_of(rows_s, 18)}

The pipelined run (sets of two factorisations beside the solves of the items before them; skipping on):

| kernel | calls | avg us | % of GPU time |
|---|---|---|---|
{table_of(rows_p, 12)}

## Config 5: 3-D 27-point, 256 x 256 x 128, 5 Hz x 16 sources under the profiler (`r06_config5_rocprofv3_kernel_stats.csv`)

| kernel | calls | avg us | % of GPU time |
|---|---|---|---|
{table_of(rows3, 10)}
'''
open(P + 'README.md', 'w').write(text)
print('profiles/README.md regenerated for round 6 at git', HASH)
