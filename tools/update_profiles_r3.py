#!/usr/bin/env python3
"""Copy a finished gpurun_out/<dir> of tools/run_profiles_r3.sh into profiles/ (r03_* names, the git hash of the collection stamped into
every JSON / text file) and regenerate profiles/README.md from the numbers in those files, so the README never drifts from the artefacts.

    python tools/update_profiles_r3.py gpurun_out/r3p2
"""
import csv, glob, json, os, re, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
S = sys.argv[1]
P = os.path.join(ROOT, 'profiles') + os.sep
HASH = open(os.path.join(S, 'githash.txt')).read().strip() if os.path.exists(os.path.join(S, 'githash.txt')) else 'unknown'


def last_json(path):
    return json.loads(open(path).read().strip().splitlines()[-1])


def put_json(obj, name):
    obj = dict(obj)
    obj['collected_at_git'] = HASH
    json.dump(obj, open(P + name, 'w'), indent=1)


def find(pattern):
    fs = glob.glob(os.path.join(S, pattern), recursive=True)
    return fs[0] if fs else None


d = last_json(os.path.join(S, 'bench_n1.json')); put_json(d, 'r03_bench_n1.json')
dd = last_json(os.path.join(S, 'bench_driver.json')); put_json(dd, 'r03_bench_driver_cmd.json')
dp = last_json(os.path.join(S, 'bench_pipelined_under_rocprof.json')); put_json(dp, 'r03_bench_pipelined_under_rocprofv3.json')
ds = last_json(os.path.join(S, 'bench_serial_under_rocprof.json')); put_json(ds, 'r03_bench_serial_under_rocprofv3.json')
shutil.copyfile(find('stats_pipe/**/s_kernel_stats.csv'), P + 'r03_bench_pipelined_rocprofv3_kernel_stats.csv')
shutil.copyfile(find('stats_serial/**/s_kernel_stats.csv'), P + 'r03_bench_serial_rocprofv3_kernel_stats.csv')
shutil.copyfile(find('stats3d/**/s_kernel_stats.csv'), P + 'r03_config5_rocprofv3_kernel_stats.csv')
for a, b in (('pmc_traffic_zgemm.json', 'r03_pmc_traffic_zgemm.json'), ('pmc_traffic_resid.json', 'r03_pmc_traffic_resid_nm.json'),
             ('pmc_traffic_stencil_micro.json', 'r03_pmc_traffic_stencil_apply.json')):
    put_json(json.load(open(os.path.join(S, a))), b)
open(P + 'r03_direct_per_level_trace.txt', 'w').write('# HELM_ND_TRACE=1 python tools/bench_direct.py --freqs 5.5   (collected at git %s)\n' % HASH +
                                                       ''.join(l for l in open(os.path.join(S, 'trace.txt')) if l.startswith('[nd trace]')))
open(P + 'r03_githash.txt', 'w').write(HASH + '\n')
b3 = last_json(os.path.join(S, 'bench3d_under_rocprof.txt')); put_json(b3, 'r03_config5_5hz_under_rocprofv3.json')

pz = json.load(open(P + 'r03_pmc_traffic_zgemm.json')); pr = json.load(open(P + 'r03_pmc_traffic_resid_nm.json')); ps = json.load(open(P + 'r03_pmc_traffic_stencil_apply.json'))
rows_s = list(csv.DictReader(open(P + 'r03_bench_serial_rocprofv3_kernel_stats.csv')))
rows_p = list(csv.DictReader(open(P + 'r03_bench_pipelined_rocprofv3_kernel_stats.csv')))
rows3 = list(csv.DictReader(open(P + 'r03_config5_rocprofv3_kernel_stats.csv')))


def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('HIP_vector_type<double, 2u>', 'cplx')
    m = re.match(r'(void )?([\w:]+(<[^(]*>)?)', n)
    return m.group(2) if m else n[:40]


def table_of(rs, k):
    tot = sum(int(r['TotalDurationNs']) for r in rs)
    return '\n'.join('| `%s` | %s | %.1f | %.1f |' % (short(r['Name']), r['Calls'], float(r['AverageNs']) / 1e3, 100.0 * int(r['TotalDurationNs']) / tot) for r in rs[:k])


def gemm_avg(rs):
    g = [r for r in rs if 'k_zgemm2' in r['Name']]
    gc = sum(int(r['Calls']) for r in g); gn = sum(int(r['TotalDurationNs']) for r in g)
    return gc, gn, (gn / gc / 1e3 if gc else 0.0), 100.0 * gn / sum(int(r['TotalDurationNs']) for r in rs)


def resid_avg(rs):
    g = [r for r in rs if 'k_resid_nm' in r['Name']]
    gc = sum(int(r['Calls']) for r in g); gn = sum(int(r['TotalDurationNs']) for r in g)
    return gc, (gn / gc / 1e3 if gc else 0.0)


R = d['roofline']; St = d['stencil_roofline']; cb = d['cpu_baseline']; c2 = d.get('cpu_baseline_2n', {}); cpool = d.get('cpu_baseline_pool', {})
Rd = dd['roofline']; Rs = ds['roofline']
c5 = d['config5']; ha = d['value_host_api']
gc_s, gn_s, gavg_s, gpct_s = gemm_avg(rows_s)
rc_s, ravg_s = resid_avg(rows_s)
nB = d['config']['sources_per_step']; N = d['config']['grid'][0] * d['config']['grid'][1]
alg_resid = St['bytes_per_launch_algorithmic']
text = f'''# profiles/ -- round 3 (MI355X, 1 GPU; collected at git `{HASH}`)

Collected by `tools/run_profiles_r3.sh` on the GPU box (one `gpurun` call) and summarised by `tools/update_profiles_r3.py`, which stamps the git
hash of the collection into every JSON / text file (`collected_at_git`, `r03_githash.txt`).  Round-2 (`r02_*`) and round-1 (`r01_*`) files are
kept for the before / after comparison; their description is in the git history of this file.

## The bench job: 1024 x 1024 Eurus, 16 frequencies x 256 sources (work item = create + assemble + factor one frequency + solve 256 sources to relres <= 1e-10)

| file | what |
|---|---|
| `r03_bench_n1.json` | `python bench.py` (default: {d['steps']} timed items after {d['warmup']} warm-up items, frequencies {', '.join('%g' % f for f in d['config']['freqs_hz_this_run'])} Hz): **{d['value']:.0f} wavefields/s**, {d['ms_per_step']:.1f} ms per item through the device pipeline with the per-launch HIP events on ({d['unprofiled']['value']:.0f} with them off); passes per wavefield {d['config']['solves_or_iterations_per_rhs_mean']:.2f}; `parity_vs_lu_max_rel` = {d['parity_vs_lu_max_rel']:.2e} (8 sources at 6 Hz against the SuperLU wavefields of the CPU leg).  `roofline` (kernel pass over the same items with nothing else on the GPU): all `k_zgemm2` launches, {R['achieved']:.1f} TFLOP/s = **{100 * R['frac']:.0f} %** of the 78.6 TFLOP/s nominal fp64 peak ({R['launches_timed']} launches, avg {R['avg_launch_us']:.0f} us; inside the pipelined region, where two streams share the CUs: {100 * R['in_pipeline']['frac']:.0f} %).  `stencil_roofline` / `roofline_northstar`: the node-major residual launches (`k_resid_nm_lds`), {St['achieved']:.0f} GB/s = **{100 * St['frac']:.0f} %** of 8 TB/s on N((32 + 16)B + 144) = {alg_resid / 1e9:.2f} GB (x and q in, the wavefield out), avg {St['avg_launch_us']:.0f} us; rhs-major apply microbenchmark {', '.join('%.0f' % (100 * m['frac_of_peak']) for m in St['apply_microbench'])} % at B = 1 / 8 / 32 / 64.  `value_host_api`: the whole job through `MultiFreq * q` with scipy-sparse sources in and numpy wavefields out, **{ha['value']:.0f} wavefields/s** ({ha['seconds']:.2f} s for 4096 wavefields, 4.3 GB per frequency over PCIe, {ha['workers_per_device']} workers per GPU; runs: {', '.join('%d worker(s): %.0f' % (r['workers_per_device'], r['value']) for r in d['value_host_api_runs'])}).  `config5`: {c5['job_seconds']:.2f} s for the 3-D job through the device pipeline (set-up of frequency k+1 beside the iterations of frequency k; {c5.get('job_seconds_one_after_the_other', c5['job_seconds']):.2f} s one frequency after the other: {', '.join('%g Hz %.2f s / %d its' % (p['freq_hz'], p['seconds'], max(p['iterations'])) for p in c5['per_frequency'])}; 27-point apply {', '.join('%.0f' % (100 * a['frac_of_peak']) for a in c5['apply'])} % of 8 TB/s at B = 1 / 4 / 8 / 16).  CPU legs on the GPU box's own host (256 logical CPUs): 1 core, M1-only LU {cb['value']:.2f} wavefields/s (assemble {cb['assemble_s']:.1f} s, factor {cb['factor_s']:.1f} s, {cb['per_rhs_s']:.3f} s per source); the faithful 2N x 2N system at 512^2: {c2.get('value', float('nan')):.2f}; 16 processes, one per frequency: {cpool.get('value', float('nan')):.1f} |
| `r03_bench_driver_cmd.json` | the driver's command line, `python bench.py --steps 20 --warmup 5 --no-cpu` (all 16 frequencies): **{dd['value']:.0f} wavefields/s**, {dd['ms_per_step']:.1f} ms per item ({dd['unprofiled']['value']:.0f} with the events off), passes per wavefield {dd['config']['solves_or_iterations_per_rhs_mean']:.2f}, `roofline.frac` {Rd['frac']:.3f}, `stencil_roofline.frac` {dd['stencil_roofline']['frac']:.3f} |
| `r03_bench_serial_rocprofv3_kernel_stats.csv`, `r03_bench_serial_under_rocprofv3.json` | `rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu --no-config5 --no-host-api --no-pipeline --steps 8 --warmup 2 --no-plain-pass`: the kernels with nothing else on the GPU -- the run `roofline` must agree with |
| `r03_bench_pipelined_rocprofv3_kernel_stats.csv`, `r03_bench_pipelined_under_rocprofv3.json` | the same command without `--no-pipeline` ({dp['value']:.0f} wavefields/s under the profiler): durations stretched by the sharing |
| `r03_pmc_traffic_zgemm.json`, `r03_pmc_traffic_resid_nm.json`, `r03_pmc_traffic_stencil_apply.json` | `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (two separate passes) of one serial work item, reduced per kernel by `tools/pmc_reduce.py` (FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950): `k_zgemm2` {pz['traffic_bytes_per_launch'] / 1e6:.0f} MB per launch over {pz['launches_fetch_pass']} launches; the residual kernel {pr['traffic_bytes_per_launch'] / 1e9:.2f} GB per launch against {alg_resid / 1e9:.2f} GB algorithmic (the difference is the vertical halo rows of x); the rhs-major apply of the microbenchmark {ps['traffic_bytes_per_launch'] / 1e9:.2f} GB per launch averaged over B = 1, 8, 32, 64 (algorithmic average 0.95 GB) |
| `r03_direct_per_level_trace.txt` | `HELM_ND_TRACE=1 python tools/bench_direct.py --freqs 5.5`: device milliseconds per tree level of the factorisation and of the forward / backward sweeps |
| `r03_fp64_clock_probe.txt` | `tools/fp64_clock.hip`: shader clock and fp64 issue rate measured INSIDE long probe kernels (clock64 against the 100-MHz wall clock) plus `rocm-smi` during the run: the clock holds 2.1-2.4 GHz under fp64 load (665 W), so the ~48-57 TFLOP/s of the tile kernel's instruction mix is an issue limit, not throttling |

Agreement check (serial run): the profiler's average over all `k_zgemm2<...>` instantiations is {gc_s} launches, {gn_s / 1e6:.1f} ms, **{gavg_s:.1f} us**
({gpct_s:.0f} % of the GPU time); bench.py's HIP-event average in that run is {Rs['avg_launch_us']:.1f} us ({R['avg_launch_us']:.1f} us in `r03_bench_n1.json`).  Residual kernel: profiler
{ravg_s:.0f} us over {rc_s} launches, HIP events {ds['stencil_roofline']['avg_launch_us']:.0f} us.

Kernel time of the serial profiled run, top rows:

| kernel | calls | avg us | % of GPU time |
|---|---|---|---|
{table_of(rows_s, 18)}

The pipelined run (same items; the factorisation of item k+1 runs beside the solve of item k):

| kernel | calls | avg us | % of GPU time |
|---|---|---|---|
{table_of(rows_p, 10)}

(`k_zgemm2<TM, IDX, RN, KS, UNR, OCC>`: tile height, operand addressing 0 dense / 1 row table / 2 forward gather, columns per thread, K slab, k-loop unroll, waves per SIMD asked for;
`k_zgemm2_la` = blocked Gauss-Jordan update with the next pivot sweep riding along; `k_gj_panel` / `k_gj_slices` = pivot block inverse + row / column panels;
`k_nd_build_front` = stencil entries + both children's Schur complements gathered into a front in one pass; `k_resid_nm_lds` = node-major true residual with LDS-staged
coefficients, `||q||^2` and the wavefield store fused; `k_front_absmax` / `k_lu_factor*` / `k_lu_solve` = condition estimates and the pivoted-LU treatment of ill-conditioned fronts;
`k_stencil_t` = the rhs-major apply of the microbenchmark.)

## Config 5: 3-D 27-point, 256 x 256 x 128, 5 Hz x 16 sources under the profiler (`r03_config5_rocprofv3_kernel_stats.csv`, `r03_config5_5hz_under_rocprofv3.json`)

| kernel | calls | avg us | % of GPU time |
|---|---|---|---|
{table_of(rows3, 10)}

(`k_zgemm2<64, 0, 4, ..>` / `k_zgemm2_la` / `k_gj_*` / `k_nd_build_front`: the column dissection of the directly solved level -- the 2-D multifrontal solver over z-columns;
`k_zgemm2<128, 0, 2, ..>` (128 x 16 tile) / `<32, 0, 1, ..>` (the few big fronts at the top) / `k_nd_fwd_rows` / `k_nd_bwd_*`: its forward / backward passes for 16 right-hand sides inside the cycles; `k_stencil3<false, EPI>`: 4 = residual,
5 = l1-Jacobi sweep, 1 / 6 = the outer BiCGSTAB applies with fused dots.)

The whole 4-frequency job is the `config5` block of `r03_bench_n1.json` (above).  Round-2 files `r02_config5_*` (tool runs, standard-cycle comparison) are kept.
'''
open(P + 'README.md', 'w').write(text)
print('profiles/README.md regenerated for round 3 at git', HASH)
