#!/usr/bin/env python3
"""Copy a finished gpurun_out/<dir> of tools/run_profiles.sh into profiles/ (r02_* names) and regenerate profiles/README.md from the
numbers in those files, so the README never drifts from the artefacts.

    python tools/update_profiles.py gpurun_out/r02p2
"""
import csv, json, os, re, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
S = sys.argv[1]
P = os.path.join(ROOT, 'profiles') + os.sep


def cp(a, b):
    shutil.copyfile(os.path.join(S, a), P + b)


cp('bench_n1.json', 'r02_bench_n1.json')
cp('bench_under_rocprof.json', 'r02_bench_n1_under_rocprofv3.json')
cp('stats/s_kernel_stats.csv', 'r02_bench_n1_rocprofv3_kernel_stats.csv')
cp('pmc_traffic_zgemm.json', 'r02_pmc_traffic_zgemm.json')
cp('pmc_traffic_resid.json', 'r02_pmc_traffic_resid_nm.json')
cp('pmc_traffic_stencil_micro.json', 'r02_pmc_traffic_stencil_apply.json')
cp('bench3d.txt', 'r02_config5_bench3d.txt')
open(P + 'r02_config5_bench3d.json', 'w').write(open(os.path.join(S, 'bench3d.txt')).read().strip().split('\n')[-1] + '\n')
cp('stats3d/s_kernel_stats.csv', 'r02_config5_rocprofv3_kernel_stats.csv')
if os.path.exists(os.path.join(S, 'bench3d_standard.txt')):
    open(P + 'r02_config5_standard_cycle_4src.json', 'w').write(open(os.path.join(S, 'bench3d_standard.txt')).read().strip().split('\n')[-1] + '\n')
open(P + 'r02_config5_under_rocprofv3.json', 'w').write(open(os.path.join(S, 'bench3d_under_rocprof.txt')).read().strip().split('\n')[-1] + '\n')
if os.path.exists(os.path.join(S, 'trace.txt')):
    open(P + 'r02_direct_per_level_trace.txt', 'w').write(''.join(l for l in open(os.path.join(S, 'trace.txt')) if l.startswith('[nd trace]')))

d = json.load(open(P + 'r02_bench_n1.json')); d2 = json.load(open(P + 'r02_bench_n1_under_rocprofv3.json'))
pz = json.load(open(P + 'r02_pmc_traffic_zgemm.json')); pr = json.load(open(P + 'r02_pmc_traffic_resid_nm.json')); ps = json.load(open(P + 'r02_pmc_traffic_stencil_apply.json'))
rows = list(csv.DictReader(open(P + 'r02_bench_n1_rocprofv3_kernel_stats.csv')))
tot = sum(int(r['TotalDurationNs']) for r in rows)
g = [r for r in rows if 'k_zgemm2' in r['Name']]
gc = sum(int(r['Calls']) for r in g); gn = sum(int(r['TotalDurationNs']) for r in g)


def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('HIP_vector_type<double, 2u>', 'cplx')
    m = re.match(r'(void )?([\w:]+(<[^(]*>)?)', n)
    return m.group(2) if m else n[:40]


def table_of(rs, k):
    return '\n'.join('| `%s` | %s | %.1f | %s |' % (short(r['Name']), r['Calls'], float(r['AverageNs']) / 1e3, r['Percentage']) for r in rs[:k])


R = d['roofline']; St = d['stencil_roofline']; cb = d['cpu_baseline']; c2 = d['cpu_baseline_2n']; cpool = d['cpu_baseline_pool']
rows3 = list(csv.DictReader(open(P + 'r02_config5_rocprofv3_kernel_stats.csv')))
b3 = json.load(open(P + 'r02_config5_bench3d.json'))
std_note = ''
if os.path.exists(P + 'r02_config5_standard_cycle_4src.json'):
    b3s = json.load(open(P + 'r02_config5_standard_cycle_4src.json'))
    std_note = ': ' + ', '.join('%g Hz %.1f s, %d iterations' % (x['freq'], x['seconds_device_resident'], max(x['iterations'])) for x in b3s['solve'])
nB = d['config']['sources_per_step']; N = d['config']['grid'][0] * d['config']['grid'][1]
alg_norm_only = N * (32.0 * nB + 144.0)
text = f'''# profiles/ -- round 2 (MI355X, 1 GPU)

Collected by `tools/run_profiles.sh` on the GPU box (one `gpurun` call) and summarised by `tools/update_profiles.py`; round-1 files (`r01_*`)
are kept for the before / after comparison and described at the end.

## Default path: `python bench.py` on the 1024 x 1024 Eurus job (work item = create + assemble + factor one frequency + solve 256 sources)

| file | what |
|---|---|
| `r02_bench_n1.json` | `python bench.py` (2 timed items, {d['config']['freqs_hz_this_run'][0]:g} and {d['config']['freqs_hz_this_run'][1]:g} Hz): **{d['value']:.0f} wavefields/s**, {d['ms_per_step']:.1f} ms per item with the per-launch HIP events on ({d['unprofiled']['value']:.0f} with them off), device {d['config']['device_ms_per_step']['solve_call']:.1f} ms in the solve call of which {d['config']['device_ms_per_step']['of_which_factorisation']:.1f} ms factorisation (round 1: 4245 wavefields/s, 60.3 ms, 28.7 ms); `parity_vs_lu_max_rel` = {d['parity_vs_lu_max_rel']:.2e} (8 sources at 6 Hz against the SuperLU wavefields of the CPU leg); `roofline`: all `k_zgemm2` launches of the timed items, {R['achieved']:.1f} TFLOP/s = **{100 * R['frac']:.0f} %** of the 78.6 TFLOP/s nominal fp64 peak ({R['launches_timed']} launches, avg {R['avg_launch_us']:.0f} us; the launches of >= 1 GFLOP: {R['launches_of_at_least_1_GFLOP']['achieved']:.1f} TFLOP/s); `stencil_roofline` / `roofline_northstar`: the node-major residual launches, {St['achieved']:.0f} GB/s = **{100 * St['frac']:.0f} %** of 8 TB/s on N(32B + 144), plus the rhs-major apply microbenchmark ({', '.join('%.0f' % (100 * m['frac_of_peak']) for m in St['apply_microbench'])} % at B = 1 / 8 / 32 / 64); CPU legs on the same host (256 logical CPUs): 1 core, M1-only LU {cb['value']:.2f} wavefields/s (assemble {cb['assemble_s']:.1f} s, factor {cb['factor_s']:.1f} s, {cb['per_rhs_s']:.3f} s per source); the faithful 2N x 2N system at 512^2: {c2['value']:.2f} wavefields/s (factor {c2['factor_s']:.1f} s, {c2['per_rhs_s']:.3f} s per source); 16 processes, one per frequency: {cpool['value']:.1f} wavefields/s |
| `r02_bench_n1_rocprofv3_kernel_stats.csv` | `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-cpu` (summary only) |
| `r02_bench_n1_under_rocprofv3.json` | the bench line of that profiled run ({d2['value']:.0f} wavefields/s) |
| `r02_pmc_traffic_zgemm.json`, `r02_pmc_traffic_resid_nm.json`, `r02_pmc_traffic_stencil_apply.json` | `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (two separate passes) of `python3 bench.py --steps 1 --warmup 0 --no-cpu --no-plain-pass`, reduced per kernel by `tools/pmc_reduce.py` (FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950): `k_zgemm2` {pz['traffic_bytes_per_launch'] / 1e6:.0f} MB of HBM traffic per launch over {pz['launches_fetch_pass']} launches; `k_resid_nm` {pr['traffic_bytes_per_launch'] / 1e9:.2f} GB per launch against {alg_norm_only / 1e9:.2f} GB algorithmic N(32B + 144) (norm-only launch: x and q' in, nothing out -- the difference is the vertical halo rows of x re-read from HBM); the rhs-major apply of the microbenchmark {ps['traffic_bytes_per_launch'] / 1e9:.2f} GB per launch averaged over B = 1, 8, 32, 64 (algorithmic average 0.95 GB) |
| `r02_direct_per_level_trace.txt` | `HELM_ND_TRACE=1 python tools/bench_direct.py --freqs 5.5`: device milliseconds per tree level of the factorisation and of the forward / backward sweeps (HIP events between levels) |

Agreement check: the profiler's average over all `k_zgemm2<...>` instantiations is {gc} launches, {gn / 1e6:.1f} ms, **{gn / gc / 1e3:.1f} us**
({100 * gn / tot:.0f} % of the GPU time); bench.py's HIP-event average is {R['avg_launch_us']:.1f} us ({d2['roofline']['avg_launch_us']:.1f} us in the run under the profiler).

Kernel time of the profiled run (1 warm-up + 2 timed + 2 un-profiled items + the parity item), top rows of the stats file:

| kernel | calls | avg us | % of GPU time |
|---|---|---|---|
{table_of(rows, 16)}

(`k_zgemm2<TM, IDX, RN, KS, UNR, OCC>`: tile height, operand addressing 0 dense / 1 row table / 2 forward gather, columns per thread, K slab,
k-loop unroll, waves per SIMD asked for; `k_gj_panel` = pivot block inverse + row / column panels of the blocked Gauss-Jordan;
`k_nd_build_front` = stencil entries + both children's Schur complements gathered into a front in one pass; `k_resid_nm` =
node-major true residual; `k_stencil_t` = the rhs-major apply of the microbenchmark.)

### What the fp64 units give (`r02_fp64_rate_probe.txt`, `tools/fp64_rate.hip`)

A register-only loop with the tile kernel's exact instruction mix (4 x 4 complex block: 64 `v_fma_f64` on 8 operand pairs and 32
accumulator pairs, no LDS, no memory) runs at **45-49 TFLOP/s** at 1-3 waves per SIMD; a chain `acc = fma(a, acc, b)` with two
loop-invariant operands reaches 67; `v_mfma_f64_16x16x4` 46-50; and issuing MFMAs next to the vector FMAs of the same wave does
not add up (45-54 in total for 0-4 MFMAs per 64 FMAs): the two pipes share the fp64 throughput.  The practical ceiling of a complex
GEMM on this part is therefore ~48 TFLOP/s (61 % of the nominal 78.6); `k_zgemm2` reaches 43-45.5 on its large launches
(`r02_zgemm_lab_variants.txt`: v0 = round-1 kernel, v1 = shipped, v2-v5 = K slab 16 / unroll 2 / 4-wave budget, dropped), i.e. ~93 % of
what the units deliver for this mix.  `r02_zgemm_tiles_underfilled.txt`: every tile shape on the small-batch launches of the upper tree
levels (basis of the under-filled-launch rule in `gemm()`).  A 3M complex product (three real products per complex one) was written and
measured too: 3 % slower and two more frequencies pushed into a second pass (DESIGN.md section 8).

## Config 5: 3-D 27-point, 256 x 256 x 128, c = 2000 m/s, h = 10 m, 4 frequencies x 16 sources (`tools/bench3d.py --freqs 2 3 4 5 --nsrc 16`)

| file | what |
|---|---|
| `r02_config5_bench3d.json` / `.txt` | batched apply {', '.join('%.0f' % a['GBps_algorithmic'] for a in b3['apply'])} GB/s at B = 1 / 4 / 8 / 16 against N(32B + 432) ({', '.join('%.0f' % (a['GBps_algorithmic'] / 80) for a in b3['apply'])} % of 8 TB/s); the whole job through host buffers: {' + '.join('%.1f' % s['seconds'] for s in b3['solve'])} s = **{sum(s['seconds'] for s in b3['solve']):.1f} s** for the 64 wavefields at 2 / 3 / 4 / 5 Hz to rtol 1e-8 ({', '.join('%d' % (sum(s['iterations']) / len(s['iterations'])) for s in b3['solve'])} BiCGSTAB iterations per source on average); with right-hand sides and wavefields resident in HBM {' + '.join('%.1f' % s['seconds_device_resident'] for s in b3['solve'])} s = **{sum(s['seconds_device_resident'] for s in b3['solve']):.1f} s**, of which {sum(s['seconds_device_resident'] - s['seconds_device_resident_reusing_setup'] for s in b3['solve']):.1f} s are the preconditioner set-ups (plane inverses of the coarsest level) |
| `r02_config5_rocprofv3_kernel_stats.csv`, `r02_config5_under_rocprofv3.json` | `rocprofv3 --kernel-trace --stats -- python3 tools/bench3d.py --freqs 5 --nsrc 16` |
| `r02_config5_standard_cycle_*` | the same job with the cycle this round started with (standard coarsening, weak layer, shift 6-8; collected before the layer-preserving hierarchy existed): 134 s, 500-1500 iterations per source; `r02_config5_standard_cycle_4src.json`: that cycle again on the final code (`--standard-cycle`, 4 sources at 2 and 5 Hz){std_note} |

Top rows of the 3-D stats file (5 Hz, 16 sources):

| kernel | calls | avg us | % of GPU time |
|---|---|---|---|
{table_of(rows3, 10)}

(`k_stencil3<false, EPI>`: 4 = residual, 5 = l1-Jacobi sweep of the multigrid levels; `k_zgemm2<64, 1, 4, 8>` + `k_gj_panel` + `k_bt_schur_t`: the plane inverses of the set-up; `k_bt_apply` / `k_bt_rhs` / `k_bt_reduce`: the block-tridiagonal solve of the coarsest level, 1 / 6 = the outer BiCGSTAB applies with fused dot
products.)

## Round 1 files

`r01_bench_n1*`, `r01_pmc_traffic_direct.json`: the default path as of round 1 (4245 wavefields/s, `k_zgemm` 23.2 TFLOP/s over 1150 launches of
55 us, 575 launches per work item against 169 now).  `r01_krylov_*`: `python bench.py --method mg --batch 64`, the multigrid-preconditioned
BiCGSTAB that remains the solver for 3-D and the fallback (18.2 wavefields/s then, 19.5-20.9 now at batch 64 / 256; outer-iteration stencil
launches 4250 GB/s = 53 % of 8 TB/s, PMC traffic 3.21 GB per launch against 3.19 GB algorithmic).
'''
open(P + 'README.md', 'w').write(text)
print('profiles/README.md regenerated')
