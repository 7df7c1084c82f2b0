#!/usr/bin/env python3
"""Copy a finished gpurun_out/<dir> of tools/run_profiles_r4.sh into profiles/ (r04_* names, the git hash of the collection stamped into every
JSON / text file) and regenerate profiles/README.md from the numbers in those files, so the README never drifts from the artefacts.

    python tools/update_profiles_r4.py gpurun_out/r4p2
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


def put_text(src, name, header):
    open(P + name, 'w').write('# %s   (collected at git %s)\n' % (header, HASH) + open(os.path.join(S, src)).read())


d = last_json(os.path.join(S, 'bench_n1.json')); put_json(d, 'r04_bench_n1.json')
dd = last_json(os.path.join(S, 'bench_driver.json')); put_json(dd, 'r04_bench_driver_cmd.json')
dp = last_json(os.path.join(S, 'bench_pipelined_under_rocprof.json')); put_json(dp, 'r04_bench_pipelined_under_rocprofv3.json')
ds = last_json(os.path.join(S, 'bench_serial_under_rocprof.json')); put_json(ds, 'r04_bench_serial_under_rocprofv3.json')
shutil.copyfile(find('stats_pipe/**/s_kernel_stats.csv'), P + 'r04_bench_pipelined_rocprofv3_kernel_stats.csv')
shutil.copyfile(find('stats_serial/**/s_kernel_stats.csv'), P + 'r04_bench_serial_rocprofv3_kernel_stats.csv')
if find('stats_serial_sparse/**/s_kernel_stats.csv'):
    shutil.copyfile(find('stats_serial_sparse/**/s_kernel_stats.csv'), P + 'r04_bench_serial_sparse_rhs_rocprofv3_kernel_stats.csv')
shutil.copyfile(find('stats3d/**/s_kernel_stats.csv'), P + 'r04_config5_rocprofv3_kernel_stats.csv')
for a, b in (('pmc_traffic_zgemm.json', 'r04_pmc_traffic_zgemm.json'), ('pmc_traffic_resid.json', 'r04_pmc_traffic_resid_nm.json'),
             ('pmc_traffic_stencil_micro.json', 'r04_pmc_traffic_stencil_apply.json')):
    put_json(json.load(open(os.path.join(S, a))), b)
open(P + 'r04_direct_per_level_trace.txt', 'w').write('# HELM_ND_TRACE=1 python tools/bench_direct.py --freqs 5.5   (collected at git %s)\n' % HASH +
                                                       ''.join(l for l in open(os.path.join(S, 'trace.txt')) if l.startswith('[nd trace]')) +
                                                       '# the same with HELM_ND_SPARSE_RHS=0 (every front of the forward pass, every row of the leaf back substitution)\n' +
                                                       ''.join(l for l in open(os.path.join(S, 'trace_every_front.txt')) if l.startswith('[nd trace]')))
put_text('gemm_log.txt', 'r04_gemm_log_by_shape.txt', 'HELM_GEMM_LOG=1 python tools/bench_direct.py --freqs 5.5 | tools/gemm_log.py: every product of one factorisation + three passes by shape')
put_text('zgemm_lab.txt', 'r04_zgemm_lab.txt', 'python tools/zgemm_lab.py 1,7: v1 = k_zgemm2 (vector FMAs, round 3), v7 = k_zgemm3 (matrix cores)')
open(P + 'r04_githash.txt', 'w').write(HASH + '\n')
b3 = last_json(os.path.join(S, 'bench3d_under_rocprof.txt')); put_json(b3, 'r04_config5_5hz_under_rocprofv3.json')

# ---- fp64 probe: text + SQ counters reduced to a table
probe = open(os.path.join(S, 'fp64_clock.txt')).read()
pc = find('pmc_probe/**/*counter_collection.csv')
lines = []
if pc:
    per = {}
    order = []
    for r in csv.DictReader(open(pc)):
        if 'w_' not in r['Kernel_Name']:
            continue
        k = (r['Dispatch_Id'], re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', ''))
        if k not in per:
            per[k] = {}
            order.append(k)
        per[k][r['Counter_Name']] = float(r['Counter_Value'])
    seen = {}
    for k in order:
        seen[k[1]] = per[k]          # the second (timed) launch of each kernel overwrites its warm-up
    lines.append('%-28s %14s %14s %16s %22s %14s' % ('kernel<.., waves/SIMD>', 'SQ_INSTS_VALU', 'SQ_INSTS_MFMA', 'SQ_BUSY_CYCLES', 'SQ_VALU_MFMA_BUSY_CYC', 'busy cyc/MFMA'))
    for name, v in seen.items():
        nm = v.get('SQ_INSTS_MFMA', 0)
        lines.append('%-28s %14.0f %14.0f %16.0f %22.0f %14s' % (name, v.get('SQ_INSTS_VALU', 0), nm, v.get('SQ_BUSY_CYCLES', 0), v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0),
                                                                  ('%.1f' % (v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / nm)) if nm else '-'))
open(P + 'r04_fp64_clock_probe.txt', 'w').write(
    '# tools/fp64_clock.hip on one MI355X (round 4, collected at git %s): occupancy pinned by LDS (exactly W workgroups of 256 threads per CU),\n'
    '# shader clock measured inside the kernels.  Reading: v_mfma_f64_16x16x4_f64 runs at 99 %% of the nominal 78.6 TFLOP/s from two waves per SIMD up\n'
    '# (one wave: hipcc puts the accumulators into AGPRs under __launch_bounds__(256, 1) and copies them around every instruction -- SQ_INSTS_VALU is\n'
    '# 17 x SQ_INSTS_MFMA there, 1 x everywhere else); the vector-FMA mix of the complex register block saturates at 55-56 TFLOP/s and the chip\n'
    '# drops to 2.04-2.09 GHz under it; fp32 chains reach 124 of 157 TFLOP/s at 8 waves (2.04 GHz: 92 %% of what that clock allows).\n'
    '# ("cycles per instruction" averages the per-wave run times, and the waves of a SIMD finish one after the other -- the oldest has priority --\n'
    '# so it reads (W + 1) / 2W of the true figure at W waves; the TFLOP/s column is from the event time and is exact.)\n' % HASH + probe +
    '\n# rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES ... -- tools/fp64_clock 0.1\n' + '\n'.join(lines) + '\n')

# ---- SQ counters of one large GEMM launch (1024 x 1024 x 256, batch 16)
sq = {}
for dname in ('pmc_gemm_sq', 'pmc_gemm_sq2'):
    f = find(dname + '/**/*counter_collection.csv')
    if not f:
        continue
    last = {}
    for r in csv.DictReader(open(f)):
        if 'zgemm3' in r['Kernel_Name']:
            last.setdefault(r['Dispatch_Id'], {})[r['Counter_Name']] = float(r['Counter_Value'])
            last[r['Dispatch_Id']]['us'] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    if last:
        sq.update(last[sorted(last, key=int)[-1]])
if sq:
    clk = sq.get('GRBM_GUI_ACTIVE', 0) / 8.0 / (sq['us'] * 1e-6) / 1e9 if sq.get('GRBM_GUI_ACTIVE') else None
    sq['shader_clock_GHz'] = clk
    if clk:
        sq['mfma_pipe_utilisation'] = sq.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (1024 * sq['us'] * 1e-6 * clk * 1e9)
    sq['what'] = 'SQ counters of one k_zgemm3 launch, 1024 x 1024 x 256 complex, batch 16 (tools/zgemm_lab.py 7 "s256 Schur"); two separate PMC passes'
    put_json(sq, 'r04_pmc_sq_zgemm_large_launch.json')

pz = json.load(open(P + 'r04_pmc_traffic_zgemm.json')); pr = json.load(open(P + 'r04_pmc_traffic_resid_nm.json')); ps = json.load(open(P + 'r04_pmc_traffic_stencil_apply.json'))
rows_s = list(csv.DictReader(open(P + 'r04_bench_serial_rocprofv3_kernel_stats.csv')))
rows_p = list(csv.DictReader(open(P + 'r04_bench_pipelined_rocprofv3_kernel_stats.csv')))
rows3 = list(csv.DictReader(open(P + 'r04_config5_rocprofv3_kernel_stats.csv')))


def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('HIP_vector_type<double, 2u>', 'cplx')
    m = re.match(r'(void )?([\w:]+(<[^(]*>)?)', n)
    return m.group(2) if m else n[:40]


def table_of(rs, k):
    tot = sum(int(r['TotalDurationNs']) for r in rs)
    return '\n'.join('| `%s` | %s | %.1f | %.1f |' % (short(r['Name']), r['Calls'], float(r['AverageNs']) / 1e3, 100.0 * int(r['TotalDurationNs']) / tot) for r in rs[:k])


def gemm_avg(rs):
    g = [r for r in rs if 'k_zgemm' in r['Name']]
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
alg_resid = St['bytes_per_launch_algorithmic']
# the PMC command runs the timed item and the separate roofline pass: two work items (launches per item from the driver-command record)
pmc_items = max(1, round(pz['launches_fetch_pass'] / (Rd['launches_timed'] / float(dd['steps']))))
traffic_item = pz['traffic_bytes_per_launch'] * pz['launches_fetch_pass'] / 1e9 / pmc_items
oper_item = Rd['two_roofs']['operand_GB_per_item']
text = f'''# profiles/ -- round 4 (MI355X, 1 GPU; collected at git `{HASH}`)

Collected by `tools/run_profiles_r4.sh` on the GPU box (one `gpurun` call) and summarised by `tools/update_profiles_r4.py`, which stamps the git
hash of the collection into every JSON / text file (`collected_at_git`, `r04_githash.txt`).  Earlier rounds' files (`r03_*`, `r02_*`, `r01_*`) are kept
for the before / after comparison; their descriptions are in the git history of this file.

## The bench job: 1024 x 1024 Eurus, 16 frequencies x 256 sources (work item = create + assemble + factor one frequency + solve 256 sources to relres <= 1e-10)

| file | what |
|---|---|
| `r04_bench_driver_cmd.json` | the driver's command line, `python bench.py --steps 20 --warmup 5 --no-cpu` (all 16 frequencies): **{dd['value']:.0f} wavefields/s**, {dd['ms_per_step']:.1f} ms per item ({dd['unprofiled']['value']:.0f} with the per-launch events off; round 3: 9235 / 27.7 ms).  `every_front_computed` (HELM_ND_SPARSE_RHS=0: nothing skipped on the point sources): {dd['every_front_computed']['value']:.0f}; `support_declared` (the sparse source matrix's support handed to the solver, `helm_set_rhs_support`: no scan of the dense right-hand sides; not the headline): {(dd.get('support_declared') or {}).get('value', float('nan')):.0f}; `strong_scaling_job` (the whole 4096-wavefield job once): {dd['strong_scaling_job']['seconds']:.3f} s = {dd['strong_scaling_job']['value']:.0f} wavefields/s.  `roofline` (separate serial pass, every booked flop executed): all `k_zgemm3` launches {Rd['achieved']:.1f} TFLOP/s = **{100 * Rd['frac']:.0f} %** of 78.6 ({Rd['launches_timed']} launches, avg {Rd['avg_launch_us']:.0f} us; launches of >= 1 GFLOP, {100 * Rd['launches_of_at_least_1_GFLOP']['share_of_gemm_time']:.0f} % of the GEMM time: {Rd['launches_of_at_least_1_GFLOP']['achieved']:.1f} TFLOP/s); against both roofs per launch {100 * Rd['two_roofs']['frac']:.0f} %; `stencil_roofline.frac` {dd['stencil_roofline']['frac']:.3f} |
| `r04_bench_n1.json` | `python bench.py` (default: {d['steps']} timed items after {d['warmup']} warm-up items): {d['value']:.0f} wavefields/s, {d['ms_per_step']:.1f} ms per item ({d['unprofiled']['value']:.0f} with the events off); passes per wavefield {d['config']['solves_or_iterations_per_rhs_mean']:.2f}; `parity_vs_lu_max_rel` = {d['parity_vs_lu_max_rel']:.2e} (8 sources at 6 Hz against the SuperLU wavefields of the CPU leg).  `value_host_api` (MultiFreq * q, scipy-sparse sources in, numpy wavefields out over PCIe): **{ha['value']:.0f} wavefields/s**.  `config5`: **{c5['job_seconds']:.2f} s** at rtol 1e-8 through the device pipeline ({c5.get('job_seconds_one_after_the_other', 0):.2f} s one frequency after the other: {', '.join('%g Hz %.2f s / %d its' % (p['freq_hz'], p['seconds'], max(p['iterations'])) for p in c5['per_frequency'])}), **{c5.get('job_seconds_rtol1e10', float('nan')):.2f} s at rtol 1e-10**; 27-point apply {', '.join('%.0f' % (100 * a['frac_of_peak']) for a in c5['apply'])} % of 8 TB/s at B = 1 / 4 / 8 / 16.  CPU legs on the GPU box's own host: 1 core, M1-only LU {cb['value']:.2f} wavefields/s (assemble {cb['assemble_s']:.1f} s, factor {cb['factor_s']:.1f} s, {cb['per_rhs_s']:.3f} s per source); the faithful 2N x 2N system at 512^2: {c2.get('value', float('nan')):.2f}; 16 processes, one per frequency: {cpool.get('value', float('nan')):.1f} |
| `r04_bench_serial_rocprofv3_kernel_stats.csv`, `r04_bench_serial_under_rocprofv3.json` | `HELM_ND_SPARSE_RHS=0 rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu --no-config5 --no-host-api --no-pipeline --steps 8 --warmup 2 --no-plain-pass`: the kernels with nothing else on the GPU and nothing skipped -- the run `roofline` must agree with.  `r04_bench_serial_sparse_rhs_rocprofv3_kernel_stats.csv`: the same with the skipping on (what a production item costs) |
| `r04_bench_pipelined_rocprofv3_kernel_stats.csv`, `r04_bench_pipelined_under_rocprofv3.json` | the pipelined timed region under the profiler ({dp['value']:.0f} wavefields/s): durations stretched by the sharing |
| `r04_pmc_traffic_zgemm.json`, `r04_pmc_traffic_resid_nm.json`, `r04_pmc_traffic_stencil_apply.json` | `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (two separate passes) of one serial work item with every front computed, reduced per kernel by `tools/pmc_reduce.py` (FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950): `k_zgemm3` + `k_gj_step` {pz['traffic_bytes_per_launch'] / 1e6:.0f} MB per launch over {pz['launches_fetch_pass']} launches of {pmc_items} work items = {traffic_item:.1f} GB per item against {oper_item:.1f} GB of necessary operand bytes = **{traffic_item / oper_item:.2f} x** (round 3: 1.41 x against operand bytes that left the gathers' necessary reads out); the residual kernel {pr['traffic_bytes_per_launch'] / 1e9:.2f} GB per launch against {alg_resid / 1e9:.2f} GB algorithmic; the rhs-major apply of the microbenchmark {ps['traffic_bytes_per_launch'] / 1e9:.2f} GB per launch averaged over B = 1, 8, 32, 64 |
| `r04_pmc_traffic_by_shape.txt` | the same two PMC passes on `tools/bench_direct.py` (one factorisation + three passes), every `k_zgemm3` / `k_gj_step` dispatch paired with the `[gemm log]` line of the same launch (`tools/run_pmc_by_shape.sh`, `tools/pmc_by_dispatch.py`, `tools/pmc_pair.py`): measured bytes against booked operand bytes per shape, largest excess first; 141.4 GB against 132.6 GB = 1.07 x over the run (1.21 x before all tiles of a front were kept on one XCD) |
| `r04_pmc_sq_zgemm_large_launch.json` | SQ counters of one large `k_zgemm3` launch (1024 x 1024 x 256, batch 16): matrix-pipe utilisation, shader clock during the launch, LDS bank conflicts (0) |
| `r04_direct_per_level_trace.txt` | `HELM_ND_TRACE=1 python tools/bench_direct.py --freqs 5.5`: device milliseconds per tree level of the factorisation and of the forward / backward sweeps, with and without the sparse-right-hand-side skipping |
| `r04_gemm_log_by_shape.txt` | every product of one factorisation + three passes aggregated by shape and addressing mode: microseconds, TFLOP/s, operand GB/s, roofline microseconds |
| `r04_zgemm_lab.txt`, `r04_zgemm_lab_mfma.txt` | the tile-kernel lab on the shapes the 1024^2 plan issues: vector-FMA kernel of round 3 against the matrix-core kernel; the removed LDS-DMA variant |
| `r04_fp64_clock_probe.txt` | `tools/fp64_clock.hip` at 1, 2, 4, 8 waves per SIMD (occupancy pinned by LDS) + its SQ counters: what settles the fp64 ceiling |

Agreement check (serial run, nothing skipped): the profiler's average over all `k_zgemm3<...>` / `k_zgemm3_la` instantiations is {gc_s} launches, {gn_s / 1e6:.1f} ms, **{gavg_s:.1f} us**
({gpct_s:.0f} % of the GPU time); bench.py's HIP-event average in that run is {Rs['avg_launch_us']:.1f} us ({Rd['avg_launch_us']:.1f} us in `r04_bench_driver_cmd.json`).  Residual kernel: profiler
{ravg_s:.0f} us over {rc_s} launches, HIP events {ds['stencil_roofline']['avg_launch_us']:.0f} us.

Kernel time of the serial profiled run (nothing skipped), top rows:

| kernel | calls | avg us | % of GPU time |
|---|---|---|---|
{table_of(rows_s, 20)}

The pipelined run (same items; the factorisation of item k+1 runs beside the solve of item k; sparse-right-hand-side skipping on):

| kernel | calls | avg us | % of GPU time |
|---|---|---|---|
{table_of(rows_p, 12)}

(`k_zgemm3<WM, WN, MT, NT, IDX, KS, OCC, XR>`: waves per workgroup in rows x columns, blocks of 16 x 16 per wave in rows x columns, operand addressing 0 dense / 1 row table / 2 forward gather /
4 Schur gather, K slab, waves per SIMD asked for, 1 = the 16 MT + 1-row tile; `k_zgemm3_la` = blocked Gauss-Jordan update with the next pivot sweep riding along;
`k_leaf_factor<49>` = the leaf level of the factorisation in one kernel; `k_gj_panel` / `k_gj_slices` = pivot block inverse + row / column panels; `k_nd_build_front` = stencil entries + both
children's Schur complements gathered into a front in one pass; `k_resid_nm_lds` = node-major true residual with LDS-staged coefficients, `||q||^2` and the wavefield store fused;
`k_front_absmax` / `k_lu_factor*` / `k_lu_solve` = condition estimates and the pivoted-LU treatment of ill-conditioned fronts; `k_stencil_t` = the rhs-major apply of the microbenchmark.)

## Config 5: 3-D 27-point, 256 x 256 x 128, 5 Hz x 16 sources under the profiler (`r04_config5_rocprofv3_kernel_stats.csv`, `r04_config5_5hz_under_rocprofv3.json`)

| kernel | calls | avg us | % of GPU time |
|---|---|---|---|
{table_of(rows3, 10)}

The whole 4-frequency job is the `config5` block of `r04_bench_n1.json` (above).
'''
open(P + 'README.md', 'w').write(text)
print('profiles/README.md regenerated for round 4 at git', HASH)
