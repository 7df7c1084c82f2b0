#!/usr/bin/env python3
"""Copy a finished gpurun_out/<dir> of the default-path profile runs into profiles/ and regenerate the "Default path"
section of profiles/README.md from the numbers in those files (so the README never drifts from the artifacts).

    python tools/update_profiles.py gpurun_out/r01e
"""
import csv, json, os, re, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
P = os.path.join(ROOT, 'profiles')
shutil.copyfile(os.path.join(src, 'bench_n1.json'), os.path.join(P, 'r01_bench_n1.json'))
shutil.copyfile(os.path.join(src, 'stats', 's_kernel_stats.csv'), os.path.join(P, 'r01_bench_n1_rocprofv3_kernel_stats.csv'))
shutil.copyfile(os.path.join(src, 'bench_under_rocprof.json'), os.path.join(P, 'r01_bench_n1_under_rocprofv3.json'))
shutil.copyfile(os.path.join(src, 'pmc_traffic_direct.json'), os.path.join(P, 'r01_pmc_traffic_direct.json'))
d = json.load(open(os.path.join(P, 'r01_bench_n1.json')))
d2 = json.load(open(os.path.join(P, 'r01_bench_n1_under_rocprofv3.json')))
pm = json.load(open(os.path.join(P, 'r01_pmc_traffic_direct.json')))
rows = list(csv.DictReader(open(os.path.join(P, 'r01_bench_n1_rocprofv3_kernel_stats.csv'))))
g = [r for r in rows if 'k_zgemm' in r['Name']]
calls = sum(int(r['Calls']) for r in g); ns = sum(int(r['TotalDurationNs']) for r in g)
tot = sum(int(r['TotalDurationNs']) for r in rows)


def short(name):
    m = re.search(r'(k_\w+(<[^(]*?>)?|__amd\w+)', name)
    s = m.group(1) if m else name[:40]
    return s.replace('HIP_vector_type<double, 2u>', 'cplx')


table = '\n'.join('| `%s` | %s | %.1f | %s |' % (short(r['Name']), r['Calls'], float(r['AverageNs']) / 1e3, r['Percentage']) for r in rows[:14])
R = d['roofline']
text = '''## Default path

| file | what |
|---|---|
| `r01_bench_n1.json` | `python bench.py`: **%.0f wavefields/s**, %.1f ms per 256-source work item (device: %.1f ms in the solve call, of which %.1f ms factorisation); `roofline` = all `k_zgemm` launches of the timed items, HIP events on the solver stream: %.1f TFLOP/s = **%.0f %% of the 78.6 TFLOP/s fp64 peak** (%d launches, avg %.1f us); `stencil_roofline` = the residual launches of the stencil kernel (%.0f GB/s) + the SURVEY 8(d) apply microbenchmark; CPU baselines on the same box: 1 core %.2f wavefields/s, 16 processes (one per frequency, the reference's pool mode) %.1f wavefields/s |
| `r01_bench_n1_rocprofv3_kernel_stats.csv` | `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-cpu` (summary only) |
| `r01_bench_n1_under_rocprofv3.json` | the bench line printed by that profiled run (%.0f wavefields/s) |
| `r01_pmc_traffic_direct.json` | `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (two separate passes) of `python3 bench.py --steps 1 --warmup 0 --no-cpu`, reduced over all `k_zgemm` dispatches by `tools/pmc_reduce.py` (FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950): %.0f MB of HBM traffic per launch |

Agreement check: the profiler's average over the `k_zgemm<TM, IDX>` instantiations is %d launches, %.1f ms, **%.1f us**
(%.0f %% of the GPU time); bench.py's HIP-event average over its %d timed launches is %.1f us (%.1f us in the run under the profiler).

Arithmetic intensity of the average GEMM launch: %.2f GFLOP algorithmic against %.0f MB moved = %.1f flop/B, just below the
fp64 ridge of the part (78.6 TFLOP/s / 8 TB/s = 9.8 flop/B): the big launches (leaf level: 16 384 fronts x 256 right-hand
sides) stream their operands once and are co-limited by HBM, the several hundred small launches of the upper tree levels are
latency-bound.  The tile kernel itself sustains 45-47 TFLOP/s on the flops it executes (leaf-level launches).

Kernel time of the profiled run (3 work items), top rows of the stats file:

| kernel | calls | avg us | %% of GPU time |
|---|---|---|---|
%s

(`k_zgemm<TM, IDX>`: TM = tile height 64/32/16 chosen per shape, IDX = operand rows addressed through the row table;
`k_stencil_t<.., 4, ..>` = true residual q' - A x; `k_gj_inverse` = pivoted Gauss-Jordan base blocks of the front inversions.)

''' % (d['value'], d['ms_per_step'], d['config']['device_ms_per_step']['solve_call'], d['config']['device_ms_per_step']['of_which_factorisation'],
       R['achieved'], 100 * R['frac'], R['launches_timed'], R['avg_launch_us'], d['stencil_roofline']['achieved'],
       d['cpu_baseline']['value'], d['cpu_baseline_pool']['value'] if isinstance(d.get('cpu_baseline_pool'), dict) else float('nan'),
       d2['value'], pm['traffic_bytes_per_launch'] / 1e6, calls, ns / 1e6, ns / calls / 1e3, 100.0 * ns / tot,
       R['launches_timed'], R['avg_launch_us'], d2['roofline']['avg_launch_us'],
       R['flops_per_launch_algorithmic'] / 1e9, pm['traffic_bytes_per_launch'] / 1e6, R['flops_per_launch_algorithmic'] / pm['traffic_bytes_per_launch'], table)
readme = open(os.path.join(P, 'README.md')).read()
a = readme.index('## Default path')
b = readme.index('## Krylov path')
open(os.path.join(P, 'README.md'), 'w').write(readme[:a] + text + readme[b:])
print('profiles/ updated: %.0f wavefields/s, roofline %.3f' % (d['value'], R['frac']))
