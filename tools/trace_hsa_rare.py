#!/usr/bin/env python3
"""HSA / HIP calls of a rocprofv3 api trace that are NOT the per-launch ones (memory lock / pool allocate / queue / signal create ...), with time stamps relative to
the first kernel -- what the runtime did around a stall.   tools/trace_hsa_rare.py <dir> [t0 ms] [t1 ms]"""
import csv, sys, glob, os, collections
d = sys.argv[1]; w0 = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0; w1 = float(sys.argv[3]) if len(sys.argv) > 3 else 1e12
def load(pat):
    f = glob.glob(os.path.join(d, '**', pat), recursive=True)
    return list(csv.DictReader(open(f[0]))) if f else []
K = load('*kernel_trace.csv'); t00 = min(int(r['Start_Timestamp']) for r in K)
common = ('hsa_signal_', 'hsa_queue_load', 'hsa_queue_add', 'hsa_queue_store', 'hsa_queue_cas', 'hsa_amd_signal_async', 'hsa_system_get_info', 'hsa_amd_profiling_get', 'hsa_agent_get_info', 'hsa_amd_pointer_info', 'hsa_amd_memory_async_copy', 'hsa_amd_signal_create', 'hsa_amd_agents_allow', 'hsa_executable', 'hsa_code_object', 'hsa_isa', 'hsa_amd_memory_pool_get_info', 'hsa_amd_agent_memory_pool', 'hsa_amd_memory_fill')
cnt = collections.Counter(); rows = []
for r in load('*hsa_api_trace.csv'):
    fn = r['Function']
    if fn.startswith(common): continue
    s = (int(r['Start_Timestamp']) - t00) / 1e6
    if s < w0 or s > w1: continue
    cnt[fn] += 1; rows.append((s, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6, fn, r['Thread_Id']))
print(cnt.most_common(40))
for s, dur, fn, th in sorted(rows)[:400]: print('%10.2f ms  %8.3f ms  %-44s thread %s' % (s, dur, fn, th))
