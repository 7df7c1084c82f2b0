"""bench.py --gpus N: rank spawning (CPU part: environment, relay of rank 0's line, failure propagation) and, on the GPU
box, a 2-rank run on one GPU over gloo."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

FAKE_RANK = r'''
import json, os, sys
rank = int(os.environ['RANK'])
if rank == 0:
    print(json.dumps({k: os.environ.get(k) for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT', 'HSA_ENABLE_IPC_MODE_LEGACY')} | {'argv': sys.argv[1:]}))
sys.exit(int(os.environ.get('FAIL_RANK', '-1')) == rank)
'''


def test_spawn_ranks_sets_rendezvous_env_and_propagates_failures(tmp_path, monkeypatch, capsys):
    import bench
    script = tmp_path / 'fake_bench.py'
    script.write_text(FAKE_RANK)
    monkeypatch.setattr(bench, '__file__', str(script))
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    assert bench.spawn_ranks(3, ['--gpus', '3', '--steps', '2']) == 0
    line = json.loads(capsys.readouterr().out.strip())
    assert line['WORLD_SIZE'] == '3' and line['RANK'] == '0' and line['LOCAL_RANK'] == '0'
    assert line['MASTER_ADDR'] == '127.0.0.1' and int(line['MASTER_PORT']) > 0 and line['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'
    assert line['argv'] == ['--gpus', '3', '--steps', '2']
    monkeypatch.setenv('FAIL_RANK', '2')
    assert bench.spawn_ranks(3, ['--gpus', '3']) == 1


@pytest.mark.gpu
def test_bench_gpus_2_runs_two_ranks(helm_lib):
    """`python bench.py --gpus 2` with no launcher starts two rank processes (here both on the one GPU of the box, rendezvous over
    gloo) and prints ONE line with n_gpus == 2 and the work of both ranks."""
    env = dict(os.environ, HELM_BENCH_BACKEND='gloo')
    env.pop('WORLD_SIZE', None); env.pop('RANK', None); env.pop('LOCAL_RANK', None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--grid', '128', '--steps', '2', '--no-cpu'],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    lines = [l for l in out.stdout.decode().splitlines() if l.startswith('{')]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec['n_gpus'] == 2 and rec['steps'] == 2 and rec['scaling'] == 'weak'
    assert rec['value'] > 0 and abs(rec['value'] - 2 * 2 * 256 / (rec['ms_per_step'] * 2e-3)) < 1e-6 * rec['value']
    # what the next multi-GPU run of the driver can be checked against: the driver keeps the first 24 SCALAR keys of `config` (and 120 characters of a
    # string) and nothing nested -- the multi-GPU diagnostics must be among them
    cfg = rec['config']
    assert len(cfg) <= 24 and len(cfg['workload']) <= 120
    assert all(isinstance(v, (str, int, float, type(None))) for v in cfg.values()), [k for k, v in cfg.items() if isinstance(v, (list, dict))]
    more = rec['detail']['flat']
    assert cfg['rccl_ranks_seen'] == 2 and more['collective_backend'] == 'gloo'
    assert more['ms_per_step_rank0'] > 0 and more['ms_per_step_rank1'] > 0
    assert abs(cfg['ms_per_step_slowest_rank'] - max(more['ms_per_step_rank0'], more['ms_per_step_rank1'])) < 1e-4
    assert cfg['ms_per_step_slowest_rank'] <= rec['ms_per_step'] * 1.05          # the headline takes the slowest rank (+ the closing barrier)
    # config 4 ran sharded over the two ranks: both legs timed, the gradient crossed a real two-rank all-reduce
    assert cfg['c4_dpred_s'] > 0 and cfg['c4_jtvec_s'] > 0 and cfg['gradient_allreduce_ms'] > 0 and more['c4_allreduce_ms_1024'] > 0
    assert cfg['strong_job_wfs'] > 0
    # the timed region created nothing: pools, events, pinned records and every kernel's dispatch record exist before it starts
    for k in ('timed_max_item_gap_ms', 'timed_p50_item_ms', 'timed_dev_allocs', 'timed_dev_alloc_ms', 'timed_cpu_throttled_ms', 'cpu_quota_cores', 'timed_first_launches'):
        assert k in cfg
    assert rec['config4']['gradient_repeatable_rel'] <= 1e-12
