"""CPU, world_size 2 and 8, gloo: bench.py's main() itself -- launch through torch.distributed.run, per-rank states,
barriers, MAX-reduce of the elapsed time, the chunked validation all-gather, the recomputation of a remote rank's
states and the checksums -- with a CPU stand-in for the evaluator that this directory supplies
(PJ_BENCH_EVALUATOR=stub_evaluator:make, Jacobians from the oracle), so that the first 8-GPU run of the driver is
not also the first execution of that code."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launch(world, validate='512', extra_env=None, timeout=600):
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, PJ_BENCH_EVALUATOR='stub_evaluator:make', MASTER_ADDR='127.0.0.1', PJ_VALIDATE_CHUNK='100',
               PYTHONPATH=os.path.join(ROOT, 'tests') + os.pathsep + os.environ.get('PYTHONPATH', ''))
    env.update(extra_env or {})
    return subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(world),
                           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.join(ROOT, 'bench.py'),
                           '--gpus', str(world), '--steps', '3', '--warmup', '1', '--workload', 'h2', '--states', '700',
                           '--validate-states', validate, '--no-cpu-baseline'], env=env, capture_output=True, text=True,
                          timeout=timeout)


def test_validate_all_states_at_world_two():
    """--validate-states all: the WHOLE per-rank batch goes through the chunked all-gather (700 states in 100-state chunks),
    every chunk checked, the remote rank's first 512 states recomputed."""
    out = _launch(2, validate='all')
    assert out.returncode == 0, out.stderr[-2000:]
    j = json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][0])
    v = j['validation_allgather']
    assert v['ok'] is True and v['states_per_rank'] == 700 and v['gathered_bytes'] == 2 * 100 * 700 * 8
    assert v['remote_states_recomputed'] == 512 and v['remote_max_err_over_tolerance'] == 0.0


def test_every_rank_stops_when_one_has_no_library():
    """A rank without an up-to-date library: no rank compiles, none waits in a barrier -- the ranks agree (one all-reduce of a
    flag, BEFORE any barrier) and all leave with exit code 3 and ONE message; no JSON line."""
    import time
    t0 = time.time()
    out = _launch(2, extra_env={'STUB_MISSING_RANK': '1'}, timeout=300)
    assert out.returncode != 0 and time.time() - t0 < 120
    assert not [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert out.stderr.count('no up-to-date mechanism-specific library') == 1, out.stderr[-2000:]
    assert 'nothing is compiled inside one' in out.stderr


def test_process_group_gets_an_explicit_timeout():
    src = open(os.path.join(ROOT, 'bench.py')).read()
    assert "kw['timeout'] = datetime.timedelta(seconds=int(os.environ.get('PJ_DIST_TIMEOUT_S', 1800)))" in src


@pytest.mark.parametrize('world', [1, 2, 8])
def test_bench_multi_rank_branch_over_gloo(world):
    out = _launch(world)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1                   # rank 0 prints ONE JSON line
    j = json.loads(lines[0])
    assert j['n_gpus'] == world and j['steps'] == 3 and j['warmup'] == 1 and j['scaling'] == 'weak'
    v = j['validation_allgather']
    # (world size 1 -- what torch.distributed.run sets up for the N = 1 leg of a scaling run: the peer is the rank itself)
    assert v['ok'] is True and v['states_per_rank'] == 512 and v['remote_rank_checked'] == 1 % world
    assert v['remote_states_recomputed'] == 512 and v['remote_max_err_over_tolerance'] == 0.0
    assert v['gathered_bytes'] == world * 100 * 512 * 8      # ranks x NSP^2 rows x 512 states, in 100-state chunks
    assert j['evaluator'] == 'injected:stub_evaluator:make' and 'INJECTED' in j['config']['kernel']
    assert j['metric'] == 'fp64 analytical Jacobians/s' and j['config']['states_per_gpu'] == 700
    assert j['value'] > 0 and j['ms_per_step'] > 0
