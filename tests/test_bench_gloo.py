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


@pytest.mark.parametrize('world', [1, 2, 8])
def test_bench_multi_rank_branch_over_gloo(world):
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, PJ_BENCH_EVALUATOR='stub_evaluator:make', MASTER_ADDR='127.0.0.1', PJ_VALIDATE_CHUNK='100',
               PYTHONPATH=os.path.join(ROOT, 'tests') + os.pathsep + os.environ.get('PYTHONPATH', ''))
    out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(world),
                          '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.join(ROOT, 'bench.py'),
                          '--gpus', str(world), '--steps', '3', '--warmup', '1', '--workload', 'h2', '--states', '700',
                          '--validate-states', '512', '--no-cpu-baseline'], env=env, capture_output=True, text=True,
                         timeout=600)
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
