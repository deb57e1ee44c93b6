"""CPU, world_size 2, gloo: bench.py's own multi-rank branch (launch through torch.distributed.run,
partition, barriers, MAX-reduce of the elapsed time, validation all-gather + checksums) with a CPU
stand-in for the evaluator (PJ_BENCH_STUB=1), so that the first 8-GPU run of the driver is not also the
first execution of that code."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_multi_rank_branch_over_gloo():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, PJ_BENCH_STUB='1', MASTER_ADDR='127.0.0.1')
    out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
                          '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.join(ROOT, 'bench.py'),
                          '--gpus', '2', '--steps', '3', '--warmup', '1'], env=env, capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1                   # rank 0 prints ONE JSON line
    j = json.loads(lines[0])
    assert j['n_gpus'] == 2 and j['steps'] == 3 and j['warmup'] == 1 and j['scaling'] == 'weak'
    assert j['validation_allgather']['ok'] is True
    assert j['value'] > 0 and j['ms_per_step'] > 0
