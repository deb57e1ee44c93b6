"""GPU: pyjac_amd/dist.py through RCCL (backend "nccl") on the one GPU a test box has -- world size 1.

The 8-GPU scaling run is the driver's; this makes sure its collectives are not executed for the first time there:
`all_gather_into_tensor` on the `view(-1)` buffers, the chunk loop with its ragged last chunk and the checksum gather
run through RCCL on device tensors that hold a real Jacobian shard (512 GRI-shaped states from the HIP path), and what
comes back is the local shard bit for bit (BASELINE config 4 = config 3 per rank + this)."""
import os
import socket

import numpy as np
import pytest

from conftest import MECHS

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def rccl_world1():
    import torch
    import torch.distributed as dist
    assert torch.cuda.is_available(), 'GPU tests need a GPU'
    torch.cuda.set_device(0)
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % port, rank=0, world_size=1)
    yield dist
    dist.destroy_process_group()


@pytest.mark.parametrize('name,n', [('gri30_shaped', 512), ('h2o2_n2', 1000)])
def test_dist_collectives_through_rccl_on_a_real_shard(name, n, rccl_world1):
    import torch
    import pyjac_amd
    from pyjac_amd import synth
    from pyjac_amd.dist import gather_shards, global_entry, iter_gathered, shard_checksums, shard_range
    dist = rccl_world1
    assert dist.get_backend() == 'nccl' and dist.get_world_size() == 1
    ev = pyjac_amd.Evaluator(MECHS[name])
    pres, y = synth.dist_b(n, ev.nsp, seed=5, Tlo=900, Thi=2400)
    lo, hi = shard_range(n, 0, 1)
    assert (lo, hi) == (0, n)
    local = ev.jacobian(torch.from_numpy(pres).cuda(), torch.from_numpy(np.ascontiguousarray(y)).cuda())
    torch.cuda.synchronize()
    assert local.is_cuda and local.shape == (ev.nsp * ev.nsp, n) and bool(torch.isfinite(local).all())
    # one all_gather_into_tensor of the whole shard (rank-major [world][rows][n])
    g = gather_shards(local)
    assert g.shape == (1,) + tuple(local.shape) and torch.equal(g[0], local)
    for st in (0, n // 2, n - 1):
        assert torch.equal(global_entry(g, st, n, 1), local[:, st])
    # the chunked form bench.py validates with: 77 states at a time through one receive buffer, ragged last chunk
    seen, chunks = 0, 0
    for c0, part in iter_gathered(local, 77):
        cols = part.shape[2]
        assert part.shape[:2] == (1, local.shape[0]) and cols == min(77, n - c0)
        assert torch.equal(part[0], local[:, c0:c0 + cols])
        seen += cols
        chunks += 1
    assert seen == n and chunks == -(-n // 77) and n % 77 != 0
    # checksum of checksums
    cs = shard_checksums(local)
    assert cs.shape == (1, 2)
    assert torch.equal(cs[0], torch.stack([local.sum(), (local * local).sum()]).to(torch.float64))
    # a barrier + MAX all-reduce as bench.py brackets its timed region
    t = torch.tensor([1.25], device='cuda', dtype=torch.float64)
    dist.barrier()
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert float(t) == 1.25


def test_bench_under_torchrun_at_world_size_one():
    """bench.py launched the way the driver launches the legs of a scaling run (`python -m torch.distributed.run
    --nproc-per-node N ... bench.py --gpus N`), with N = 1 on this box's one GPU: the process group is RCCL, the barriers,
    the MAX-reduce of the elapsed time, the chunked validation all-gather, the checksum gather and the recomputation of the
    peer's (= its own) states run on the device, on Jacobians from the HIP path."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', MASTER_ADDR='127.0.0.1', PJ_VALIDATE_CHUNK='1000')
    env.pop('PJ_BENCH_EVALUATOR', None)
    out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1',
                          '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.join(root, 'bench.py'),
                          '--gpus', '1', '--steps', '2', '--warmup', '1', '--states', '4096', '--no-cpu-baseline', '--no-also'],
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    j = json.loads(lines[0])
    v = j['validation_allgather']
    assert v['ok'] is True and v['states_per_rank'] == 4096 and v['remote_rank_checked'] == 0
    assert v['remote_max_err_over_tolerance'] <= 1.0 and v['gathered_bytes'] == 53 * 53 * 4096 * 8
    assert j['n_gpus'] == 1 and j['evaluator'].startswith('native') and j['config']['kernel'].startswith('pj_rblk')


def test_mechanism_handle_belongs_to_one_device():
    """A handle, its workspaces and the scratch of an attached library live on the device that was current at its first
    evaluation (one process per GPU is the supported launch, bench.py sets the device from LOCAL_RANK before anything else);
    a call with another device current is refused with PJ_EINVAL and a message that says so -- not evaluated on the wrong
    GPU's memory.  Needs two devices: skipped on the one-GPU test box, runs on the driver's 8-GPU node."""
    import torch
    import pyjac_amd
    from pyjac_amd import synth
    from pyjac_amd._lib import PyjacError
    if torch.cuda.device_count() < 2:
        pytest.skip('one visible GPU: the guard needs a second device to be current')
    torch.cuda.set_device(0)
    ev = pyjac_amd.Evaluator(MECHS['h2o2_n2'])
    pres, y = synth.dist_a(256, ev.nsp)
    j0 = ev.jacobian(torch.from_numpy(pres).cuda(), torch.from_numpy(y).cuda())
    torch.cuda.set_device(1)
    try:
        with pytest.raises(PyjacError, match='belongs to HIP device 0'):
            ev.jacobian(torch.from_numpy(pres).cuda(), torch.from_numpy(y).cuda())
        # a handle of its own on the second device gives the same Jacobians
        ev1 = pyjac_amd.Evaluator(MECHS['h2o2_n2'])
        j1 = ev1.jacobian(torch.from_numpy(pres).cuda(), torch.from_numpy(y).cuda())
        assert torch.equal(j0.cpu(), j1.cpu())
    finally:
        torch.cuda.set_device(0)
