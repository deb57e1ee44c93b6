"""CPU, world_size 2, gloo: sharding + validation all-gather logic of the
multi-GPU path (pyjac_amd/dist.py).  The evaluation itself is GPU-only; here the
"Jacobian" is a deterministic function of the global state index."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pyjac_amd.dist import gather_shards, global_entry, iter_gathered, shard_checksums, shard_range


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 64, 1000003):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _fake_jac(lo, hi, rows):
    s = torch.arange(lo, hi, dtype=torch.float64)
    r = torch.arange(rows, dtype=torch.float64)[:, None]
    return torch.sin(0.001 * s)[None, :] * (r + 1.0) + r * 1e-3


def _worker(rank, world, port, n, rows, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        lo, hi = shard_range(n, rank, world)
        local = _fake_jac(lo, hi, rows)
        g = gather_shards(local)
        ok = g.shape == (world, rows, n // world)
        for st in (0, n // world - 1, n // world, n - 1):
            ok &= torch.equal(global_entry(g, st, n, world), _fake_jac(st, st + 1, rows)[:, 0])
        # the same gathered batch, 77 states at a time through one receive buffer (ragged last chunk)
        seen = 0
        for c0, part in iter_gathered(local, 77):
            ok &= torch.equal(part, g[:, :, c0:c0 + part.shape[2]])
            seen += part.shape[2]
        ok &= seen == n // world
        cs = shard_checksums(local)
        ok &= torch.allclose(cs[rank], torch.stack([local.sum(), (local * local).sum()]))
        ok &= cs.shape == (world, 2)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 8])
def test_gather_reassembles_global_batch(world):
    """world 2 and 8 (the driver's scaling run is 1 / 2 / 4 / 8 ranks)."""
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, 1000, 9, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(r, True) for r in range(world)]
