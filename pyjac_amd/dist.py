"""Multi-GPU: the batch of states shards trivially.

One process per GPU; rank r of G evaluates the contiguous block of states
[r*n/G, (r+1)*n/G) with its own copy of the (KB-sized) mechanism tables.  There is
no collective on the data path.  The only exchange is the validation
all-gather of Jacobian shards (RCCL over xGMI when the backend is "nccl"),
outside any timed region: shards are gathered rank-major, ``[G][rows][n/G]``,
because a rank's SoA shard is contiguous but its place in a global SoA array
is not.
"""
from __future__ import annotations

from typing import Tuple


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block partition; the first n % world ranks get one extra state."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_shards(local, group=None):
    """All-gather equally-shaped shard tensors into a rank-major tensor
    ``[world, *local.shape]`` (one all_gather_into_tensor; in place on the
    receiving side, so the peak is the gathered buffer only)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    out = torch.empty((world,) + tuple(local.shape), dtype=local.dtype, device=local.device)
    if dist.get_backend(group) == 'gloo':
        parts = [out[r] for r in range(world)]
        dist.all_gather(parts, local.contiguous(), group=group)
    else:
        dist.all_gather_into_tensor(out.view(-1), local.contiguous().view(-1), group=group)
    return out


def iter_gathered(local, chunk_cols: int, group=None):
    """The rank-major gathered shards ``[world, rows, cols]`` of equally-shaped SoA shards ``(rows, n)``, a chunk
    of ``chunk_cols`` states at a time through ONE reusable receive buffer: yields ``(c0, g)`` with
    ``g[r] == shard of rank r[:, c0:c0 + cols]``.  The full gathered batch (8 x 22.5 GB for 8e6 GRI-size
    Jacobians) never exists at once; a chunk's send side is a ``rows x chunk_cols`` staging copy."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rows, n = local.shape
    buf = torch.empty((world, rows, min(chunk_cols, n)), dtype=local.dtype, device=local.device)
    for c0 in range(0, n, chunk_cols):
        cols = min(chunk_cols, n - c0)
        send = local[:, c0:c0 + cols].contiguous()
        out = buf if cols == buf.shape[2] else torch.empty((world, rows, cols), dtype=local.dtype, device=local.device)
        if dist.get_backend(group) == 'gloo':
            dist.all_gather([out[r] for r in range(world)], send, group=group)
        else:
            dist.all_gather_into_tensor(out.view(-1), send.view(-1), group=group)
        yield c0, out


def global_entry(gathered, state: int, n: int, world: int):
    """Column of the gathered rank-major SoA buffer holding global state index
    ``state`` (all rows)."""
    base, rem = divmod(n, world)
    for r in range(world):
        lo, hi = shard_range(n, r, world)
        if lo <= state < hi:
            return gathered[r][:, state - lo]
    raise IndexError(state)


def shard_checksums(local, group=None):
    """fp64 (sum, sum of squares) per shard, all-gathered: a checksum of
    checksums proving each rank saw every other rank's bytes."""
    import torch
    cs = torch.stack([local.sum(), (local * local).sum()]).to(torch.float64)
    return gather_shards(cs, group)
