"""Multi-GPU sharding of the cull path (NEW — the reference is single-GPU).

Instances are independent (shaders/emit_draws.wgsl:38-63 touches only slot i), so rank r owns
the contiguous shard [shard_range(n, r, world)).  Each rank culls + compacts its shard with
GLOBAL base_instance values; concatenating the per-rank lists in rank order is bit-identical to
the single-GPU compaction.  Exchange = one tiny all-gather of the counts, then every rank sends
its exact-size list straight into every peer's final buffer (one-shot direct all-gather: on the
xGMI full mesh each pair has its own link, so the seven transfers run in parallel instead of
hopping around a ring).  Backend "nccl" is RCCL on ROCm; the same code runs on gloo for tests.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

DRAW_BYTES = 20


def shard_range(n: int, rank: int, world: int):
    """Contiguous ranges [r*N/G, (r+1)*N/G)."""
    return (n * rank) // world, (n * (rank + 1)) // world


def allgather_counts(local_count: torch.Tensor, group=None) -> torch.Tensor:
    """local_count: 1-element int32/int64 tensor on the compute device -> [world] int64 (host)."""
    world = dist.get_world_size(group)
    mine = local_count.reshape(1).to(torch.int64)
    out = torch.empty(world, dtype=torch.int64, device=mine.device)
    dist.all_gather_into_tensor(out, mine, group=group)
    return out.cpu()


def allgather_draws(local_draws_u8: torch.Tensor, counts: torch.Tensor, out_u8: torch.Tensor, group=None):
    """Place rank q's first counts[q] commands at byte offset 20*sum(counts[:q]) of out_u8 on
    every rank.  local_draws_u8 / out_u8 are flat uint8 tensors on the compute device."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    counts = [int(c) for c in counts]
    offs = [0]
    for c in counts:
        offs.append(offs[-1] + c)
    total = offs[-1]
    assert out_u8.numel() >= total * DRAW_BYTES
    mine = local_draws_u8[: counts[rank] * DRAW_BYTES]
    out_u8[offs[rank] * DRAW_BYTES: offs[rank + 1] * DRAW_BYTES].copy_(mine)
    if world == 1:
        return total
    ops = []
    for step in range(1, world):
        dst = (rank + step) % world
        src = (rank - step) % world
        if counts[rank]:
            ops.append(dist.P2POp(dist.isend, mine, dist.get_global_rank(group, dst) if group else dst, group))
        if counts[src]:
            view = out_u8[offs[src] * DRAW_BYTES: offs[src + 1] * DRAW_BYTES]
            ops.append(dist.P2POp(dist.irecv, view, dist.get_global_rank(group, src) if group else src, group))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return total
