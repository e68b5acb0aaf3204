"""Target sharding and model replication for the multi-GPU evaluation sweep.

The path shards by evaluation targets only: targets are independent, the
model is replicated with ONE broadcast from rank 0 (RBF: the weight vector; barycentric:
the packed DAG records + leaf table).  No reduction and no all-to-all exists on the path;
`gather_shards` is only for callers that want the full output on one rank.

Works with any torch.distributed backend: "nccl" (= RCCL over xGMI on ROCm) on GPUs,
"gloo" in the CPU tests.
"""
from typing import List, Tuple


def shard_bounds(m_total: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous shard [first, first+count) of rank `rank`: ceil-sized shards, the last
    ones possibly short or empty, every target covered exactly once."""
    per = (m_total + world - 1) // world
    first = min(rank * per, m_total)
    return first, max(0, min(per, m_total - first))


def shard_sizes(m_total: int, world: int) -> List[int]:
    return [shard_bounds(m_total, world, r)[1] for r in range(world)]


def broadcast_model(tensors, src: int = 0):
    """Replicate the model tensors (already allocated with identical shapes on every rank)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    for t in tensors:
        dist.broadcast(t, src)


def gather_shards(local, m_total: int, dst: int = 0):
    """Concatenate per-rank output shards on `dst` (ragged shards are padded to the ceil size)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    per = (m_total + world - 1) // world
    padded = torch.zeros(per, dtype=local.dtype, device=local.device)
    padded[: local.numel()] = local
    bucket = [torch.empty_like(padded) for _ in range(world)] if rank == dst else None
    dist.gather(padded, bucket, dst=dst)
    if rank != dst:
        return None
    return torch.cat([b[:n] for b, n in zip(bucket, shard_sizes(m_total, world))])
