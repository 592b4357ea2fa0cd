"""Batched mode over several GPUs: independent frame pairs are sharded statically,
one process per GPU, and the ONLY exchange is the final gather of the 16-byte flow
records (RCCL over xGMI on GPUs; gloo in the CPU tests).  SURVEY.md section 8e."""
from __future__ import annotations


def shard_range(n_total: int, rank: int, world: int):
    """Contiguous shard [begin, end) of rank; the remainder goes to the first ranks."""
    assert 0 <= rank < world and n_total >= 0
    base, rem = divmod(n_total, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def gather_flows(local_flows, n_total: int, group=None):
    """all_gather of per-rank flow records ([n_local, 16] uint8) into [n_total, 16], in
    pair order, on every rank.  Shards may be ragged: records are padded to the largest
    shard for the collective and trimmed afterwards."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local_flows
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    b, e = shard_range(n_total, rank, world)
    assert local_flows.shape[0] == e - b and local_flows.shape[1] == 16
    width = (n_total + world - 1) // world
    padded = local_flows
    if e - b != width:
        padded = torch.zeros((width, 16), dtype=torch.uint8, device=local_flows.device)
        padded[:e - b] = local_flows
    out = torch.empty((world * width, 16), dtype=torch.uint8, device=local_flows.device)
    dist.all_gather_into_tensor(out, padded.contiguous(), group=group)
    if n_total == world * width:
        return out
    parts = []
    for r in range(world):
        rb, re_ = shard_range(n_total, r, world)
        parts.append(out[r * width:r * width + (re_ - rb)])
    return torch.cat(parts, dim=0)
