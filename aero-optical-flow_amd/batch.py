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
    return gather_flows_async(local_flows, n_total, group).wait()


class PendingGather:
    """An all_gather of flow records in flight (see gather_flows_async)."""

    def __init__(self, work, out, n_total, width, world):
        self.work, self.out, self.n_total, self.width, self.world = work, out, n_total, width, world

    def wait_host(self, poll_s=20e-6, timeout_s=60.0):
        """Blocks the HOST until the collective has completed (no stream is made to wait).  Polls the
        collective's completion with a short sleep in between -- the caller only asks for gathers that
        were issued 2 G steps ago, which are done long since, so the first poll normally returns; the
        sleep keeps a late one from burning the core that launches the next steps.  A collective that
        does not complete within timeout_s raises (a hung peer must not hang this rank silently)."""
        import time
        if self.work is None or self.work.is_completed():
            return
        t0 = time.monotonic()
        while not self.work.is_completed():
            if time.monotonic() - t0 > timeout_s:
                raise TimeoutError(f"all_gather of {self.n_total} flow records did not complete within {timeout_s:.0f} s")
            time.sleep(poll_s)

    def wait(self):
        """Blocks the CURRENT STREAM (not the host, on GPUs) until the records have landed
        and returns them as [n_total, 16] in pair order."""
        import torch
        if self.work is not None:
            self.work.wait()
        if self.n_total == self.world * self.width:
            return self.out
        parts = []
        for r in range(self.world):
            rb, re_ = shard_range(self.n_total, r, self.world)
            parts.append(self.out[r * self.width:r * self.width + (re_ - rb)])
        return torch.cat(parts, dim=0)


def gather_flows_async(local_flows, n_total: int, group=None, force=False, out=None) -> PendingGather:
    """Starts the gather and returns at once, so the next batch's search can be enqueued
    while the 16-byte records cross xGMI (the collective runs on RCCL's own stream).
    `local_flows` must stay untouched until wait() -- alternate two buffers.  `out`: optional
    [world * ceil(n_total / world), 16] buffer to gather into."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not force):
        return PendingGather(None, local_flows, n_total, n_total, 1)   # (force: issue the collective with one rank all the same)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    b, e = shard_range(n_total, rank, world)
    assert local_flows.shape[0] == e - b and local_flows.shape[1] == 16
    width = (n_total + world - 1) // world
    padded = local_flows
    if e - b != width:
        padded = torch.zeros((width, 16), dtype=torch.uint8, device=local_flows.device)
        padded[:e - b] = local_flows
    if out is None:   # (a caller that gathers every step passes its own buffer: no allocation per step)
        out = torch.empty((world * width, 16), dtype=torch.uint8, device=local_flows.device)
    assert out.shape == (world * width, 16) and out.dtype == torch.uint8
    work = dist.all_gather_into_tensor(out, padded.contiguous(), group=group, async_op=True)
    return PendingGather(work, out, n_total, width, world)
