"""world_size-2 gloo tests of the batched multi-GPU mode (CPU, no GPU): static
sharding of independent pairs + the single gather of 16-byte flow records.
The per-shard compute is stood in for by the oracle (this is tests/, the only
place allowed to call it); on GPUs the same code path runs the HIP engine."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n_total, out_dir):
    sys.path.insert(0, ROOT)
    import importlib
    import __graft_entry__ as ge
    ge.load_package()
    batch = importlib.import_module("aero_optical_flow_amd.batch")
    synth = importlib.import_module("aero_optical_flow_amd.synth")
    from oracle import pyoracle as orc
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        p = orc.default_params(64, 64)
        b, e = batch.shard_range(n_total, rank, world)
        prevs, curs, _ = synth.make_batch(64, 64, e - b, 4, first_index=b, noise=2)
        _, flows, _ = orc.flow_batch(p, prevs, curs, threads=1) if e > b else (None, np.zeros(0, orc.FLOW_DTYPE), 1)
        local = torch.from_numpy(flows.view(np.uint8).reshape(e - b, 16).copy())
        full = batch.gather_flows(local, n_total)
        assert full.shape == (n_total, 16)
        # pipelined form used by bench.py: start, do other work, then wait
        pending = batch.gather_flows_async(local.clone(), n_total)
        again = pending.wait()
        assert torch.equal(again, full)
        # the form bench.py uses for N > 1: the records of G steps in one ring segment, ONE collective
        # into a buffer of the caller, completion checked from the host (no stream waits)
        if n_total % world == 0 and n_total > 0:
            G, per = 3, n_total // world
            seg = torch.stack([local + g for g in range(G)]).view(G * per, 16)   # (byte values wrap: still a unique tag per step)
            out = torch.zeros((world * G * per, 16), dtype=torch.uint8)
            pend = batch.gather_flows_async(seg, world * G * per, out=out)
            pend.wait_host()
            got = out.view(world, G, per, 16)
            for g in range(G):
                assert torch.equal(got[:, g].reshape(n_total, 16), full + g), g
        np.save(os.path.join(out_dir, f"rank{rank}.npy"), full.numpy())
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [8, 7, 1])
def test_sharded_batch_gathers_in_pair_order(tmp_path, orc, synth, n_total):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), n_total, str(tmp_path)), nprocs=world, join=True)
    prevs, curs, _ = synth.make_batch(64, 64, n_total, 4, first_index=0, noise=2)
    _, ref, _ = orc.flow_batch(orc.default_params(64, 64), prevs, curs, threads=1)
    ref = ref.view(np.uint8).reshape(n_total, 16)
    for r in range(world):
        got = np.load(tmp_path / f"rank{r}.npy")
        assert np.array_equal(got, ref), f"rank {r} sees a different gathered batch"


def test_shard_range_partitions_exactly(aof):
    import importlib
    batch = importlib.import_module("aero_optical_flow_amd.batch")
    for n in (0, 1, 7, 8, 1024, 1025):
        for world in (1, 2, 3, 8):
            spans = [batch.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1
