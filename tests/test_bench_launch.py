"""`python bench.py --gpus N` must start from a PLAIN command line: with N > 1 and no
WORLD_SIZE it launches the ranks itself (torch.distributed.run on 127.0.0.1) before anything
touches a GPU.  The CPU rehearsal (--rendezvous-only) runs the whole N > 1 control flow --
rendezvous, shard_range, the gather of the 16-byte flow records over gloo, pair order checked
on every rank -- with placeholder records."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*args):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True,
                         timeout=300, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


@pytest.mark.parametrize("scaling,pairs,total", [("strong", 7, 7), ("weak", 5, 10), ("strong", 1024, 1024)])
def test_plain_invocation_spawns_two_ranks(scaling, pairs, total):
    j = run_bench("--gpus", "2", "--rendezvous-only", "--scaling", scaling, "--pairs", str(pairs))
    devices = j.pop("devices")
    assert len(devices) == 2 and devices[0].startswith("rank 0:") and devices[1].startswith("rank 1:")
    assert j == {"rendezvous_only": True, "n_ranks": 2, "global_pairs": total, "scaling": scaling,
                 "gathered_in_pair_order_on_every_rank": True, "ranks_seen": 2,
                 "configs3": {"global_pairs": 1024, "pairs_per_gpu": 512, "scaling": "strong",
                              "gathered_in_pair_order_on_every_rank": True}}


def test_eight_ranks_rehearse_configs3_on_cpu():
    """BASELINE configs[3] as the driver would start it: 1 024 pairs sharded over EIGHT ranks, 128 each,
    the records gathered in pair order on every rank (gloo, no GPU)."""
    j = run_bench("--gpus", "8", "--rendezvous-only", "--scaling", "strong", "--pairs", "1024")
    devices = j.pop("devices")
    assert [d.split(":")[0] for d in devices] == [f"rank {r}" for r in range(8)]
    assert j == {"rendezvous_only": True, "n_ranks": 8, "global_pairs": 1024, "scaling": "strong",
                 "gathered_in_pair_order_on_every_rank": True, "ranks_seen": 8,
                 "configs3": {"global_pairs": 1024, "pairs_per_gpu": 128, "scaling": "strong",
                              "gathered_in_pair_order_on_every_rank": True}}
    # the driver's default command (weak scaling, 1 024 pairs per GPU) carries configs[3]'s shape all the same
    k = run_bench("--gpus", "8", "--rendezvous-only")
    assert k["scaling"] == "weak" and k["global_pairs"] == 8192 and k["ranks_seen"] == 8
    assert k["configs3"] == j["configs3"]


def test_single_rank_needs_no_launcher():
    j = run_bench("--rendezvous-only", "--pairs", "3")
    assert j["n_ranks"] == 1 and j["global_pairs"] == 3 and j["gathered_in_pair_order_on_every_rank"]


@pytest.mark.gpu
def test_plain_two_rank_bench_line_on_one_gpu():
    """The real N = 2 bench path from a plain command line: two ranks share the one GPU of the
    box (gloo for the gather of the flow records), max-over-ranks timing, one JSON line."""
    j = run_bench("--gpus", "2", "--backend", "gloo", "--steps", "2", "--warmup", "1", "--settle-steps", "3",
                  "--pairs", "8", "--cpu-seconds", "0", "--configs3-pairs", "64")
    assert j["n_gpus"] == 2 and j["config"]["global_pairs"] == 16 and j["steps"] == 2 and j["settle_steps"] == 3
    assert j["value"] > 0 and j["parity"]["oracle_pairs_bit_exact"] and j["parity"]["all_pairs_return_known_shift"]
    # the weak line carries the strong shape beside it: 64 pairs over the two ranks against 64 on one GPU
    assert j["ranks_seen"] == 2 and len(j["devices"]) == 2 and all("cuda:0" in d for d in j["devices"])
    c3 = j["configs3"]
    assert c3["global_pairs"] == 64 and c3["pairs_per_gpu"] == 32 and c3["scaling"] == "strong"
    assert c3["ms_per_step"] > 0 and c3["one_gpu_ms_per_step"] > 0 and c3["vs_one_gpu_1024"] > 0
    k = run_bench("--gpus", "2", "--backend", "gloo", "--steps", "2", "--warmup", "1", "--settle-steps", "3",
                  "--pairs", "16", "--scaling", "strong", "--cpu-seconds", "0", "--configs3-pairs", "16")
    assert k["scaling"] == "strong" and k["config"]["global_pairs"] == 16 and k["config"]["pairs_per_gpu"] == 8
    assert k["configs3"]["pairs_per_gpu"] == 8 and k["configs3"]["ms_per_step"] == pytest.approx(k["ms_per_step"], rel=1e-3)


@pytest.mark.gpu
def test_four_ranks_with_two_batches_in_flight_on_one_gpu():
    """configs[3]'s per-GPU share as the strong-scaling line runs it: 128 VGA pairs per rank, two batches
    in flight per rank (bench lanes on two HIP streams), the reduction inside the search launch, the
    step replayed as a hipGraph -- four ranks sharing the one GPU of the box, gloo for the gather."""
    j = run_bench("--gpus", "4", "--backend", "gloo", "--steps", "4", "--warmup", "1", "--settle-steps", "4",
                  "--pairs", "512", "--scaling", "strong", "--cpu-seconds", "0", "--configs3-pairs", "512")
    c = j["config"]
    c3 = j["configs3"]
    assert j["ranks_seen"] == 4 and c3["pairs_per_gpu"] == 128 and c3["streams"] == 2 and c3["reduce"] == "fused" and c3["graph_replay"]
    assert c3["one_gpu_ms_per_step"] > 0 and c3["vs_one_gpu_1024"] > 0
    assert j["n_gpus"] == 4 and c["global_pairs"] == 512 and c["pairs_per_gpu"] == 128
    assert c["streams"] == 2 and c["reduce"] == "fused" and c["graph_replay"]
    assert j["value"] > 0 and j["parity"]["oracle_pairs_bit_exact"] and j["parity"]["all_pairs_return_known_shift"]


@pytest.mark.gpu
def test_one_rank_rccl_rehearsal_of_the_gather_path():
    """`bench.py --force-dist`: the N > 1 code path with ONE rank on the one-GPU box -- RCCL initialised,
    the step captured into a hipGraph beside its watchdog thread, the flow records of G steps shipped by
    one all_gather_into_tensor, completion checked from the host -- for the headline step and for
    configs[3]'s per-GPU share (two lanes, reduction in the launch)."""
    j = run_bench("--force-dist", "--steps", "40", "--warmup", "2", "--settle-steps", "20", "--cpu-seconds", "0")
    assert j["config"]["streams"] == 1 and "every 4 steps" in j["config"]["parallelism"]
    assert j["parity"]["oracle_pairs_bit_exact"] and j["parity"]["all_pairs_return_known_shift"]
    # the keys the first real 8-GPU line will be read by: ranks and devices as the process group saw them,
    # and configs[3]'s strong shape (here: 1 024 pairs on the one rank, against themselves)
    assert j["ranks_seen"] == 1 and len(j["devices"]) == 1 and "gfx950" in j["devices"][0]
    c3 = j["configs3"]
    assert c3["global_pairs"] == 1024 and c3["pairs_per_gpu"] == 1024 and c3["scaling"] == "strong"
    assert 0.8 < c3["vs_one_gpu_1024"] < 1.25, c3   # one rank: the sharded batch IS the one-GPU batch (+ the gather)
    k = run_bench("--force-dist", "--pairs", "128", "--steps", "64", "--warmup", "2", "--settle-steps", "20", "--cpu-seconds", "0")
    assert k["config"]["streams"] == 2 and k["config"]["graph_replay"] and "every 16 steps" in k["config"]["parallelism"]
    assert k["parity"]["oracle_pairs_bit_exact"] and k["parity"]["all_pairs_return_known_shift"]
