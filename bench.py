#!/usr/bin/env python3
"""Benchmark of the flow hot path: frame-pairs/s at N GPUs, % of the HBM roofline.

A "step" = one pass of the hot path (SAD search K2 + histogram reduce K3, plus
K1 and the level-1 passes when the workload has a pyramid) over one batch of
synthetic frame pairs already resident in HBM.  Weak scaling: every rank holds
--pairs frame pairs; with N > 1 the ranks run independently and only exchange
the 16-byte per-pair flow records (one RCCL all_gather per step).

Prints ONE JSON line (rank 0).  `roofline` prices the dominant kernel (K2) with
SURVEY.md section 8d's algorithmic bytes per pair against 8 TB/s; `cpu_baseline`
times the CPU oracle (this repo's restatement -- the reference's own engine
source is absent, so kind = "port") on the same workload on the host cores.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec

WORKLOADS = {
    # name: (description, width, height, params overrides, synthetic reach)
    "c2": ("C2 640x480 grey pairs, 8x8 SAD, +-4 search, dense grid (4661 blocks/pair)",
           640, 480, dict(), 4),
    "c3": ("C3 640x480, 2-level mean-subtracted pyramid + 4x4 gate/histogram filter",
           640, 480, dict(pyramid_levels=2, mean_subtract=1), 9),
    "c3n": ("C3 without the mean equalisation (pyramid + predictor only)",
            640, 480, dict(pyramid_levels=2), 9),
    "c2h": ("C2 geometry with half-pixel refinement on the dense grid (origin 5, 4582 blocks/pair)",
            640, 480, dict(subpixel=1), 4),
    "c1b": ("C1 geometry in batch: 64x64, published sparse 5x5 grid, half-pixel refinement",
            64, 64, dict(_px4flow=1), 4),
    "c5": ("C5 1280x960 pairs, 16x16 SAD, +-8 search",
           1280, 960, dict(tile=16, search=8, value_threshold=12000), 8),
    "c4k": ("3840x2160 grey pairs, 8x8 SAD, +-4 search, dense grid",
            3840, 2160, dict(), 4),
    "c5p": ("C5 geometry with the 2-level mean-subtracted pyramid",
            1280, 960, dict(tile=16, search=8, value_threshold=12000, pyramid_levels=2, mean_subtract=1), 17),
    "c5h": ("C5 geometry with half-pixel refinement (origin 9)",
            1280, 960, dict(tile=16, search=8, value_threshold=12000, subpixel=1), 8),
}


def make_batch_gpu(width, height, n, reach, seed, device, brightness=0, noise=0, contrast=1.0, half=False):
    """Same recipe as aero-optical-flow_amd/synth.py, generated on the GPU: blurred random
    canvas cropped twice at an integer shift.  Returns prev, cur (u8 [n,H,W]) and shifts.
    noise / contrast / half as synth.make_pair: +-noise LSB on cur, cur's contrast about 128, and an extra
    half-pixel displacement per pair drawn from {-1, 0, 1}^2 (cur averaged with its one-pixel neighbour;
    the integer shifts then stay one short of the reach)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    prev = torch.empty((n, height, width), dtype=torch.uint8, device=device)
    cur = torch.empty_like(prev)
    lim = reach - 1 if half else reach
    shifts = torch.randint(-lim, lim + 1, (n, 2), generator=g, device=device)
    hs = shifts.cpu().numpy()
    hh = torch.randint(-1, 2, (n, 2), generator=g, device=device).cpu().numpy() if half else None
    chunk = 64
    for s in range(0, n, chunk):
        m = min(chunk, n - s)
        raw = torch.randint(0, 256, (m, height + 2 * reach + 2, width + 2 * reach + 2),
                            generator=g, device=device, dtype=torch.int32)
        hc, wc = height + 2 * reach, width + 2 * reach
        acc = torch.zeros((m, hc, wc), dtype=torch.int32, device=device)
        for oy in range(3):
            for ox in range(3):
                acc += raw[:, oy:oy + hc, ox:ox + wc]
        canvas = ((acc + 4) // 9).to(torch.uint8)
        prev[s:s + m] = canvas[:, reach:reach + height, reach:reach + width]
        for i in range(m):
            dx, dy = int(hs[s + i, 0]), int(hs[s + i, 1])
            c = canvas[i, reach - dy:reach - dy + height, reach - dx:reach - dx + width]
            if half or contrast != 1.0 or noise or brightness:
                c = c.to(torch.int32)
                if half and (hh[s + i, 0] or hh[s + i, 1]):
                    hx, hy = int(hh[s + i, 0]), int(hh[s + i, 1])
                    c = (c + canvas[i, reach - dy - hy:reach - dy - hy + height, reach - dx - hx:reach - dx - hx + width].to(torch.int32)) >> 1
                if contrast != 1.0:
                    c = torch.round((c - 128).to(torch.float32) * contrast + 128).to(torch.int32)
                if noise:
                    c = c + torch.randint(-noise, noise + 1, c.shape, generator=g, device=device, dtype=torch.int32)
                c = (c + brightness).clamp_(0, 255).to(torch.uint8)
            cur[s + i] = c
    return prev, cur, hs


def host_cores():
    """CPU cores this process may actually use: the cgroup quota when there is one
    (the GPU box gives each GPU a 16-core share of a 256-thread host)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def spawn_ranks(n):
    """Re-launch this command line under torch.distributed.run with n ranks on this node."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def rendezvous_only(args, rank, world):
    """The N>1 control flow of the batched mode with placeholder records and no GPU: every rank
    fills its shard's 16-byte records with the GLOBAL pair index, the ranks gather them, and
    every rank checks that it sees all pairs in order -- for the shape the command line asks for
    and for the strong shape every N > 1 line also carries (configs3)."""
    import importlib
    import socket
    import torch.distributed as dist
    ge.load_package()
    batch = importlib.import_module(ge.PKG_NAME + ".batch")
    if world > 1:
        dist.init_process_group("gloo")

    def gathered_in_order(total):
        b, e = batch.shard_range(total, rank, world)
        local = torch.arange(b, e, dtype=torch.int32).view(-1, 1).repeat(1, 4).contiguous().view(torch.uint8).view(-1, 16)
        full = batch.gather_flows(local, total)
        ok = bool(torch.equal(full.view(torch.int32).view(-1, 4)[:, 0], torch.arange(total, dtype=torch.int32)))
        if world > 1:
            t = torch.tensor([1 if ok else 0])
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            ok = bool(t.item())
        return ok, e - b

    total = args.pairs if args.scaling == "strong" else args.pairs * world
    ok, _ = gathered_in_order(total)
    line = {"rendezvous_only": True, "n_ranks": world, "global_pairs": total, "scaling": args.scaling,
            "gathered_in_pair_order_on_every_rank": ok}
    if world > 1:
        line["ranks_seen"] = dist.get_world_size()
        names = [None] * world
        dist.all_gather_object(names, f"rank {rank}: cpu ({socket.gethostname()}, no GPU touched)")
        line["devices"] = names
        if args.configs3_pairs > 0 and args.configs3_pairs % world == 0:
            ok3, mine = gathered_in_order(args.configs3_pairs)
            line["configs3"] = {"global_pairs": args.configs3_pairs, "pairs_per_gpu": mine, "scaling": "strong",
                                "gathered_in_pair_order_on_every_rank": ok3}
            ok = ok and ok3
        else:
            line["configs3"] = None
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(line), flush=True)
    return 0 if ok else 1


def bench_c1(args, aof, rank, world, dist):
    """configs[0]: the reference's own call shape -- ONE 64x64 frame per calcFlow() call from
    a host buffer through the C++ facade (PX4Flow sparse grid, half-pixel refinement).
    Latency-bound and PCIe-inclusive by construction: every call copies the frame in, runs
    two kernels and copies 16 bytes back.  Step = --pairs consecutive calls."""
    import importlib
    synth = importlib.import_module(ge.PKG_NAME + ".synth")
    frames, _ = synth.make_sequence(64, 64, 64, 4, seed=1, max_step=3)
    flow = aof.OpticalFlowPX4(216.6677, 216.2457, 15, 64, 64)
    calls = args.pairs
    t_us = [0]

    def step():
        for k in range(calls):
            flow.calcFlow(frames[k & 63], t_us[0])
            t_us[0] = (t_us[0] + 13333) & 0xFFFFFFFF

    for _ in range(args.warmup):
        step()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    elapsed = time.perf_counter() - t0
    out = {"metric": "frames/s through calcFlow() (64x64, 8x8 SAD, +-4 search, one host frame per call)",
           "value": round(calls * args.steps / elapsed, 1), "unit": "frames/s", "n_gpus": 1, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
           "config": {"workload": "C1 64x64 frames, PX4Flow 5x5 sparse grid, facade OpticalFlowPX4::calcFlow, "
                                  "host buffers (PCIe-inclusive)", "calls_per_step": calls,
                      "us_per_call": round(elapsed / (args.steps * calls) * 1e6, 2)}}
    # the class mainloop.cpp:423 actually creates, at the reference's default crop (main.cpp:57-58):
    # OpticalFlowOpenCV, 128x128, two levels + mean equalisation
    frames2, _ = synth.make_sequence(128, 128, 64, 8, seed=2, max_step=6)
    flow2 = aof.OpticalFlowOpenCV(216.6677, 216.2457, 15, 128, 128)
    for k in range(200):
        flow2.calcFlow(frames2[k & 63], (k * 13333) & 0xFFFFFFFF)
    t0 = time.perf_counter()
    n2 = max(1000, calls)
    for k in range(n2):
        flow2.calcFlow(frames2[k & 63], ((200 + k) * 13333) & 0xFFFFFFFF)
    out["config"]["opencv_facade_128x128_two_levels_us_per_call"] = round((time.perf_counter() - t0) / n2 * 1e6, 2)
    # the same two call shapes served by the resident kernel (setResidentKernel: no launch per call)
    for key, mk, fr, wh in (("resident_kernel_us_per_call", aof.OpticalFlowPX4, frames, (64, 64)),
                            ("opencv_facade_128x128_two_levels_resident_kernel_us_per_call", aof.OpticalFlowOpenCV, frames2, (128, 128))):
        f3 = mk(216.6677, 216.2457, 15, *wh)
        f3.setResidentKernel(True)
        for k in range(200):
            f3.calcFlow(fr[k & 63], (k * 13333) & 0xFFFFFFFF)
        t0 = time.perf_counter()
        for k in range(n2):
            f3.calcFlow(fr[k & 63], ((200 + k) * 13333) & 0xFFFFFFFF)
        out["config"][key] = round((time.perf_counter() - t0) / n2 * 1e6, 2)
        f3.close()
    if rank == 0:
        from oracle import pyoracle as orc
        o = orc.Px4(orc.px4flow_params(64, 64), 216.6677, 216.2457, 15)
        simd = orc.fast_sad_available()
        orc.set_fast_sad(simd)   # timing leg: the CPU's own SAD instruction
        t1 = time.perf_counter()
        m, t = 0, 0
        try:
            while time.perf_counter() - t1 < min(args.cpu_seconds, 5.0):
                o.calc_flow(frames[m & 63], t)
                t = (t + 13333) & 0xFFFFFFFF
                m += 1
        finally:
            orc.set_fast_sad(False)
        spent = time.perf_counter() - t1
        if m:
            out["cpu_baseline"] = {"value": round(m / spent, 1), "unit": "frames/s", "cores": 1, "kind": "port",
                                   "sample": f"{m} calls of the oracle's calcFlow on the same sequence "
                                             f"(SAD via {'SSE2 psadbw' if simd else 'the byte loop'}), {spent:.1f} s"}
        print(json.dumps(out), flush=True)


def bench_ingest(args, aof, device, rank, world, dist):
    """Caller-side row of the scope table: sensor frame -> centre crop +
    auto-exposure histogram on the device.  Step = one launch over --pairs*8 resident frames."""
    cam_w, cam_h, crop = 640, 480, 128
    n = args.pairs * 8
    g = torch.Generator(device=device)
    g.manual_seed(77 + rank)
    cam = torch.randint(0, 256, (n, cam_h, cam_w), generator=g, device=device, dtype=torch.uint8)
    cropped = torch.empty((n, crop, crop), dtype=torch.uint8, device=device)
    hist = torch.empty((n, 10), dtype=torch.int32, device=device)
    for _ in range(args.warmup):
        aof.ingest_batch(cam, crop, crop, cropped=cropped, hist=hist)
    torch.cuda.synchronize(device)
    if world > 1:
        dist.barrier()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for a, b in ev:
        a.record()
        aof.ingest_batch(cam, crop, crop, cropped=cropped, hist=hist)
        b.record()
    torch.cuda.synchronize(device)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    k_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))  # memset + kernel on torch's stream
    alg = 2 * crop * crop + 40
    achieved = alg * n / (k_ms * 1e-3) / 1e9
    out = {"metric": "sensor frames/s (640x480 -> 128x128 centre crop + 10-bin exposure histogram)",
           "value": round(world * n * args.steps / elapsed, 1), "unit": "frames/s", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
           "config": {"workload": "ingest: 640x480 sensor frames, crop 128x128, mask 128x128", "frames_per_gpu": n},
           "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": pmc_traffic(f"ingest:{n}")[0],
                        "traffic_source": pmc_traffic(f"ingest:{n}")[1], "kernel": "k_ingest (K0)",
                        "kernel_ms": round(k_ms, 5), "algorithmic_bytes_per_frame": alg, "frames_per_launch": n}}
    if rank == 0:
        from oracle import pyoracle as orc
        hc = cam[:4].cpu().numpy()
        ok = True
        for i in range(4):
            ec, eh = orc.ingest(hc[i], crop, crop)
            ok &= bool(np.array_equal(cropped[i].cpu().numpy(), ec) and
                       np.array_equal(hist[i].cpu().numpy().view(np.uint32), eh))
        out["parity"] = {"oracle_frames_bit_exact": ok, "frames_checked": 4}
        if world == 1 and args.cpu_seconds > 0:
            m = min(n, 2048)
            hc = cam[:m].cpu().numpy()
            done, spent = 0, 0.0
            while spent < min(args.cpu_seconds, 5.0):
                t1 = time.perf_counter()
                for i in range(m):
                    orc.ingest(hc[i], crop, crop)
                spent += time.perf_counter() - t1
                done += m
            out["cpu_baseline"] = {"value": round(done / spent, 1), "unit": "frames/s", "cores": 1, "kind": "port",
                                   "sample": f"{done} frames, scalar C oracle restating mainloop.cpp:295-298,203-214, "
                                             f"one thread, {spent:.1f} s"}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def bench_sequence(args, aof, device, rank, world, dist):
    """The reference's whole per-frame loop (mainloop.cpp:295-373) over a recorded sequence as ONE device
    pipeline (aof_sequence_device): --pairs * 64 sensor frames of 320x240 (main.cpp:54-55 defaults) -> 128x128
    crop -> OpticalFlowOpenCV's configuration (two levels + mean equalisation) -> rate limiter at 15 Hz ->
    de-rotation -> OPTICAL_FLOW_RAD frames.  Step = one call over the resident sequence."""
    import importlib
    synth = importlib.import_module(ge.PKG_NAME + ".synth")
    cam_w, cam_h, crop = 320, 240, 128
    n = args.pairs * 64
    base, _ = synth.make_sequence(cam_w, cam_h, 96, 6, seed=9 + rank, max_step=4)
    cam = torch.from_numpy(base).to(device)[torch.arange(n, device=device) % 96].contiguous()
    g = torch.Generator(device=device)
    g.manual_seed(12 + rank)
    times = torch.cumsum(torch.randint(12500, 14200, (n,), generator=g, device=device, dtype=torch.int64), 0)
    times = times - times[0]
    gyro = torch.randn((n, 4), generator=g, device=device, dtype=torch.float32) * 0.003
    gyro[:, 3] = 0.0133
    p = aof.px4flow_params(crop, crop, pyramid_levels=2, mean_subtract=1)
    eng = aof.FlowEngine(p, device.index or 0)
    sp = aof.sequence_params(cam_w, cam_h, crop, crop, 216.6677, 216.2457, 15, 5_000_000, 1, 100, 0, derotate=(9.5, 0.01))
    ws, L = eng.sequence(sp, cam, times, gyro)
    for _ in range(max(args.warmup, 3)):
        eng.sequence(sp, cam, times, gyro, workspace=ws)
    torch.cuda.synchronize(device)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.sequence(sp, cam, times, gyro, workspace=ws)
    torch.cuda.synchronize(device)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    out = eng.sequence_outputs(sp, ws, L, n)
    step_ms = elapsed / args.steps * 1e3
    # compulsory bytes per frame: the crop region of the sensor frame read once, the cropped frame written once
    # and read twice by the flow (it is cur of one pair and prev of the next), 16 B flow record, ~56 B of frame
    # per publication
    alg = crop * crop * 4 + 40 + 16 + 8
    achieved = alg * n / (step_ms * 1e-3) / 1e9
    line = {"metric": "sensor frames/s through the whole per-frame loop on the device (320x240 -> 128x128 crop, 8x8 SAD +-4 on two "
                      "levels, 15 Hz rate limiter, de-rotation, OPTICAL_FLOW_RAD frames)",
            "value": round(world * n * args.steps / elapsed, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": max(args.warmup, 3), "ms_per_step": round(step_ms, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "seq: aof_sequence_device over a resident recording", "frames_per_gpu": n,
                       "records_published": int(len(out["records"])), "frames_sent": out["frames_sent"],
                       "limiter_rounds": int(np.ceil(np.log(n + 1) / np.log(4)))},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None, "kernel": "whole pipeline (all kernels of one call)",
                         "kernel_ms": round(step_ms, 5), "algorithmic_bytes_per_frame": alg, "frames_per_launch": n}}
    if rank == 0:
        from oracle import pyoracle as orc
        # parity on a prefix: the oracle's calcFlow chain over the same cropped frames
        m = min(n, 600)
        o = orc.Px4(orc.params_from(p), 216.6677, 216.2457, 15)
        cropped = out["cropped"][:m]
        th = times[:m].cpu().numpy()
        ok, k_rec = True, 0
        for k in range(m):
            q, dt, ax, ay = o.calc_flow(cropped[k], int(th[k]))
            if q < 0:
                continue
            r = out["records"][k_rec]
            ok &= (int(r["frame"]), int(r["quality"]), int(r["dt_us"])) == (k, q, dt)
            ok &= np.float32(ax).tobytes() == r["flow_x"].tobytes() and np.float32(ay).tobytes() == r["flow_y"].tobytes()
            k_rec += 1
        line["parity"] = {"oracle_records_bit_exact": bool(ok), "frames_checked": m, "records_checked": k_rec,
                          "note": "oracle = this repo's CPU restatement (upstream PX4 source unavailable)"}
        if world == 1 and args.cpu_seconds > 0:
            o2 = orc.Px4(orc.params_from(p), 216.6677, 216.2457, 15)
            simd = orc.fast_sad_available()
            orc.set_fast_sad(simd)
            t1 = time.perf_counter()
            done = 0
            try:
                while time.perf_counter() - t1 < min(args.cpu_seconds, 10.0):
                    o2.calc_flow(cropped[done % m], int(th[done % m]))
                    done += 1
            finally:
                orc.set_fast_sad(False)
            spent = time.perf_counter() - t1
            line["cpu_baseline"] = {"value": round(done / spent, 1), "unit": "frames/s", "cores": 1, "kind": "port",
                                    "sample": f"{done} calcFlow calls of the oracle on the cropped frames (crop, limiter and angle "
                                              f"included; SAD via {'SSE2 psadbw' if simd else 'the byte loop'}), {spent:.1f} s"}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def being_profiled():
    """A profiler is already wrapped around this process (its library is preloaded): no nested profiler runs."""
    if any(k.startswith(("ROCPROF", "ROCP_", "ROCTRACER")) for k in os.environ):
        return True
    return "rocprof" in os.environ.get("LD_PRELOAD", "") or "roctracer" in os.environ.get("LD_PRELOAD", "")


def live_traffic(args, n, search_flag):
    """HBM bytes per launch of the dominant kernel and of the whole step, MEASURED now on this box: two child runs of
    this same command (a handful of steps) under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` -- separate passes,
    only with --kernel-trace, the program itself after `--`, as /opt/skills/guides/MI355X_MICROARCH.md prescribes --
    summarised with its gfx950 correction (FETCH_SIZE x 2 for wide coalesced reads) by tools/pmc_summary.py.
    Returns (kernel_bytes, step_bytes, source) or None when the profiler is not usable here (the caller then falls
    back to the committed summary of the same command)."""
    import importlib.util
    import shutil
    import subprocess
    import tempfile
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof) or being_profiled():
        return None
    spec = importlib.util.spec_from_file_location("pmc_summary", os.path.join(ROOT, "tools", "pmc_summary.py"))
    if spec is None or not os.path.exists(os.path.join(ROOT, "tools", "pmc_summary.py")):
        return None
    pmc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pmc)
    tmp = tempfile.mkdtemp(prefix="aof_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    child = [sys.executable, os.path.abspath(__file__), "--workload", args.workload, "--pairs", str(args.pairs), "--steps", "4",
             "--warmup", "1", "--settle-steps", "0", "--cpu-seconds", "0", "--traffic", "file", "--streams", "1", "--graph", "off",
             "--legs", "none", "--input", args.input, "--noise", str(args.noise)] + search_flag
    if args.brightness is not None:
        child += ["--brightness", str(args.brightness)]
    if args.max_shift is not None:
        child += ["--max-shift", str(args.max_shift)]
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            subprocess.run([rocprof, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", os.path.join(tmp, counter), "--"] + child,
                           cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=120, check=True)
        ent = pmc.summarise(tmp, args.workload, n)
    except Exception as e:
        ent = None
        live_traffic.error = f"{type(e).__name__}: {e}"[:200]
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    if not ent:
        return None
    return ent["hbm_bytes_per_launch"], ent["step_bytes"], (f"measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE around two child runs of this "
                                                           f"command (kernel {ent['kernel']}; FETCH_SIZE x 2 per the gfx950 correction)")


def pmc_traffic(key, field="hbm_bytes_per_launch"):
    """HBM bytes per launch from the committed PMC summary of the same command (profiles/), and its source.
    field "step_bytes": all kernels of one step instead of the dominant kernel alone."""
    tfile = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tfile):
        try:
            ent = json.load(open(tfile)).get(key)
            if ent and field in ent:
                return ent[field], ent.get("source", "profiles/pmc_traffic.json")
        except Exception as e:
            return None, f"none: profiles/pmc_traffic.json unreadable ({e})"
    return None, f"none: profiles/pmc_traffic.json has no entry {key}"


def bench_derotate(args, aof, device, rank, world, dist):
    """Output-side row of the scope table: gyro de-rotation of a batch of flow
    records on the device.  Step = one launch over --pairs * 1024 records (default 1 Mi)."""
    n = args.pairs * 1024
    g = torch.Generator(device=device)
    g.manual_seed(5 + rank)
    flows = torch.zeros((n, 16), dtype=torch.uint8, device=device)
    flows.view(torch.float32)[:, 0:2] = torch.rand((n, 2), generator=g, device=device) * 8 - 4
    gyro = torch.rand((n, 4), generator=g, device=device, dtype=torch.float32) * 0.02
    for _ in range(args.warmup):
        aof.derotate_batch(flows, gyro, 216.6677, 216.2457, 4.5, 0.01)
    torch.cuda.synchronize(device)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for a, b in ev:
        a.record()
        out = aof.derotate_batch(flows, gyro, 216.6677, 216.2457, 4.5, 0.01)
        b.record()
    torch.cuda.synchronize(device)
    elapsed = time.perf_counter() - t0
    k_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    alg = 16 + 16 + 8   # flow record + gyro sample in, two floats out
    achieved = alg * n / (k_ms * 1e-3) / 1e9
    traffic, src = pmc_traffic(f"derotate:{n}")
    line = {"metric": "flow records/s (gyro de-rotation, published PX4Flow compensation)",
            "value": round(n * args.steps / elapsed, 1), "unit": "records/s", "n_gpus": 1, "steps": args.steps,
            "warmup": args.warmup, "settle_steps": 0, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "derotate: 16-byte flow records + 16-byte gyro samples -> 2 floats", "records_per_launch": n},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": src,
                         "kernel": "k_derotate", "kernel_ms": round(k_ms, 5), "algorithmic_bytes_per_record": alg,
                         "records_per_launch": n}}
    from oracle import pyoracle as orc
    m = min(n, 2048)
    hf = flows[:m].cpu().numpy().view(np.float32).reshape(m, 4)
    hg = gyro[:m].cpu().numpy()
    ref = np.array([orc.derotate(float(hf[i, 0]), float(hf[i, 1]), float(hg[i, 0]), float(hg[i, 1]), float(hg[i, 3]),
                                 216.6677, 216.2457, 4.5, 0.01) for i in range(m)], dtype=np.float32)
    line["parity"] = {"oracle_records_bit_exact": bool(np.array_equal(out[:m].cpu().numpy().view(np.uint32),
                                                                      ref.view(np.uint32))),
                      "records_checked": m}
    print(json.dumps(line), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--settle-steps", type=int, default=-1,
                    help="untimed steps before the warm-up steps (clock settling under sustained load); 0 = off; -1 (default) = "
                         "as many as make about 0.2 s of load by the step's nominal work (1 024 VGA pairs: ~1 000 steps, 128 "
                         "pairs: ~8 000: a 25 us step needs thousands of steps before the clocks have settled -- 200-step runs of "
                         "the 128-pair step read 23 us, 2 000-step runs 19.3 us, profiles/r05_one_kernel_per_translation_unit_ab.txt); "
                         "the number actually run is in the line (settle_steps)")
    ap.add_argument("--pairs", type=int, default=1024, help="frame pairs per GPU per step")
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS) + ["ingest", "c1", "derotate", "seq"])
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU baseline budget; 0 = skip")
    ap.add_argument("--search", default="auto", choices=["auto", "exhaustive", "pruned", "adaptive"],
                    help="auto (default): what a fresh context runs = exact-adaptive for every tile size (exact pruning where "
                         "it pays: 16x16 tiles by a probe of the pair, 8x8 tiles by what the context's previous launches "
                         "reported; the exhaustive scan otherwise) -- an input-dependent rate; exhaustive (every candidate "
                         "summed completely: the data-independent rate) / pruned (exact partial-distortion elimination "
                         "always) / adaptive force a mode.  All modes write identical records; the line carries the other "
                         "modes (exhaustive_search = the data-independent rate of the same batch) beside the headline")
    ap.add_argument("--input", default="baseline", choices=["baseline", "realistic"],
                    help="baseline (default): SURVEY 8(d)'s synthetic pairs -- pure integer translations of a blurred-noise "
                         "texture, the best case of an exact pruned search (SAD 0 at the true shift); realistic: the same texture "
                         "as a camera would deliver it -- +-4 LSB sensor noise, a half-pixel displacement on top of the integer "
                         "shift, and the newer frame at half the contrast (exposure drift) -- the default line times this "
                         "input too (realistic_input)")
    ap.add_argument("--legs", default="all", choices=["all", "none"],
                    help="all (default): behind the timed region the line also times the other search modes on the same batch, "
                         "the realistic input and (workload c2, one GPU) the other BASELINE configurations in child runs; "
                         "none: the headline only")
    ap.add_argument("--max-shift", type=int, default=None,
                    help="largest synthetic shift per axis (default: the workload's search reach)")
    ap.add_argument("--force-generic", action="store_true", help="time the generic wave-per-block kernel")
    ap.add_argument("--noise", type=int, default=0, help="+-LSB uniform noise added to the current frames")
    ap.add_argument("--brightness", type=int, default=None,
                    help="exposure step added to the current frames (default: 9 grey levels for workloads with "
                         "mean equalisation, so that it has work to do; 0 = SURVEY 8(d)'s pure translations)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: --pairs per GPU (default); strong: --pairs in total, sharded over the "
                         "GPUs (BASELINE configs[3]: 1024 pairs over 8 GPUs)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (one rank per GPU); gloo = rehearsal of the N>1 "
                         "control flow with several ranks sharing the GPUs that exist")
    ap.add_argument("--coarse", default="auto", choices=["auto", "split"],
                    help="two-level workloads: how the coarse passes run -- auto (the fused kernel k_coarse where the "
                         "geometry allows) or split (K1 / level-1 search / K3 as separate kernels)")
    ap.add_argument("--reduce", default="auto", choices=["auto", "separate", "fused"],
                    help="separate: K3 behind the search; fused: the 8x8 search kernel reduces in its own launch "
                         "(votes through agent-scope atomics, finaliser waves behind the search); auto (default): "
                         "fused when several batches are in flight and the launch has at most 512 pairs -- the only "
                         "place where it measured faster (one kernel per batch leaves no gap for the other lane)")
    ap.add_argument("--graph", nargs="?", const="on", default="auto", choices=["auto", "on", "off"],
                    help="replay one step's launch sequence as a hipGraph: auto (default) = for launch-bound steps "
                         "(fewer than 8 G abs-diffs per step), on, off")
    ap.add_argument("--streams", type=int, default=0,
                    help="independent batches in flight: step i runs on HIP stream i %% S with its own context, "
                         "record buffers and workspace.  0 (default) = automatic: 2 when a step is launch-bound "
                         "(fewer than 2 Mi blocks: configs[3]'s 128 pairs per GPU), else 1")
    ap.add_argument("--gather-every", type=int, default=0,
                    help="N > 1: steps of a lane whose flow records one all_gather ships (0 = automatic: 4, and 16 for "
                         "launch-bound steps)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the process group and run the per-step gather even with ONE rank: rehearses the "
                         "N > 1 code path (RCCL, graph capture beside its watchdog thread) on a one-GPU box")
    ap.add_argument("--traffic", default="auto", choices=["auto", "live", "file"],
                    help="roofline.traffic (HBM bytes per launch from the PMC counters): auto / live = measured in this run by two short "
                         "child runs under rocprofv3 --pmc (one GPU only; falls back to the committed summary when the profiler is not "
                         "usable or a profiler is already around this process), file = the committed summary of the same command")
    ap.add_argument("--configs3-pairs", type=int, default=1024,
                    help="N > 1: pairs of the strong-scaling shape every multi-GPU line also times (BASELINE configs[3]: "
                         "1024 pairs sharded over the GPUs, against the same pairs on one GPU); 0 = skip")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="rehearse the N>1 control flow without a GPU: rendezvous, shard, gather a batch of "
                         "placeholder flow records over gloo, print one line and leave (CPU test of the launch path)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1 and args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Plain `python bench.py --gpus N`: start the N ranks ourselves (one process per GPU,
        # torch.distributed.run on the loopback address) BEFORE anything here touches the GPU,
        # and leave with the launcher's exit code.
        return spawn_ranks(args.gpus)
    if world != args.gpus:
        args.gpus = world
    if args.rendezvous_only:
        return rendezvous_only(args, rank, world)
    if args.backend == "nccl" and torch.cuda.device_count() < world:
        # one rank per GPU: fail before the rendezvous, with the reason, instead of inside RCCL
        sys.exit(f"bench.py: {world} ranks over RCCL need {world} GPUs, this node shows {torch.cuda.device_count()} "
                 f"(use --backend gloo to rehearse several ranks on the GPUs that exist)")
    dev_index = local_rank if args.backend == "nccl" else local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if "MASTER_PORT" not in os.environ:
                import socket
                with socket.socket() as sk:
                    sk.bind(("127.0.0.1", 0))
                    os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group("gloo")

    aof = ge.load_package()
    import importlib
    batch = importlib.import_module(ge.PKG_NAME + ".batch")
    if args.workload == "ingest":
        return bench_ingest(args, aof, device, rank, world, dist)
    if args.workload == "c1":
        return bench_c1(args, aof, rank, world, dist)
    if args.workload == "derotate":
        return bench_derotate(args, aof, device, rank, world, dist)
    if args.workload == "seq":
        return bench_sequence(args, aof, device, rank, world, dist)
    desc, W, H, over, reach = WORKLOADS[args.workload]
    if args.max_shift is not None:
        reach = args.max_shift
    over = dict(over)
    p = aof.px4flow_params(W, H, **over) if over.pop("_px4flow", 0) else aof.default_params(W, H, **over)

    if args.scaling == "strong":
        sb, se = batch.shard_range(args.pairs, rank, world)
        n = se - sb
        if (args.pairs % world) != 0:
            sys.exit("--scaling strong needs --pairs divisible by the number of GPUs")
    else:
        n = args.pairs

    def auto_config(n_pairs, streams=0, reduce="auto", graph="auto"):
        """Lanes, reduction and graph replay for a step of n_pairs pairs per GPU (the automatic choice)."""
        launch_bound = aof.abs_diffs(p) * n_pairs < 8e9   # a few dozen microseconds of search per step (C2: up to 330 pairs)
        if streams <= 0:
            streams = 2 if launch_bound else 1
        if reduce == "auto":
            reduce = "fused" if streams > 1 and n_pairs <= 512 else "separate"
        use_graph = graph == "on" or (graph == "auto" and launch_bound)
        return dict(streams=streams, reduce_mode=reduce, use_graph=use_graph, launch_bound=launch_bound)

    def settle_steps_for(n_pairs):
        """About 0.2 s of load, from the nominal work of a step alone (the same number on every rank)."""
        if args.settle_steps >= 0:
            return args.settle_steps
        est_us = max(20.0, aof.abs_diffs(p) * n_pairs / 120e6)   # ~120 T abs-diffs/s: 1 024 VGA pairs ~ 200 us
        return int(min(20000, max(200, 200000.0 / est_us)))

    cfg = auto_config(n, args.streams, args.reduce, args.graph)
    args.streams, reduce_mode, use_graph, launch_bound = cfg["streams"], cfg["reduce_mode"], cfg["use_graph"], cfg["launch_bound"]
    brightness = args.brightness if args.brightness is not None else (9 if p.mean_subtract else 0)
    REALISTIC = dict(noise=4, contrast=0.5, half=True)
    prev, cur, shifts = make_batch_gpu(W, H, n, reach, 0xA0F + 7919 * rank, device, brightness=brightness,
                                       **(REALISTIC if args.input == "realistic" else {}))
    if args.noise:
        g = torch.Generator(device=device)
        g.manual_seed(99 + rank)
        nz = torch.randint(-args.noise, args.noise + 1, cur.shape, generator=g, device=device, dtype=torch.int16)
        cur = (cur.to(torch.int16) + nz).clamp_(0, 255).to(torch.uint8)
        del nz

    class Lane:
        pass

    class Runner:
        """One configuration of the step on this rank.  A LANE = one context with its own stream, record
        buffers and workspace; the frames are shared (read-only).  streams == 1: one lane on torch's
        current stream.  streams == S: step i runs on lane i % S, so that the launch gaps and the
        low-occupancy reduction of one batch are covered by the search of the next one (independent
        batches in flight on separate HIP streams).  with_dist: the flow records of G consecutive steps
        of a lane are shipped by ONE all_gather."""

        def __init__(self, prev, cur, streams, reduce_mode, use_graph, launch_bound, with_dist, search=None):
            self.search = search or args.search
            self.n = n_ = prev.shape[0]
            self.dist = dist if with_dist else None
            self.reduce_mode, self.use_graph, self.launch_bound = reduce_mode, use_graph, launch_bound
            grid = aof.grid(p, 0)
            nb = grid[4] * grid[5]
            L = aof.workspace_layout(p, n_)
            self.G = G = 1 if self.dist is None else (args.gather_every if args.gather_every > 0 else (16 if launch_bound else 4))
            self.lanes = []
            for li in range(max(1, streams)):
                ln = Lane()
                ln.eng = aof.FlowEngine(p, dev_index)
                self.configure(ln.eng)
                ln.stream = torch.cuda.current_stream(device) if streams <= 1 else torch.cuda.Stream(device)
                ln.blocks = torch.empty((n_, nb), dtype=torch.int32, device=device)
                ln.sub = torch.empty((n_, nb), dtype=torch.uint8, device=device) if p.subpixel else None
                ln.ws = torch.empty(L.total_bytes, dtype=torch.uint8, device=device)
                # The flow records of G consecutive steps of a lane land in one ring segment, and ONE all_gather
                # per G steps ships it (RCCL and torch's collective call cost 25-40 us of host time and a
                # cross-queue dependency per call: per step that was +14 us on a 1 024-pair step and made a 128-pair
                # step host-bound; every step's records still cross xGMI inside the timed region).  Two segments per
                # lane take turns; the compute stream never WAITS for a gather -- before a segment is written again
                # the HOST checks that the gather that read it has completed.
                ln.rings = [torch.empty((G, n_, 16), dtype=torch.uint8, device=device) for _ in range(2)]
                ln.flows = [ln.rings[r][g] for r in range(2) for g in range(G)]
                ln.reads = [None, None]    # the gather in flight that reads ring segment r
                ln.gathered = ([torch.empty((world * G * n_, 16), dtype=torch.uint8, device=device) for _ in range(2)]
                               if self.dist is not None else None)
                ln.enqueue = [ln.eng.bind_batch(prev, cur, ln.blocks, f, ln.ws, subdirs=ln.sub,
                                                stream=ln.stream.cuda_stream if streams > 1 else None)
                              for f in ln.flows]   # one ctypes call per step
                ln.graphs = None
                ln.i = 0
                self.lanes.append(ln)
            self.eng = self.lanes[0].eng
            self.state = {"i": 0, "pending": None, "gathered": None, "last": (self.lanes[0], 0)}
            self.multi = len(self.lanes) > 1
            self.profiling = False
            if use_graph:   # a short step is launch-bound: replay it as one hipGraph
                for ln in self.lanes:
                    ln.graphs = []
                    for e in ln.enqueue:
                        e()
                        torch.cuda.synchronize(device)
                        g = torch.cuda.CUDAGraph()
                        if self.multi:
                            with torch.cuda.graph(g, stream=ln.stream):
                                e()
                        else:   # (captured on torch's side stream, replayed on the current one)
                            with torch.cuda.graph(g):
                                e()
                        ln.graphs.append(g)

        def configure(self, e):
            if self.search != "auto":
                e.set_search_mode({"exhaustive": aof.SEARCH_EXHAUSTIVE, "pruned": aof.SEARCH_PRUNED,
                                   "adaptive": aof.SEARCH_ADAPTIVE}[self.search])
            if args.force_generic:
                e.force_generic(True)
            e.set_reduce_fusion(self.reduce_mode == "fused")
            if args.coarse != "auto":
                e.set_split_coarse(True)

        def set_profiling(self, on, kernels=None):
            self.profiling = bool(on)
            self.eng.set_profiling(on, kernels=kernels) if kernels is not None else self.eng.set_profiling(on)

        def step(self):
            state, lanes, G, n_ = self.state, self.lanes, self.G, self.n
            ln = lanes[state["i"] % len(lanes)]
            k = ln.i % len(ln.flows)
            ln.i += 1
            state["i"] += 1
            state["last"] = (ln, k)
            r, g = divmod(k, G)
            if self.dist is not None and g == 0 and ln.reads[r] is not None:   # (2 G steps of this lane ago: done long since)
                ln.reads[r].wait_host()
                ln.reads[r] = None
            if ln.graphs is not None and not self.profiling:
                if self.multi:
                    torch.cuda.set_stream(ln.stream)
                ln.graphs[k].replay()
            else:
                ln.enqueue[k]()
            if self.dist is None or g != G - 1:
                return
            if self.multi:
                torch.cuda.set_stream(ln.stream)
            seg = ln.rings[r].view(G * n_, 16)
            src = seg.cpu() if args.backend == "gloo" else seg  # gloo (rehearsal) gathers host copies
            state["pending"] = batch.gather_flows_async(src, world * G * n_, force=args.force_dist,
                                                        out=ln.gathered[r] if args.backend == "nccl" else None)
            ln.reads[r] = state["pending"]

        def drain(self):
            for ln in self.lanes:
                for k, g in enumerate(ln.reads):
                    if g is not None:
                        g.wait_host()
                        ln.reads[k] = None
            if self.state["pending"] is not None:
                self.state["gathered"] = self.state["pending"].wait()
                self.state["pending"] = None

        def fence(self):
            self.drain()
            torch.cuda.synchronize(device)
            if self.dist is not None:
                self.dist.barrier()
                torch.cuda.synchronize(device)

        def settle(self, steps):
            # Untimed settling phase: the device's clocks take tens of milliseconds of sustained load to
            # settle (measured: K2 0.251 ms in a cold 20-step run, 0.227 ms after 150 ms of load).  A fixed
            # number of steps, so that every rank issues the same collectives.
            for i in range(steps):
                self.step()
                if i % 50 == 49:
                    self.drain()
                    torch.cuda.synchronize(device)
            self.fence()

        def timed(self, steps):
            """EXACTLY `steps` steps between barrier + synchronize on both sides; MAX over ranks."""
            self.fence()
            t0 = time.perf_counter()
            for _ in range(steps):
                self.step()
            self.fence()
            elapsed = time.perf_counter() - t0
            if self.dist is not None:
                t = torch.tensor([elapsed], dtype=torch.float64, device=device if args.backend == "nccl" else "cpu")
                self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
                elapsed = float(t.item())
            return elapsed

        def close(self):
            self.fence()
            if self.multi:
                torch.cuda.set_stream(torch.cuda.default_stream(device))
            for ln in self.lanes:
                ln.graphs = None
                ln.eng.close()

    run = Runner(prev, cur, args.streams, reduce_mode, use_graph, launch_bound, with_dist=dist is not None)
    eng, lanes, state, multi, G = run.eng, run.lanes, run.state, run.multi, run.G
    nb = eng.nblocks(0)
    blocks = lanes[0].blocks
    step, fence, drain = run.step, run.fence, run.drain
    eng_set_profiling = run.set_profiling

    settle_n = settle_steps_for(n)
    run.settle(settle_n)   # (W is often only a handful of steps)
    # HIP events around the dominant kernel (K2, level 0) on the launch stream.  Steps of a millisecond
    # or so carry them inside the timed region; every event pair costs the stream a few microseconds of
    # serialisation, so short steps (and replayed graphs, several lanes) are timed
    # without events and K2 is measured in a second pass of the same K steps on lane 0 right after it.
    eng_set_profiling(True, kernels=[aof.K_SEARCH])
    lanes[0].enqueue[0]()
    fence()
    lps = max(1, len(eng.profile_ms(aof.K_SEARCH)))   # K2 launches per step (a batch of more than 2^31 blocks is cut into several)
    events_in_timed_region = lps == 1 and not use_graph and not multi and not launch_bound
    eng_set_profiling(events_in_timed_region, kernels=[aof.K_SEARCH])
    for _ in range(args.warmup):
        step()
    elapsed = run.timed(args.steps)
    # Per-step times for the median: a further pass of the same K steps with one event behind every step
    # (outside the timed region, so that `value` is not taxed with the event records).
    marks = []
    for _ in range(args.steps):
        step()
        ln, _k = state["last"]
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(ln.stream)
        marks.append((ln, ev))
    fence()
    per_step_ms = []
    stride = len(lanes)
    for j in range(stride, len(marks)):   # lane-local interval = `stride` steps of the job
        per_step_ms.append(marks[j - stride][1].elapsed_time(marks[j][1]) / stride)
    if multi:
        torch.cuda.set_stream(torch.cuda.default_stream(device))
    if not events_in_timed_region:
        eng_set_profiling(True, kernels=[aof.K_SEARCH])
        for _ in range(args.steps):
            lanes[0].enqueue[0]()
        fence()
    eng_set_profiling(False)

    # ---- dominant kernel against its roofline (rank 0's launches) ----
    # (the event ring keeps the last 256 launches: whole steps only)
    k2 = eng.profile_ms(aof.K_SEARCH)
    k2 = k2[len(k2) - min(args.steps, len(k2) // lps) * lps:]
    k2_ms = float(np.sum(k2)) / (len(k2) // lps) if k2 else float("nan")
    # Every kernel of the step bracketed by events, in ONE further pass of its own (outside the timed region: an event
    # pair costs the stream a few microseconds), with that pass's own step time beside the parts -- so that the parts are
    # held against the whole they were measured in.  `ms_per_step` (wall clock over the K timed steps) stays the
    # authoritative figure `value` is computed from; roofline.kernel_ms is K2 from the timed region where it carries events.
    kpass = max(1, min(args.steps, 64, 256 // lps))   # (the event ring keeps 256 launches per kernel)
    eng_set_profiling(True)
    fence()
    t1 = time.perf_counter()
    for _ in range(kpass):
        lanes[0].enqueue[0]()
    fence()
    kpass_ms = (time.perf_counter() - t1) / kpass * 1e3
    eng_set_profiling(False)
    per_kernel = {}
    for name, kid in (("pyramid", aof.K_PYRAMID), ("search_l1", aof.K_SEARCH_L1),
                      ("reduce_l1", aof.K_REDUCE_L1), ("search", aof.K_SEARCH), ("reduce", aof.K_REDUCE)):
        v = eng.profile_ms(kid)
        if v:   # per step: the sum over the step's sub-batch launches
            per_kernel[name] = round(float(np.sum(v)) / kpass, 5)
    if "pyramid" in per_kernel and "search_l1" not in per_kernel and p.pyramid_levels == 2:
        # the AOF_K_PYRAMID bracket timed k_coarse: sums + pyramid + level-1 search + level-1 reduction
        per_kernel["coarse_fused"] = per_kernel.pop("pyramid")
    per_kernel["step_in_this_pass"] = round(kpass_ms, 5)
    per_kernel["note"] = (f"one separate pass of {kpass} steps on lane 0 with HIP events around every kernel; step_in_this_pass is "
                          "that pass's own wall time per step: the parts add up to less than it; ms_per_step is authoritative")
    # ---- BASELINE configs[3] on every N > 1 line, whatever --scaling says: the STRONG shape ----
    # (--configs3-pairs, default 1 024, sharded over the ranks: 128 per GPU at N = 8, run as the automatic
    # choice runs it -- two lanes, graph replay, reduction in the search launch -- with the gather inside the
    # timed region; beside it the one-GPU step over ALL of those pairs, timed on rank 0 of the same job, so
    # that the line itself answers "how many times faster than one GPU is the sharded batch")
    configs3 = None
    ranks_seen, devices_seen = 1, None
    if dist is not None:
        ranks_seen = dist.get_world_size()
        props = torch.cuda.get_device_properties(device)
        mine = f"rank {rank}: cuda:{dev_index} {props.name} ({getattr(props, 'gcnArchName', '?')}, {props.total_memory >> 30} GiB)"
        devices_seen = [None] * ranks_seen
        dist.all_gather_object(devices_seen, mine)
    if dist is not None and args.configs3_pairs > 0:
        total3 = args.configs3_pairs
        if total3 % world:
            configs3 = {"skipped": f"{total3} pairs do not divide over {world} ranks"}
        else:
            n3 = total3 // world
            c3 = auto_config(n3)
            steps3 = max(args.steps, 1000)   # (a 20 us step: the timed region is a side leg of the line, long enough to be read)
            if args.scaling == "strong" and args.pairs == total3 and cfg == auto_config(n):
                ms3 = elapsed / args.steps * 1e3   # the main region IS this shape
            else:
                if n3 <= n:
                    prev3, cur3 = prev[:n3], cur[:n3]
                else:
                    prev3, cur3, _ = make_batch_gpu(W, H, n3, reach, 0xC3 + 7919 * rank, device, brightness=brightness)
                r3 = Runner(prev3, cur3, c3["streams"], c3["reduce_mode"], c3["use_graph"], c3["launch_bound"], with_dist=True)
                r3.settle(settle_steps_for(n3))
                for _ in range(args.warmup):
                    r3.step()
                ms3 = r3.timed(steps3) / steps3 * 1e3
                r3.close()
                del r3, prev3, cur3
            # the one-GPU reference: all total3 pairs on rank 0's GPU, the configuration the N = 1 line runs
            ms1 = ms1x = None
            if rank == 0:
                if total3 <= n:
                    prev1, cur1 = prev[:total3], cur[:total3]
                else:
                    prev1, cur1, _ = make_batch_gpu(W, H, total3, reach, 0xA0F, device, brightness=brightness)
                c1 = auto_config(total3)
                r1 = Runner(prev1, cur1, c1["streams"], c1["reduce_mode"], c1["use_graph"], c1["launch_bound"], with_dist=False)
                r1.settle(settle_steps_for(total3))
                for _ in range(args.warmup):
                    r1.step()
                ms1 = r1.timed(steps3) / steps3 * 1e3
                r1.close()
                del r1
                # ... and with the search the shares run: launches of 128 pairs are too small for the pruned kernel (its
                # waves carry their hints from chunk to chunk), so the ranks search exhaustively while one GPU with all
                # the pairs prunes -- the ratio against THIS leg is the scaling of one and the same computation
                if args.search == "auto" and eng.variant == "lane8":
                    r1 = Runner(prev1, cur1, c1["streams"], c1["reduce_mode"], c1["use_graph"], c1["launch_bound"], with_dist=False,
                                search="exhaustive")
                    r1.settle(settle_steps_for(total3))
                    for _ in range(args.warmup):
                        r1.step()
                    ms1x = r1.timed(steps3) / steps3 * 1e3
                    r1.close()
                    del r1
                del prev1, cur1
            dist.barrier()
            configs3 = {"global_pairs": total3, "pairs_per_gpu": n3, "steps": steps3, "ms_per_step": round(ms3, 5),
                        "value": round(total3 / (ms3 * 1e-3), 1), "unit": "frame-pairs/s", "scaling": "strong",
                        "streams": c3["streams"], "reduce": c3["reduce_mode"], "graph_replay": bool(c3["use_graph"]),
                        "one_gpu_ms_per_step": round(ms1, 5) if ms1 else None,
                        "one_gpu_value": round(total3 / (ms1 * 1e-3), 1) if ms1 else None,
                        "vs_one_gpu_1024": round(ms1 / ms3, 3) if ms1 else None,
                        "one_gpu_exhaustive_ms_per_step": round(ms1x, 5) if ms1x else None,
                        "vs_one_gpu_1024_exhaustive": round(ms1x / ms3, 3) if ms1x else None,
                        "note": f"{total3} pairs sharded over {world} ranks with the flow records gathered on every rank, "
                                f"against the same {total3} pairs on rank 0's GPU alone (same job, same clock)"}
    alg_bytes = aof.algorithmic_bytes(p)
    achieved = alg_bytes * n / (k2_ms * 1e-3) / 1e9
    # `traffic` is NOT measured in this run: PMC counters need rocprofv3 around the process.  It is
    # read from the committed summary of the same command under profiles/ (tools/collect_evidence.sh);
    # `traffic_source` names the file, or says that no summary exists for this workload and size.
    traffic, traffic_source = pmc_traffic(f"{args.workload}:{n}")
    traffic_step = pmc_traffic(f"{args.workload}:{n}", "step_bytes")[0]
    if args.traffic != "file" and rank == 0 and world == 1 and dist is None:
        # (behind the timed region) measure it here and now; the committed summary stays the fallback
        flag = ["--search", args.search]
        if args.search == "auto":
            # the default line also times the other modes behind its timed region: the probe run names the headline's --
            # for an 8x8 context the kernel its ADAPTIVE mode settled on (a fresh context would start with the pruned one)
            st = eng.search_stats() if eng.variant == "lane8" else None
            flag = ["--search", "exhaustive" if st and st["exhaustive_launches"] > st["pruned_launches"] else "adaptive"]
        live = live_traffic(args, n, flag)
        if live:
            traffic, traffic_step, traffic_source = live
        else:
            traffic_source = (f"live measurement failed ({getattr(live_traffic, 'error', 'rocprofv3 not usable here, or a profiler is already around this process')}); "
                              "committed summary of the same command: " + str(traffic_source))
    step_ms = elapsed / args.steps * 1e3
    achieved_step = alg_bytes * n / (step_ms * 1e-3) / 1e9   # the whole step (every kernel + gaps), per GPU

    out = {
        "metric": f"frame-pairs/s ({W}x{H}, {p.tile}x{p.tile} SAD, +-{p.search} search)",
        "value": round(world * n * args.steps / elapsed, 1),
        "unit": "frame-pairs/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "settle_steps": settle_n,
        "ms_per_step": round(step_ms, 4),
        "timed_region_ms": round(elapsed * 1e3, 3),
        "timing_note": "ms_per_step = wall clock of the K timed steps between barrier + synchronize, / K: AUTHORITATIVE, value = pairs / it.  "
                       "ms_per_step_median / p10_p90 = per-step HIP-event intervals of a FURTHER pass of K steps (each event costs the stream "
                       "a few microseconds: a little slower by construction); kernels_ms = a third pass with events around every kernel",
        # median over per-step HIP-event intervals of a further pass of the same K steps
        "ms_per_step_median": round(float(np.median(per_step_ms)), 5) if per_step_ms else None,
        "value_median": round(world * n / (float(np.median(per_step_ms)) * 1e-3), 1) if per_step_ms else None,
        "ms_per_step_p10_p90": [round(float(np.percentile(per_step_ms, 10)), 5),
                                round(float(np.percentile(per_step_ms, 90)), 5)] if per_step_ms else None,
        "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": "u8", "data": "synthetic",
        "config": {"workload": desc, "pairs_per_gpu": n, "global_pairs": world * n,
                   "search_kernel": eng.variant, "search": args.search, "coarse": args.coarse, "reduce": reduce_mode,
                   "streams": len(lanes), "graph_replay": bool(use_graph), "k2_launches_per_step": lps, "input": args.input, "noise_lsb": args.noise, "exposure_step": brightness, "parallelism": f"pairs sharded x{world}, flows all_gather ({args.backend}) every {G} steps per lane"
                   if dist is not None else "single GPU"},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                     "traffic_source": traffic_source,
                     # beyond-L2 bytes of ALL kernels of one step (same source): what to hold against the
                     # algorithmic bytes of a multi-kernel workload
                     "traffic_step": traffic_step,
                     # the same algorithmic bytes over the WHOLE step (all kernels of the workload and
                     # the gaps between them): the figure to quote for multi-kernel workloads (c3, c5p)
                     "achieved_step": round(achieved_step, 1), "frac_step": round(achieved_step / HBM_PEAK_GBS, 4),
                     "kernel": "k_search (K2)", "kernel_ms": round(k2_ms, 5),
                     "kernel_ms_from": "HIP events on the launch stream inside the timed region" if events_in_timed_region
                     else f"HIP events in a second pass of the same {args.steps} steps (sum of the {lps} "
                          "launches of a step, which other kernels overlap): quote frac_step for this workload",
                     "algorithmic_bytes_per_pair": alg_bytes, "pairs_per_launch": n,
                     # nominal abs-diffs of the exhaustive scan; meaningless when candidates are pruned
                     "abs_diff_per_s": round(aof.abs_diffs(p) * n / (k2_ms * 1e-3), 1)
                     if eng.search_mode == aof.SEARCH_EXHAUSTIVE else None},
        "kernels_ms": per_kernel,
    }
    if dist is not None:
        out["ranks_seen"] = ranks_seen
        out["devices"] = devices_seen
        out["configs3"] = configs3

    # ---- secondary, clearly separate: the OTHER search modes on the same batch ----
    # (same records bit for bit.  The exact-adaptive mode is the headline -- what a fresh context runs -- with the
    # exhaustive scan, a data-independent rate, and the always-pruned mode beside it.)
    # (the grouped small-grid lane8 kernel never prunes; the 16x16 kernel prunes per (dy row, block) item)
    mode_names = {aof.SEARCH_EXHAUSTIVE: "exhaustive", aof.SEARCH_PRUNED: "exact-pruned", aof.SEARCH_ADAPTIVE: "exact-adaptive"}
    head_mode = eng.search_mode
    out["config"]["search"] = mode_names[head_mode]
    if head_mode != aof.SEARCH_EXHAUSTIVE:
        out["config"]["search_note"] = ("exact partial-distortion elimination where it pays: every candidate that could win or tie is summed completely, "
                                        "the records are the exhaustive scan's bit for bit (compared on the device behind the timed region: "
                                        "exhaustive_search.records_identical_to_headline); exhaustive_search.per_gpu_value is the data-independent rate "
                                        "of the same batch")
    if eng.variant == "lane8" and head_mode == aof.SEARCH_ADAPTIVE:
        # which kernel the context's launches ran (ADAPTIVE 8x8: decided per launch from the pruned kernel's own reports)
        out["config"]["adaptive_search"] = eng.search_stats()
    if eng.variant == "tile16_lds" and head_mode == aof.SEARCH_ADAPTIVE:
        # ADAPTIVE 16x16: the probe's verdict per pair of the last launch (aof_ws_layout.hints): 0 = exhaustive scan, 1 / 2 / 3 / 4 =
        # pruned on two- / one- / four- / eight-row lower bounds
        try:
            L_ = aof.workspace_layout(p, n)
            hw = run.lanes[0].ws[L_.hints:L_.hints + 4 * n].cpu().numpy().view("uint32")
            hv = hw & 0xFF
            out["config"]["adaptive_search"] = {"verdicts": {str(k): int((hv == k).sum()) for k in range(5) if (hv == k).any()},
                                                "legend": "0 exhaustive, 1 two-row, 2 one-row, 3 four-row, 4 eight-row bounds",
                                                "separation_permille_median": int(sorted((hw >> 8).tolist())[len(hw) // 2])}
        except Exception as e:
            out["config"]["adaptive_search"] = {"error": f"{type(e).__name__}: {e}"[:200]}
    pruned_available = (eng.variant == "lane8" and eng.nblocks(0) > 256) or eng.variant == "tile16_lds"
    if args.search == "auto" and pruned_available and not args.force_generic and args.legs == "all":
        others = [m for m in (aof.SEARCH_EXHAUSTIVE, aof.SEARCH_PRUNED, aof.SEARCH_ADAPTIVE) if m != head_mode]
        ref_blocks = blocks.clone()
        for mode in others:
            eng.set_search_mode(mode)
            # settle like the headline: at least 0.25 s of this mode's own load before its timed steps (clocks, and
            # for the adaptive modes what the context learns from its launches)
            t1 = time.perf_counter()
            while time.perf_counter() - t1 < 0.25:
                for _ in range(50):
                    lanes[0].enqueue[0]()
                torch.cuda.synchronize(device)
            eng.set_profiling(True, kernels=[aof.K_SEARCH])
            for _ in range(2):
                lanes[0].enqueue[0]()
            torch.cuda.synchronize(device)
            t1 = time.perf_counter()
            for _ in range(args.steps):
                lanes[0].enqueue[0]()
            torch.cuda.synchronize(device)
            dt = time.perf_counter() - t1
            pk2 = eng.profile_ms(aof.K_SEARCH)
            pk2 = pk2[len(pk2) - min(args.steps, len(pk2) // lps) * lps:]
            pk2 = float(np.sum(pk2)) / (len(pk2) // lps)
            eng.set_profiling(False)
            key = {aof.SEARCH_EXHAUSTIVE: "exhaustive_search", aof.SEARCH_PRUNED: "exact_pruned_search",
                   aof.SEARCH_ADAPTIVE: "exact_adaptive_search"}[mode]
            out[key] = {
                "per_gpu_value": round(n * args.steps / dt, 1), "unit": "frame-pairs/s", "kernel_ms": round(pk2, 5),
                "roofline_frac": round(alg_bytes * n / (pk2 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "records_identical_to_headline": bool(torch.equal(ref_blocks, blocks)),
                "note": {aof.SEARCH_EXHAUSTIVE: "AOF_SEARCH_EXHAUSTIVE: every candidate summed completely, data-independent rate",
                         aof.SEARCH_PRUNED: "opt-in AOF_SEARCH_PRUNED (partial-distortion elimination always): bit-identical "
                                            "records, rate depends on the images; not the headline",
                         aof.SEARCH_ADAPTIVE: "AOF_SEARCH_ADAPTIVE"}[mode]}
        eng.set_search_mode(head_mode)
        state["last"] = (lanes[0], 0)  # its first flow buffer holds the latest records

    # ---- the same step on a REALISTIC input (one GPU): what a camera delivers instead of SURVEY 8(d)'s pure translations ----
    # (+-4 LSB noise, a half-pixel displacement on top of the integer shift, the newer frame at half the contrast:
    #  mainloop.cpp:197-275,295-322 feeds OV7251 frames under an auto-exposure loop.  Timed like the headline: same
    #  configuration, fresh contexts, settling steps, warm-up, K steps between synchronisations.)
    if args.legs == "all" and args.input == "baseline" and not args.noise and world == 1 and dist is None and rank == 0:
        prev_r, cur_r, _ = make_batch_gpu(W, H, n, reach, 0xA0F + 7919 * rank, device, brightness=brightness, **REALISTIC)
        leg = {"input": "+-4 LSB noise + a half-pixel displacement per pair + the newer frame at contrast 0.5 (bench.py --input realistic)"}
        for key, mode in (("default_search", None), ("exhaustive_search", "exhaustive")):
            if mode and not pruned_available:
                continue
            rr = Runner(prev_r, cur_r, args.streams, reduce_mode, use_graph, launch_bound, with_dist=False, search=mode)
            rr.settle(min(settle_n, 600))
            for _ in range(args.warmup):
                rr.step()
            ms_r = rr.timed(args.steps) / args.steps * 1e3
            ent = {"value": round(n / (ms_r * 1e-3), 1), "unit": "frame-pairs/s", "ms_per_step": round(ms_r, 4),
                   "frac_step": round(alg_bytes * n / (ms_r * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
            if mode is None:
                ent["search"] = mode_names[rr.eng.search_mode]
                if rr.eng.variant == "lane8" and rr.eng.search_mode == aof.SEARCH_ADAPTIVE:
                    ent["adaptive_search"] = rr.eng.search_stats()
                from oracle import pyoracle as orc
                po = orc.params_from(p)
                lnr, kr = rr.state["last"]
                gb, gf = aof.blocks_view(lnr.blocks[:2]), aof.flows_view(lnr.flows[kr][:2])
                hp, hc = prev_r[:2].cpu().numpy(), cur_r[:2].cpu().numpy()
                okr = True
                for i in range(2):
                    ref = orc.flow_pair(po, hp[i], hc[i])
                    okr &= gb[i].tobytes() == ref["blocks"].tobytes() and gf[i].tobytes() == ref["flow"].tobytes()
                ent["oracle_pairs_bit_exact"] = bool(okr)
                ent["pairs_checked"] = 2
                fr = aof.flows_view(lnr.flows[kr])
                ent["mean_quality"] = round(float(np.mean(fr["quality"])), 1)
            leg[key] = ent
            rr.close()
            del rr
        out["realistic_input"] = leg
        del prev_r, cur_r
    # The three rates side by side at the top of the line: `value` is what the library's default (exact-adaptive) search
    # does on BASELINE's synthetic input -- integer translations, the best case of exact pruning --; the exhaustive scan of
    # the same batch is the rate no input can lower; the realistic input is what a camera's frames would see.
    if head_mode != aof.SEARCH_EXHAUSTIVE and "exhaustive_search" in out:
        out["value_data_independent"] = out["exhaustive_search"]["per_gpu_value"] * world
    if "realistic_input" in out:
        out["value_realistic_input"] = out["realistic_input"]["default_search"]["value"]
    if head_mode != aof.SEARCH_EXHAUSTIVE:
        out["headline_note"] = ("value = the default exact-ADAPTIVE search on SURVEY 8(d)'s synthetic pairs (pure integer translations: SAD 0 at the "
                                "true shift, the best case of exact pruning) -- an INPUT-DEPENDENT rate; value_data_independent = "
                                "AOF_SEARCH_EXHAUSTIVE on the same batch (no input lowers it); value_realistic_input = the default search on "
                                "+-4 LSB noise + half-pixel motion + half contrast.  Records are identical in every mode.")

    # ---- the other BASELINE configurations, driver-visible: child runs of this script (workload c2, one GPU) ----
    if (args.legs == "all" and args.workload == "c2" and args.input == "baseline" and not args.noise and world == 1 and dist is None
            and rank == 0 and n == 1024 and not being_profiled()):
        import subprocess
        legs = {}
        for key, extra in (("c3", ["--workload", "c3"]),
                           ("c3_noise16", ["--workload", "c3", "--noise", "16"]),
                           ("c3_realistic", ["--workload", "c3", "--input", "realistic"]),
                           ("c5", ["--workload", "c5", "--pairs", "256"]),
                           ("c5_noise8", ["--workload", "c5", "--pairs", "256", "--noise", "8"]),
                           ("c5_realistic", ["--workload", "c5", "--pairs", "256", "--input", "realistic"])):
            cmd = [sys.executable, os.path.abspath(__file__)] + extra + [
                "--steps", str(args.steps), "--warmup", str(args.warmup), "--settle-steps", str(args.settle_steps if args.settle_steps >= 0 else 300),
                "--cpu-seconds", "0", "--traffic", "file", "--legs", "none"]
            try:
                r = subprocess.run(cmd, capture_output=True, text=True, timeout=150)
                d = json.loads(r.stdout.strip().splitlines()[-1])
                legs[key] = {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"],
                             "frac_step": d["roofline"]["frac_step"], "search": d["config"]["search"],
                             "pairs_per_launch": d["config"]["pairs_per_gpu"], "steps": d["steps"],
                             "oracle_pairs_bit_exact": d.get("parity", {}).get("oracle_pairs_bit_exact"),
                             "command": "bench.py " + " ".join(cmd[2:])}
                if "adaptive_search" in d["config"]:
                    legs[key]["adaptive_search"] = d["config"]["adaptive_search"]
            except Exception as e:   # a leg that fails says so; the headline stands on its own
                legs[key] = {"error": f"{type(e).__name__}: {e}"[:300]}
        out["workloads"] = legs

    # ---- parity on a sample + CPU baseline (rank 0, N=1 only) ----
    if rank == 0:
        from oracle import pyoracle as orc
        po = orc.params_from(p)
        ln, k = state["last"]
        flows = ln.flows[k]
        gb, gf = aof.blocks_view(ln.blocks[:4]), aof.flows_view(flows[:4])
        hp, hc = prev[:4].cpu().numpy(), cur[:4].cpu().numpy()
        ok = True
        for i in range(4):
            ref = orc.flow_pair(po, hp[i], hc[i])
            ok &= gb[i].tobytes() == ref["blocks"].tobytes() and gf[i].tobytes() == ref["flow"].tobytes()
        fl = aof.flows_view(flows)
        known = bool(np.array_equal(fl["flow_x"], shifts[:, 0].astype(np.float32)) and
                     np.array_equal(fl["flow_y"], shifts[:, 1].astype(np.float32))) if not args.noise and args.input == "baseline" else None
        out["parity"] = {"oracle_pairs_bit_exact": bool(ok), "pairs_checked": 4,
                         "all_pairs_return_known_shift": known,
                         "note": "oracle = this repo's CPU restatement (upstream PX4 source unavailable)"}
        if world == 1 and args.cpu_seconds > 0:
            cores = host_cores()
            m = min(n, max(4 * cores, 16))
            hp, hc = prev[:m].cpu().numpy(), cur[:m].cpu().numpy()
            done, used, spent = 0, cores, 0.0
            # the timing leg lets the CPU use its own SAD instruction (psadbw); the checker above
            # ran the plain byte loop, and tests/test_oracle.py pins the two to each other
            simd = orc.fast_sad_available()
            orc.set_fast_sad(simd)
            try:
                orc.flow_batch(po, hp[:cores], hc[:cores], threads=cores)  # thread pool warm-up
                while spent < args.cpu_seconds:  # bounded sample: repeat the slice until the budget is spent
                    t1 = time.perf_counter()
                    _, _, used = orc.flow_batch(po, hp, hc, threads=cores)
                    spent += time.perf_counter() - t1
                    done += m
            finally:
                orc.set_fast_sad(False)
            # the same oracle on ONE thread, a few seconds
            one_done, one_spent = 0, 0.0
            orc.set_fast_sad(simd)
            try:
                while one_spent < min(3.0, args.cpu_seconds):
                    t1 = time.perf_counter()
                    orc.flow_batch(po, hp[:4], hc[:4], threads=1)
                    one_spent += time.perf_counter() - t1
                    one_done += 4
            finally:
                orc.set_fast_sad(False)
            out["cpu_baseline"] = {"value": round(done / spent, 2), "unit": "frame-pairs/s",
                                   "cores": int(used), "kind": "port",
                                   "single_thread": {"value": round(one_done / one_spent, 2), "unit": "frame-pairs/s",
                                                     "cores": 1, "sample": f"{one_done} pairs, {one_spent:.1f} s"},
                                   "sample": f"{done} pairs ({m} distinct) of the same workload, this repo's "
                                             f"C oracle (-O2, SAD via {'SSE2 psadbw' if simd else 'the byte loop'}), "
                                             f"OpenMP over pairs, {spent:.1f} s"}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main() or 0)
