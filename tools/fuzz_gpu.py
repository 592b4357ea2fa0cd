#!/usr/bin/env python3
"""Long differential fuzz on a GPU box (not part of the test suite): random geometry, options
and image statistics; default, in-launch reduction, pruned, pruned with the reduction in its launch, generic and separate-kernel
device paths against the CPU oracle.
    python tools/fuzz_gpu.py [n_cases] [first_seed]            random configurations, two pairs per call
    python tools/fuzz_gpu.py [n_cases] [first_seed] many       the persistent coarse kernel: hundreds of pairs per call
    python tools/fuzz_gpu.py [n_cases] [first_seed] resident   the per-call path: resident kernel, tagged graph, eager launches
    python tools/fuzz_gpu.py [n_cases] [first_seed] sequence   aof_sequence_device: whole recordings against the oracle's calcFlow chain"""
import importlib
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402
from oracle import pyoracle as orc  # noqa: E402

aof = ge.load_package()
synth = importlib.import_module("aero_optical_flow_amd.synth")


def small_case(rng):
    """Frames that fit LDS with grids of at most 256 blocks: the one-workgroup kernel (k_flow_small)."""
    levels = int(rng.choice([1, 2, 2]))
    grid_mode = int(rng.choice([0, 1, 1]))
    min_dim = (8 + 2 * 5 + 8) * (2 if levels == 2 else 1)
    if grid_mode:
        w = 16 * int(rng.integers((min_dim + 15) // 16, 17))
        h = int(rng.integers(min_dim, 225))
    else:   # dense: keep the grid at or below 256 blocks (sometimes just above: the separate kernels)
        w = 16 * int(rng.integers((min_dim + 15) // 16, 11))
        h = int(rng.integers(min_dim, 150))
    if levels == 2:
        h += h & 1
    return dict(width=w, height=h, tile=8, search=4, pyramid_levels=levels, grid_mode=grid_mode,
                subpixel=int(rng.integers(0, 2)), mean_subtract=int(rng.integers(0, 2)), hist_filter=int(rng.integers(0, 2)),
                feature_threshold=int(rng.choice([0, 30, 30, 200, 2000])),
                value_threshold=int(rng.choice([0, 500, 3000, 3000, 70000])),
                min_valid=int(rng.choice([0, 10, 10, 500])), num_blocks=int(rng.integers(2, 9)))


def small_eligible(p, g0, g1):
    """Mirrors flow_small_supported (statistics only)."""
    if p.tile != 8 or p.search != 4 or p.width % 16 or not 8 <= g0[4] * g0[5] <= 256:
        return False
    lds = 2 * (p.width * p.height + 16)
    if p.pyramid_levels == 2:
        if p.height % 2 or not 8 <= g1[4] * g1[5] <= 256:
            return False
        lds += 2 * ((p.width // 2) * (p.height // 2) + 32)
    return lds + 4096 <= 160 * 1024


def case(rng):
    if rng.random() < 0.3:
        return small_case(rng)
    fast = rng.random() < 0.7          # geometries the LDS-tiled kernels serve
    fast16 = fast and rng.random() < 0.2
    tile = (16 if fast16 else 8) if fast or rng.random() < 0.5 else 16
    search = (8 if fast16 else 4) if fast else int(rng.choice([8 if tile == 16 else 4, rng.integers(1, 9)]))
    levels = int(rng.choice([1, 1, 2]))
    grid_mode = 0 if fast else int(rng.choice([0, 1]))
    subpixel = int(rng.random() < 0.35) if fast else (1 if grid_mode else int(rng.integers(0, 2)))
    min_dim = (tile + 2 * (search + 1) + 8) * (2 if levels == 2 else 1)
    w = int(rng.integers(min_dim, min_dim + 400))
    h = int(rng.integers(min_dim, min_dim + 200))
    if fast16 and rng.random() < 0.15:   # narrow and tall, or wide and flat: one or two tiles across, a hundred along
        w, h = int(rng.integers(min_dim, min_dim + 40)), int(rng.integers(1500, 3500))
        if rng.random() < 0.5:
            w, h = h, w
    if fast or rng.random() < 0.5:
        w = (w + 15) // 16 * 16
    if levels == 2:
        w += w & 1
        h += h & 1
    return dict(width=w, height=h, tile=tile, search=search, pyramid_levels=levels, grid_mode=grid_mode,
                subpixel=subpixel, mean_subtract=int(rng.integers(0, 2)), hist_filter=int(rng.integers(0, 2)),
                feature_threshold=int(rng.choice([0, 30, 30, 200, 2000])),
                value_threshold=int(rng.choice([0, 500, 3000, 3000, 70000])) * (4 if tile == 16 else 1),
                min_valid=int(rng.choice([0, 10, 10, 500])), num_blocks=int(rng.integers(2, 9)))


def many_pairs(n_cases, seed0):
    """The fused coarse kernel (k_coarse) is persistent: a workgroup walks pairs i, i + CUs, ... and lets
    the next pair's stream phase start under the current pair's last step.  Two pairs per call never get
    there: these cases run several hundred small dense two-level pairs per call, every pair against the
    oracle, the workspace's level-1 flows and sums against the separate kernels."""
    dev = torch.device("cuda:0")
    t0 = time.time()
    for s in range(seed0, seed0 + n_cases):
        rng = np.random.default_rng(770000 + s)
        w, h = 16 * int(rng.integers(4, 13)), 2 * int(rng.integers(26, 70))
        kw = dict(width=w, height=h, pyramid_levels=2, mean_subtract=int(rng.integers(0, 2)), hist_filter=int(rng.integers(0, 2)),
                  feature_threshold=int(rng.choice([0, 30, 200])), value_threshold=int(rng.choice([500, 3000, 70000])),
                  min_valid=int(rng.choice([0, 10])))
        p = aof.default_params(**kw)
        if aof.check_params(p) != 0:
            continue
        k, n = 12, int(rng.integers(300, 900))
        hp, hc, _ = synth.make_batch(w, h, k, 9, 660000 + 5 * s, noise=int(rng.choice([0, 3, 25])),
                                     brightness=int(rng.integers(-30, 31)))
        ip, ic = rng.integers(0, k, n), rng.integers(0, k, n)
        same = rng.random(n) < 0.7
        ic = np.where(same, ip, ic)          # most pairs match, some compare unrelated frames
        prevs, curs = hp[ip], hc[ic]
        po = orc.params_from(p)
        cache = {}
        tp, tc = torch.from_numpy(prevs).to(dev), torch.from_numpy(curs).to(dev)
        outs = []
        for split in (False, True):
            eng = aof.FlowEngine(p, 0)
            eng.set_split_coarse(split)
            blocks, flows, ws = eng.flow_batch(tp, tc)
            torch.cuda.synchronize()
            L = aof.workspace_layout(p, n)
            wsn = ws.cpu().numpy()
            outs.append((aof.blocks_view(blocks).copy(), aof.flows_view(flows).copy(), wsn[L.l1_flows:L.l1_flows + 16 * n].tobytes(),
                         wsn[L.sums:L.sums + 16 * n].tobytes() if p.mean_subtract else b""))
            eng.close()
        gb, gf = outs[0][0], outs[0][1]
        for i in range(n):
            key = (int(ip[i]), int(ic[i]))
            if key not in cache:
                cache[key] = orc.flow_pair(po, prevs[i], curs[i])
            ref = cache[key]
            if gb[i].tobytes() != ref["blocks"].tobytes() or gf[i].tobytes() != ref["flow"].tobytes():
                print(f"MISMATCH many-pairs seed {s} pair {i} of {n}: {kw}", flush=True)
                sys.exit(1)
        if outs[0][2] != outs[1][2] or outs[0][3] != outs[1][3] or outs[0][1].tobytes() != outs[1][1].tobytes():
            print(f"MISMATCH many-pairs seed {s}: fused and separate kernels differ in the workspace: {kw}", flush=True)
            sys.exit(1)
        if (s - seed0 + 1) % 20 == 0:
            print(f"{s - seed0 + 1} many-pairs cases ok ({time.time() - t0:.0f} s)", flush=True)
    print(f"many-pairs fuzz passed: {n_cases} cases of 300..900 pairs per call, {time.time() - t0:.0f} s")


def resident(n_cases, seed0):
    """The per-call path served by the resident kernel (aof_set_stream_resident): random small
    configurations, sequences of frames pushed one by one with random stream resets, frames that take the
    launch-per-call path in between, kernel switches and pauses beyond the idle deadline -- every record
    against the oracle."""
    t0 = time.time()
    for s in range(seed0, seed0 + n_cases):
        rng = np.random.default_rng(330000 + s)
        kw = small_case(rng)
        if kw["width"] * kw["height"] > 64 * 1024:   # (the resident path takes frames of at most 64 KB)
            kw["height"] = max(2 * ((64 * 1024 // kw["width"]) // 2), 36 * (2 if kw["pyramid_levels"] == 2 else 1))
        p = aof.default_params(**kw)
        if aof.check_params(p) != 0:
            continue
        g0 = aof.grid(p, 0)
        g1 = aof.grid(p, 1) if p.pyramid_levels == 2 else None
        if not small_eligible(p, g0, g1) or p.width * p.height > 64 * 1024:
            continue
        reach = 9 if p.pyramid_levels == 2 else 4
        frames, _ = synth.make_sequence(p.width, p.height, 14, reach, seed=s, max_step=reach - 1)
        po = orc.params_from(p)
        eng = aof.FlowEngine(p, 0)
        eng.set_stream_resident(bool(rng.random() < 0.6))   # (off: the replayed graph whose tagged record the host polls for)
        prev = None
        for k in range(14):
            r = rng.random()
            if rng.random() < 0.06:
                eng.set_stream_graph(bool(rng.random() < 0.5))   # eager launches + stream wait / the graph again
            if rng.random() < 0.05 and k >= 2:   # the other host entry point shares the stream and the pinned record
                _, _, pair = eng.flow_pair_host(frames[k - 2], frames[k - 1])
                if pair.tobytes() != orc.flow_pair(po, frames[k - 2], frames[k - 1])["flow"].tobytes():
                    print(f"MISMATCH pair entry point seed {s} frame {k}: {kw}", flush=True)
                    sys.exit(1)
            if r < 0.08:
                eng.stream_reset()
                prev = None
            elif r < 0.16:
                eng.set_stream_resident(False)
            elif r < 0.3:
                eng.set_stream_resident(True)
            elif r < 0.34:
                eng.force_generic(True)
            elif r < 0.4:
                eng.force_generic(False)
            elif r < 0.43:
                time.sleep(0.07)
            got = eng.stream_push(frames[k])
            if prev is None:
                assert got is None
            elif got.tobytes() != orc.flow_pair(po, frames[prev], frames[k])["flow"].tobytes():
                print(f"MISMATCH resident seed {s} frame {k} after {prev}: {kw}", flush=True)
                sys.exit(1)
            prev = k
        eng.close()
        if (s - seed0 + 1) % 50 == 0:
            print(f"{s - seed0 + 1} resident cases ok ({time.time() - t0:.0f} s)", flush=True)
    print(f"per-call fuzz passed: {n_cases} sequences of 14 frames, {time.time() - t0:.0f} s")


def sequence(n_cases, seed0):
    """aof_sequence_device over random recordings: random small configurations (one and two levels, sparse and
    dense grids), output rates from "every frame" to slower than the recording, time stamps with jitter, stalls
    and 32-bit wrap-arounds, dark stretches, gyro increments -- records and MAVLink frames against the oracle's
    calcFlow chain and the independent serializer of tests/test_mavlink.py."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_mavlink import py_frame
    dev = torch.device("cuda:0")
    t0 = time.time()
    done = 0
    for s in range(seed0, seed0 + n_cases):
        rng = np.random.default_rng(880000 + s)
        kw = small_case(rng)
        p = aof.default_params(**kw)
        if aof.check_params(p) != 0:
            continue
        n = int(rng.integers(1, 70))
        if rng.random() < 0.12:
            n = int(rng.integers(130, 330))   # more than 128 pairs: the separate kernels, K1's outputs left by the ingest kernel
        cam_w, cam_h = p.width + 2 * int(rng.integers(0, 20)), p.height + 2 * int(rng.integers(0, 20))
        reach = 9 if p.pyramid_levels == 2 else 4
        frames, _ = synth.make_sequence(cam_w, cam_h, max(n, 2), reach, seed=s, max_step=reach - 1)
        frames = frames[:n].copy()
        if n > 8 and rng.random() < 0.4:
            a = int(rng.integers(0, n - 4))
            frames[a:a + int(rng.integers(1, 5))] = int(rng.integers(0, 256))       # flat frames: quality 0
        steps = rng.integers(3000, 30000, n)
        if rng.random() < 0.3:
            steps[rng.integers(0, n, 3)] = 0                                        # repeated time stamps
        times = np.concatenate([[0], np.cumsum(steps[1:])]).astype(np.int64)
        if n > 4 and rng.random() < 0.4:
            k = int(rng.integers(1, n))
            times[k:] += (1 << 32) - int(times[k]) - int(rng.integers(0, 60000))   # the 32-bit time stamp wraps
        if rng.random() < 0.2:
            times += int(rng.integers(0, 1 << 33))                                  # not relative to the first frame at all
        rate = int(rng.choice([0, -3, 1, 5, 15, 15, 30, 75, 500]))
        offset = 0 if rng.random() < 0.1 else int(rng.integers(1, 1 << 50))
        first_seq = int(rng.integers(0, 256))
        gyro = np.zeros((n, 4), np.float32)
        gyro[:, :3] = rng.normal(0, 0.01, (n, 3)).astype(np.float32)
        gyro[:, 3] = rng.uniform(0, 0.05, n).astype(np.float32)
        fx, fy = float(np.float32(rng.uniform(50, 900))), float(np.float32(rng.uniform(50, 900)))
        sp = aof.sequence_params(cam_w, cam_h, p.width, p.height, fx, fy, rate, offset, int(rng.integers(1, 255)), int(rng.integers(1, 255)),
                                 first_seq, derotate=(reach + 0.5, 0.01))
        eng = aof.FlowEngine(p, 0)
        use_gyro = bool(rng.random() < 0.85)
        if not use_gyro:
            sp.derotate = 0
        ws, L = eng.sequence(sp, torch.from_numpy(frames).to(dev), torch.from_numpy(times).to(dev),
                             torch.from_numpy(gyro).to(dev) if use_gyro else None)
        torch.cuda.synchronize()
        out = eng.sequence_outputs(sp, ws, L, n)
        eng.close()
        x0, y0 = cam_w // 2 - p.width // 2, cam_h // 2 - p.height // 2
        cropped = np.ascontiguousarray(frames[:, y0:y0 + p.height, x0:x0 + p.width])
        o = orc.Px4(orc.params_from(p), fx, fy, rate)
        g = np.zeros(3, np.float64)
        m = 0
        bad = None
        for k in range(n):
            if use_gyro:
                g += gyro[k, :3].astype(np.float64)
            q, dt, ax, ay = o.calc_flow(cropped[k], int(times[k]) & 0xFFFFFFFF)
            if q < 0:
                continue
            taken, g = g.copy(), np.zeros(3, np.float64)
            if m >= len(out["records"]):
                bad = f"record {m} missing"
                break
            r = out["records"][m]
            want = (k, q, dt, np.float32(ax).tobytes(), np.float32(ay).tobytes(), np.float32(taken).tobytes())
            got = (int(r["frame"]), int(r["quality"]), int(r["dt_us"]), r["flow_x"].tobytes(), r["flow_y"].tobytes(),
                   np.array([r["gyro_x"], r["gyro_y"], r["gyro_z"]], np.float32).tobytes())
            if got != want:
                bad = f"record {m}: device {got[:3]} oracle {want[:3]}"
                break
            if offset:
                sid, cid = sp.system_id, sp.component_id
                f = bytearray(py_frame(offset, int(times[k]), dt, float(np.float32(ax)), float(np.float32(ay)),
                                       tuple(float(v) for v in taken), q, (first_seq + m) & 0xFF))
                # (py_frame writes the reference's ids 1 / 100: patch ours in and redo the checksum)
                f[5], f[6] = sid, cid
                from test_mavlink import x25
                crc = x25(bytes([138]), x25(bytes(f[1:-2])))
                f[-2], f[-1] = crc & 0xFF, crc >> 8
                if out["mavlink"][m] != bytes(f):
                    bad = f"frame {m} differs"
                    break
            elif len(out["mavlink"][m]) != 0:
                bad = f"frame {m} sent without a vehicle time"
                break
            m += 1
        if bad is None and m != len(out["records"]):
            bad = f"{len(out['records'])} records, oracle {m}"
        if bad is None and out["status"] != 0:
            bad = f"status {out['status']}"
        if bad:
            print(f"MISMATCH sequence seed {s}: {bad}; n {n} rate {rate} cam {cam_w}x{cam_h} {kw}", flush=True)
            sys.exit(1)
        done += 1
        if done % 50 == 0:
            print(f"{done} sequence cases ok ({time.time() - t0:.0f} s)", flush=True)
    print(f"sequence fuzz passed: {done} recordings of 1..69 (one in eight: 130..329) frames, {time.time() - t0:.0f} s")


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    if len(sys.argv) > 3 and sys.argv[3] == "sequence":
        return sequence(n_cases, seed0)
    if len(sys.argv) > 3 and sys.argv[3] == "many":
        return many_pairs(n_cases, seed0)
    if len(sys.argv) > 3 and sys.argv[3] == "resident":
        return resident(n_cases, seed0)
    dev = torch.device("cuda:0")
    t0, done, skipped, variants = time.time(), 0, 0, {}
    for s in range(seed0, seed0 + n_cases):
        rng = np.random.default_rng(50000 + s)
        kw = case(rng)
        p = aof.default_params(**kw)
        if aof.check_params(p) != 0:
            skipped += 1
            continue
        reach = 2 * p.search + 1 if p.pyramid_levels == 2 else p.search
        n = 2
        prevs, curs, _ = synth.make_batch(p.width, p.height, n, reach, 90000 + 3 * s, noise=int(rng.choice([0, 0, 3, 25])),
                                          brightness=int(rng.integers(-40, 41)), contrast=float(rng.choice([1.0, 1.0, 3.0, 0.15])))
        style = int(rng.integers(0, 7))
        if style == 5:  # half-pixel displaced pair: every refinement direction gets its turn
            half = [(1, 0), (-1, 0), (0, 1), (0, -1), (1, 1), (-1, -1), (1, -1), (-1, 1)][s % 8]
            sh = (int(rng.integers(-p.search + 1, p.search)), int(rng.integers(-p.search + 1, p.search)))
            if p.pyramid_levels == 1:
                prevs[0], curs[0], _ = synth.make_pair(p.width, p.height, p.search, 7000 + s, shift=sh, half=half)
        if style == 1:
            curs[0] = rng.integers(0, 256, curs[0].shape, dtype=np.uint8)
        elif style == 2:
            prevs[1][:, : p.width // 2] = int(rng.integers(0, 256))
        elif style == 3:
            curs[1] = prevs[1]
        elif style == 4:
            period = int(rng.choice([2, 3, 4, 8]))
            yy, xx = np.mgrid[0:p.height, 0:p.width]
            prevs[0] = (((xx // period + yy // period) % 2) * int(rng.integers(1, 256))).astype(np.uint8)
            curs[0] = prevs[0] if rng.random() < 0.5 else 255 - prevs[0]
        elif style == 6:
            # the LAST pair displaced by the whole reach towards one of the four corners: blocks on the frame's rim match on
            # the rim of their search window, and rings, windows and tiles of the last pair end where the arrays end
            sx, sy = [(1, 1), (-1, -1), (1, -1), (-1, 1)][s % 4]
            prevs[n - 1], curs[n - 1], _ = synth.make_pair(p.width, p.height, reach, 8000 + s, shift=(sx * reach, sy * reach),
                                                           noise=int(rng.choice([0, 0, 2])))
        po = orc.params_from(p)
        small = small_eligible(p, aof.grid(p, 0), aof.grid(p, 1) if p.pyramid_levels == 2 else None)
        refs = [orc.flow_pair(po, prevs[i], curs[i]) for i in range(n)]
        tp, tc = torch.from_numpy(prevs).to(dev), torch.from_numpy(curs).to(dev)
        for mode in ("default", "exhaustive", "fused_reduce", "pruned", "pruned_fused_reduce", "generic", "split", "sequence_view"):
            if mode == "split" and p.pyramid_levels != 2 and not small:
                continue   # (the separate kernels instead of k_coarse / k_flow_small)
            if mode == "default" and p.tile != 16:
                continue   # (a launch of two pairs: 8x8 contexts in the adaptive mode search it exhaustively; 16x16 contexts probe)
            if mode == "sequence_view":
                # the two pairs as ONE sequence of three frames viewed twice (K1 once per frame): pair 0 = (prev0, cur0),
                # pair 1 = (cur0, cur1)
                seq = torch.from_numpy(np.stack([prevs[0], curs[0], curs[1]])).to(dev)
                eng = aof.FlowEngine(p, 0)
                if rng.random() < 0.5:
                    eng.set_split_coarse(True)
                b2, f2, _ = eng.flow_batch(seq[:-1], seq[1:], n_pairs=2, pair_stride=p.width * p.height)
                torch.cuda.synchronize()
                r1 = orc.flow_pair(po, curs[0], curs[1])
                gb2, gf2 = aof.blocks_view(b2), aof.flows_view(f2)
                if (gb2[0].tobytes() != refs[0]["blocks"].tobytes() or gf2[0].tobytes() != refs[0]["flow"].tobytes() or
                        gb2[1].tobytes() != r1["blocks"].tobytes() or gf2[1].tobytes() != r1["flow"].tobytes()):
                    print(f"MISMATCH seed {s} mode sequence_view ({eng.variant}): {kw}", flush=True)
                    sys.exit(1)
                eng.close()
                continue
            eng = aof.FlowEngine(p, 0)
            if mode == "exhaustive":
                eng.set_search_mode(aof.SEARCH_EXHAUSTIVE)
            if mode == "split":
                eng.set_split_coarse(True)
            elif mode == "generic":
                eng.force_generic(True)
            elif mode == "pruned":
                eng.set_search_mode(aof.SEARCH_PRUNED)
            elif mode == "fused_reduce":
                eng.set_reduce_fusion(True)
            elif mode == "pruned_fused_reduce":   # (dense grids: the column walk that reduces in its launch, k_flow_lane8_cols)
                eng.set_search_mode(aof.SEARCH_PRUNED)
                eng.set_reduce_fusion(True)
            nb = eng.nblocks(0)
            # every output carved out of one arena with guard zones in between: a kernel that writes
            # outside its buffers is caught even when the records it returns are right
            L = aof.workspace_layout(p, n)
            G = 1024
            sizes = [L.total_bytes, n * nb * 4, n * 16, n * nb]
            offs, off = [], G
            for sz in sizes:
                offs.append(off)
                off = (off + sz + G + 255) // 256 * 256
            arena = torch.full((off,), 0xAB, dtype=torch.uint8, device=dev)
            view = lambda k: arena[offs[k]:offs[k] + sizes[k]]
            sub = view(3).view(n, nb) if p.subpixel else None
            if sub is not None:
                sub.fill_(99)
            blocks, flows, _ = eng.flow_batch(tp, tc, blocks=view(1).view(torch.int32).view(n, nb), flows=view(2).view(n, 16),
                                              subdirs=sub, workspace=view(0))
            torch.cuda.synchronize()
            mask = torch.ones(off, dtype=torch.bool, device=dev)
            for k in range(4):
                mask[offs[k]:offs[k] + sizes[k]] = False
            if not p.subpixel:
                mask[offs[3]:offs[3] + sizes[3]] = True
            if not bool((arena[mask] == 0xAB).all()):
                hit = torch.nonzero(mask & (arena != 0xAB)).reshape(-1).cpu().numpy()
                print(f"GUARD HIT seed {s} mode {mode} ({eng.variant}): {hit.size} bytes outside the buffers, arena offsets "
                      f"{hit[:6]} .. {hit[-3:]}; buffers at {offs} sizes {sizes}: {kw}", flush=True)
                sys.exit(1)
            gb, gf = aof.blocks_view(blocks), aof.flows_view(flows)
            if True:   # ... or the frames
                for name, t, ref in (("prev", tp, prevs), ("cur", tc, curs)):
                    got = t.cpu().numpy()
                    if not np.array_equal(got, ref):
                        dd = np.nonzero(got.reshape(-1) != ref.reshape(-1))[0]
                        print(f"FRAME CORRUPTED seed {s} after mode {mode}: {name} differs at {dd.size} bytes, offsets "
                              f"{dd[:8]} .. {dd[-3:]}, device bytes {got.reshape(-1)[dd[:24]]}: {kw}")
                        print(f"  pointers: prev {tp.data_ptr():#x} cur {tc.data_ptr():#x} blocks {blocks.data_ptr():#x} "
                              f"flows {flows.data_ptr():#x} sub {sub.data_ptr() if sub is not None else 0:#x} ", flush=True)
                        sys.exit(1)
            for i in range(n):
                ok = gb[i].tobytes() == refs[i]["blocks"].tobytes() and gf[i].tobytes() == refs[i]["flow"].tobytes()
                if ok and sub is not None:
                    ok = bool(np.array_equal(sub[i].cpu().numpy(), refs[i]["subdirs"]))
                if not ok:
                    print(f"MISMATCH seed {s} mode {mode} ({eng.variant}) pair {i} style {style}: {kw}", flush=True)
                    d = np.nonzero(gb[i].view(np.uint32) != refs[i]["blocks"].view(np.uint32))[0]
                    print(f"  block records differ at {d[:8]}: gpu {gb[i][d[:4]]} oracle {refs[i]['blocks'][d[:4]]}")
                    print(f"  flow gpu {gf[i]} oracle {refs[i]['flow']}")
                    if sub is not None:
                        a = sub[i].cpu().numpy()
                        d = np.nonzero(a != refs[i]["subdirs"])[0]
                        print(f"  directions differ at {d[:8]}: gpu {a[d[:8]]} oracle {refs[i]['subdirs'][d[:8]]}")
                    print(f"  device frames intact: prev {bool(np.array_equal(tp.cpu().numpy(), prevs))} "
                          f"cur {bool(np.array_equal(tc.cpu().numpy(), curs))}")
                    dd = np.nonzero(tc.cpu().numpy().reshape(-1) != curs.reshape(-1))[0]
                    if dd.size:
                        print(f"  cur differs at {dd.size} bytes, offsets {dd[:6]} .. {dd[-3:]}; device bytes {tc.cpu().numpy().reshape(-1)[dd[:16]]}")
                    print(f"  pointers: prev {tp.data_ptr():#x} cur {tc.data_ptr():#x} blocks {blocks.data_ptr():#x} "
                          f"flows {flows.data_ptr():#x}", flush=True)
                    sys.exit(1)
            name = "small_lds" if small and mode in ("exhaustive", "pruned") else eng.variant
            variants[name] = variants.get(name, 0) + 1
            eng.close()
        # 8x8 contexts on grids the flat kernel serves: the ADAPTIVE default on a launch large enough for the pruned kernel --
        # the two pairs replicated, interleaved, to 2 048 chunks of 256 blocks and more --, three launches (nothing known yet:
        # every wave's first chunk judges; then what the reports said), every replica against the oracle
        nb0 = aof.grid(p, 0)[4] * aof.grid(p, 0)[5]
        if p.tile == 8 and p.search == 4 and nb0 > 256 and rng.random() < 0.35:
            reps = (2048 * 256 + n * nb0 - 1) // (n * nb0) + 1
            if 2 * reps * n * p.width * p.height <= (3 << 29):
                eng = aof.FlowEngine(p, 0)
                rp, rc = tp.repeat(reps, 1, 1).contiguous(), tc.repeat(reps, 1, 1).contiguous()
                subs = torch.full((reps * n, nb0), 99, dtype=torch.uint8, device=dev) if p.subpixel else None
                for launch in range(3):
                    blocks, flows, _ = eng.flow_batch(rp, rc, subdirs=subs)
                    torch.cuda.synchronize()
                    gb, gf = aof.blocks_view(blocks), aof.flows_view(flows)
                    gs = subs.cpu().numpy() if subs is not None else None
                    for i in range(reps * n):
                        r = refs[i % n]
                        ok = gb[i].tobytes() == r["blocks"].tobytes() and gf[i].tobytes() == r["flow"].tobytes()
                        if ok and gs is not None:
                            ok = gs[i].tobytes() == r["subdirs"].tobytes()
                        if not ok:
                            print(f"MISMATCH seed {s} mode adaptive_replicas launch {launch} replica {i} ({eng.search_stats()}): {kw}", flush=True)
                            sys.exit(1)
                st = eng.search_stats()
                if st["pruned_launches"] < 1:
                    print(f"seed {s}: the adaptive launch of {reps * n} pairs never ran the pruned kernel: {st}: {kw}", flush=True)
                    sys.exit(1)
                variants["lane8_adaptive_replicas"] = variants.get("lane8_adaptive_replicas", 0) + 1
                eng.close()
                del rp, rc, subs
        done += 1
        if done % 25 == 0:
            print(f"{done} cases ok ({time.time() - t0:.0f} s), skipped {skipped}, kernels {variants}", flush=True)
    print(f"fuzz passed: {done} cases x 7 device paths (+ the adaptive default on 16x16 cases, the separate kernels on two-level and "
          f"small-pair cases), {skipped} skipped, kernels {variants}, {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
