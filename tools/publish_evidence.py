#!/usr/bin/env python3
"""Copies the summaries tools/collect_evidence.sh left under gpurun_out/evidence/ into profiles/ under the
round's prefix and merges its PMC traffic entries into profiles/pmc_traffic.json (the file bench.py reads
roofline.traffic from).     python tools/publish_evidence.py r03_final"""
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prefix = sys.argv[1] if len(sys.argv) > 1 else "r05_final"
ev, prof = os.path.join(ROOT, "gpurun_out", "evidence"), os.path.join(ROOT, "profiles")
n = 0
for f in sorted(glob.glob(os.path.join(ev, "bench_*.json"))):
    shutil.copy(f, os.path.join(prof, f"{prefix}_{os.path.basename(f)}")); n += 1
for f in sorted(glob.glob(os.path.join(ev, "kernel_stats_*.txt"))):
    tag = os.path.basename(f)[len("kernel_stats_"):-4]
    shutil.copy(f, os.path.join(prof, f"{prefix}_{tag}_kernel_stats.txt")); n += 1
for f in sorted(glob.glob(os.path.join(ev, "pmc_*.txt"))):
    shutil.copy(f, os.path.join(prof, f"{prefix}_{os.path.basename(f)}")); n += 1
lat = os.path.join(ev, "stream_latency.txt")
if os.path.exists(lat):
    head = ("# tools/stream_latency.sh (tools/bench_stream.cpp): microseconds per aof_stream_push_host call from C++, 5000 calls after 50\n"
            "# warm-up calls, with the streaming counters (aof_stream_stats) of each mode.  graph=1: one replayed hipGraph per call (lane8 rows:\n"
            "# ONE kernel whose record arrives tagged in pinned memory, the host polls for the tag instead of waiting for the stream); graph=0:\n"
            "# eager launches + stream wait; resident=1: aof_set_stream_resident, a one-workgroup kernel stays on the device and serves the calls\n"
            "# through a mailbox in pinned memory (no launch per call); launch call = duration of the hipLaunchKernelGGL that started it,\n"
            "# launch->first poll = from that call's return until the kernel's first store into the mailbox was seen (no HIP call in between).\n")
    open(os.path.join(prof, f"{prefix.split('_')[0]}_stream_latency.txt"), "w").write(head + open(lat).read()); n += 1
nz = os.path.join(ev, "search_modes_vs_noise.txt")
if os.path.exists(nz):
    head = ("# tools/search_modes_vs_noise.sh: bench.py at its defaults with +-N LSB of uniform noise added to the newer frame (--noise N); per line the\n"
            "# default (adaptive) search, and the exhaustive and the always-pruned search of the same batch timed behind it (records compared on the\n"
            "# device: identical); `adaptive stats` = aof_search_stats of the headline's context: which kernel its launches ran.\n")
    open(os.path.join(prof, f"{prefix}_c2_noise.txt"), "w").write(head + open(nz).read()); n += 1
ab = os.path.join(ev, "ab_r04.txt")
if os.path.exists(ab):
    shutil.copy(ab, os.path.join(prof, f"{prefix}_ab_r04.txt")); n += 1
src = os.path.join(ev, "pmc_traffic.json")
if os.path.exists(src):
    new = json.load(open(src))
    dst = os.path.join(prof, "pmc_traffic.json")
    data = json.load(open(dst)) if os.path.exists(dst) else {}
    for k, v in new.items():
        v["source"] = f"{prefix}_{v['source']}"
        data[k] = v
    json.dump(data, open(dst, "w"), indent=1, sort_keys=True)
print(f"{n} files published under profiles/{prefix}_*")
