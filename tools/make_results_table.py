#!/usr/bin/env python3
"""Markdown table of the committed bench lines profiles/<prefix>_bench_*.json (README / DESIGN)."""
import glob
import json
import os
import sys

prefix = sys.argv[1] if len(sys.argv) > 1 else "profiles/r02_final"
order = ["c2", "c3", "c2h", "c5", "c5h", "c1b", "ingest", "derotate"]
names = {"c2": "`c2` 640×480, 8×8 SAD, ±4, dense grid — the headline", "c3": "`c3` = c2 + 2-level pyramid + mean equalisation",
         "c2h": "`c2h` = c2 + half-pixel refinement", "c5": "`c5` 1280×960, 16×16 SAD, ±8 (256 pairs per launch)",
         "c5h": "`c5h` = c5 + half-pixel refinement", "c1b": "`c1b` 64×64, published sparse grid + half-pixel, 65 536 pairs per launch",
         "ingest": "`ingest` 640×480 sensor frames → 128×128 crop + exposure histogram", "derotate": "`derotate` gyro de-rotation of flow records"}
print("| workload (`bench.py --workload`) | throughput | step | whole step vs 8 TB/s | dominant kernel: time, vs 8 TB/s | beyond-L2 traffic / algorithmic | exact-pruned (opt-in) | CPU oracle |")
print("|---|---|---|---|---|---|---|---|")
for w in order:
    f = f"{prefix}_bench_{w}.json"
    if not os.path.exists(f):
        continue
    j = json.loads(open(f).read().strip().splitlines()[-1])
    r = j["roofline"]
    unit = j["unit"].replace("frame-pairs/s", "pairs/s")
    v = j["value"]
    val = f"{v/1e6:.2f} M {unit}" if v < 1e9 else f"{v/1e9:.1f} G {unit}"
    step = f"{j['ms_per_step']:.4f} ms"
    fs = f"**{100*r['frac_step']:.1f} %**" if r.get("frac_step") else "—"
    per = r.get("pairs_per_launch") or r.get("frames_per_launch") or r.get("records_per_launch")
    alg = (r.get("algorithmic_bytes_per_pair") or r.get("algorithmic_bytes_per_frame") or r.get("algorithmic_bytes_per_record")) * per
    t = r.get("traffic_step") or r.get("traffic")
    tr = f"{t/alg:.2f}×" if t else "—"
    dom = f"{r['kernel'].split(' ')[0]} {r['kernel_ms']*1e3:.1f} µs, {100*r['frac']:.1f} %"
    pr = j.get("exact_pruned_search")
    prs = f"{pr['per_gpu_value']/1e6:.2f} M ({100*pr['roofline_frac']:.1f} %)" if pr else "—"
    cb = j.get("cpu_baseline")
    cbs = f"{cb['value']:,.0f} /s on {cb['cores']} core{'s' if cb['cores'] > 1 else ''}" if cb else "—"
    print(f"| {names[w]} | {val} | {step} | {fs} | {dom} | {tr} | {prs} | {cbs} |")
