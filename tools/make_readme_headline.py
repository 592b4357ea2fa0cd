#!/usr/bin/env python3
"""Rewrites the headline paragraph of README.md (between the headline:begin / headline:end markers) from
the evidence files profiles/<prefix>_bench_*.json and profiles/<round>_stream_latency.txt.
    python tools/make_readme_headline.py r03_final"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prefix = sys.argv[1] if len(sys.argv) > 1 else "r05_final"
rnd = prefix.split("_")[0]


def load(name):
    return json.loads(open(os.path.join(ROOT, "profiles", f"{prefix}_bench_{name}.json")).read().strip().splitlines()[-1])


c2, c3, ing, c5, seq = load("c2"), load("c3"), load("ingest"), load("c5"), load("seq")
l3, l1b, share, two = load("lanes_c3"), load("lanes_c1b"), load("share_p128"), load("share_p1024_two_batches")
lat = [l for l in open(os.path.join(ROOT, "profiles", f"{rnd}_stream_latency.txt")) if "lane8" in l]


def us(w, levels, resident):
    for l in lat:
        if l.startswith(f"{w}x{w} levels={levels} ") and f"resident={resident}" in l and " graph=1 " in l:
            return float(re.search(r": ([\d.]+) us", l).group(1))
    raise SystemExit("latency line missing")


rf = c2["roofline"]
ex = c2["exhaustive_search"]
re_in = c2.get("realistic_input", {}).get("default_search")
sx = load("share_p1024_exhaustive_two_batches")
one = load("share_p1024")
c3n, c3r = load("c3_noise16"), load("c3_realistic")
c5r, c5h = load("c5_realistic"), load("c5h")
txt = (f"Headline (`bench.py`, C2: 1 024 VGA pairs per launch, 8×8 SAD, ±4, the exact-adaptive search every context runs by default, on BASELINE's "
       f"synthetic translations): **{c2['value']/1e6:.2f} M frame-pairs/s**, "
       f"K2 {rf['kernel_ms']*1e3:.1f} µs per launch = **{rf['frac']*100:.1f} % of the 8 TB/s HBM roofline** (whole step {rf['frac_step']*100:.1f} %), "
       f"{two['value']/1e6:.2f} M with two batches in flight.  That rate depends on the input; the line carries the other two beside it: the exhaustive scan of "
       f"the same batch — same records bit for bit, the rate no input lowers — **{ex['per_gpu_value']/1e6:.2f} M = {ex['roofline_frac']*100:.1f} %** (VALU-bound at "
       f"≈ 80 % of its SAD-issue floor, which caps it at 48 %), and the default search on a realistic input (±4 LSB noise, half-pixel motion, the newer frame at "
       f"half contrast) " + (f"**{re_in['value']/1e6:.2f} M = {re_in['frac_step']*100:.1f} %** (whole step; the context settles on the exhaustive kernel by itself)" if re_in else "—") +
       f"; ±8 LSB of noise alone still prunes (`profiles/{rnd}_final_c2_noise.txt`).  "
       f"C3 (two-level pyramid + equalisation) {c3['value']/1e6:.2f} M pairs/s = {c3['roofline']['frac_step']*100:.1f} % of the roofline over the whole step "
       f"({c3n['value']/1e6:.2f} M at ±16 LSB, {c3r['value']/1e6:.2f} M realistic), {l3['value']/1e6:.2f} M = **{l3['roofline']['frac_step']*100:.1f} %** with two "
       f"batches in flight; configs[3]'s per-GPU share of 128 pairs — which prunes since round 5 — takes **{share['ms_per_step']*1e3:.1f} µs** per step against "
       f"{one['ms_per_step']*1e3:.1f} µs for all 1 024 pairs on one GPU (**{one['ms_per_step']/share['ms_per_step']:.1f}×**; {two['ms_per_step']/share['ms_per_step']:.1f}× "
       f"against its fastest, two batches in flight; {sx['ms_per_step']*1e3:.1f} µs and {sx['ms_per_step']/load('share_p128_noise16')['ms_per_step']:.1f}× on inputs that do not prune); "
       f"C5 (1280×960, 16×16 SAD, ±8) {c5['value']/1e6:.2f} M pairs/s = **{c5['roofline']['frac']*100:.1f} %** with the exact-adaptive search "
       f"(exhaustive: {c5['exhaustive_search']['per_gpu_value']/1e6:.2f} M = {c5['exhaustive_search']['roofline_frac']*100:.1f} %, which is also what the realistic input gets: "
       f"{c5r['value']/1e6:.2f} M), with half-pixel refinement {c5h['value']/1e6:.2f} M = {c5h['roofline']['frac_step']*100:.1f} %; "
       f"a recording of {seq['config']['frames_per_gpu']:,} sensor frames through the whole per-frame loop on the device (`aof_sequence_device`) "
       f"{seq['value']/1e6:.0f} M frames/s; "
       f"configs[0] in batch {l1b['value']/1e6:.0f} M pairs/s; frame ingest {ing['value']/1e6:.0f} M frames/s "
       f"({ing['roofline']['frac']*100:.0f} % of the roofline, ≈ 92 % of what a plain copy of the same row pieces reaches); "
       f"one `calcFlow()` call **{us(64, 1, 0):.2f} µs at 64×64 / {us(128, 2, 0):.2f} µs at 128×128 on two levels** through a replayed hipGraph "
       f"whose tagged record the host polls for (the CPU oracle: 16 µs on one core), "
       f"{us(64, 1, 1):.2f} / {us(128, 2, 1):.2f} µs served by the resident kernel (opt-in).")
path = os.path.join(ROOT, "README.md")
s = open(path).read()
a = s.index("<!-- headline:begin -->") + len("<!-- headline:begin -->")
b = s.index("<!-- headline:end -->")
open(path, "w").write(s[:a] + "\n" + txt + "\n" + s[b:])
print(txt)
