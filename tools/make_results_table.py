#!/usr/bin/env python3
"""Markdown tables of the committed bench lines profiles/<prefix>_bench_*.json, written between the
<!-- results:begin/end --> and <!-- share:begin/end --> markers of DESIGN.md (idempotent).
    python tools/make_results_table.py [profiles/r03_final]"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prefix = sys.argv[1] if len(sys.argv) > 1 else "profiles/r04_final"
order = ["c2", "c3", "c2h", "c5", "c5h", "c1b", "ingest", "derotate", "seq"]
names = {"c2": "`c2` 640×480, 8×8 SAD, ±4, dense grid — the headline (exact-adaptive search, the default)", "c3": "`c3` = c2 + 2-level pyramid + mean equalisation",
         "c2h": "`c2h` = c2 + half-pixel refinement", "c5": "`c5` 1280×960, 16×16 SAD, ±8 (256 pairs per launch), exact-adaptive search",
         "c5h": "`c5h` = c5 + half-pixel refinement", "c1b": "`c1b` 64×64, published sparse grid + half-pixel, 65 536 pairs per launch",
         "ingest": "`ingest` 640×480 sensor frames → 128×128 crop + exposure histogram", "derotate": "`derotate` gyro de-rotation of flow records",
         "seq": "`seq` a recording of 65 536 sensor frames 320×240 through `aof_sequence_device` (crop 128×128, two levels, limiter, de-rotation, MAVLink frames)"}


def load(tag):
    f = os.path.join(ROOT, f"{prefix}_bench_{tag}.json")
    if not os.path.exists(f):
        return None
    return json.loads(open(f).read().strip().splitlines()[-1])


def fmt_value(j):
    unit = j["unit"].replace("frame-pairs/s", "pairs/s")
    v = j["value"]
    return f"{v/1e6:.2f} M {unit}" if v < 1e9 else f"{v/1e9:.1f} G {unit}"


def results():
    out = ["| workload (`bench.py --workload`) | throughput | step | whole step vs 8 TB/s | dominant kernel: time, vs 8 TB/s | beyond-L2 traffic / algorithmic | two batches in flight (`--streams 2`) | other search modes on the same batch | CPU oracle |",
           "|---|---|---|---|---|---|---|---|---|"]
    for w in order:
        j = load(w)
        if not j:
            continue
        r = j["roofline"]
        step = f"{j['ms_per_step']:.4f} ms"
        fs = f"**{100*r['frac_step']:.1f} %**" if r.get("frac_step") else "—"
        per = r.get("pairs_per_launch") or r.get("frames_per_launch") or r.get("records_per_launch")
        alg = (r.get("algorithmic_bytes_per_pair") or r.get("algorithmic_bytes_per_frame") or r.get("algorithmic_bytes_per_record")) * per
        t = r.get("traffic_step") or r.get("traffic")
        tr = f"{t/alg:.2f}×" if t else "—"
        dom = f"{r['kernel'].split(' ')[0]} {r['kernel_ms']*1e3:.1f} µs, {100*r['frac']:.1f} %"
        others = []
        for key, label in (("exhaustive_search", "exhaustive"), ("exact_pruned_search", "exact-pruned"), ("exact_adaptive_search", "exact-adaptive")):
            pr = j.get(key)
            if pr:
                others.append(f"{label} {pr['per_gpu_value']/1e6:.2f} M ({100*pr['roofline_frac']:.1f} %)")
        prs = "; ".join(others) if others else "—"
        cb = j.get("cpu_baseline")
        cbs = "—"
        if cb:
            cbs = f"{cb['value']:,.0f} /s on {cb['cores']} core{'s' if cb['cores'] > 1 else ''}"
            if cb.get("single_thread"):
                cbs += f", {cb['single_thread']['value']:,.0f} /s on one"
        l2 = load("lanes_" + w)
        ls = "—"
        if l2:
            ls = f"{fmt_value(l2)}"
            if l2["roofline"].get("frac_step"):
                ls += f" (**{100*l2['roofline']['frac_step']:.1f} %**)"
        out.append(f"| {names[w]} | {fmt_value(j)} | {step} | {fs} | {dom} | {tr} | {ls} | {prs} | {cbs} |")
    c1 = load("c1")
    if c1:
        c = c1["config"]
        out += ["", f"Per `calcFlow()` call through the Python harness (`bench.py --workload c1`; from C++: `profiles/{os.path.basename(prefix).split('_')[0]}_stream_latency.txt`): "
                f"64×64 one level {c['us_per_call']} µs per call replayed as a hipGraph, **{c.get('resident_kernel_us_per_call')} µs served by the resident kernel**; "
                f"`OpticalFlowOpenCV` at 128×128 (two levels + equalisation) {c.get('opencv_facade_128x128_two_levels_us_per_call')} µs, "
                f"**{c.get('opencv_facade_128x128_two_levels_resident_kernel_us_per_call')} µs resident**; the CPU oracle on one core: "
                f"{1e6/c1['cpu_baseline']['value']:.1f} µs." if c1.get("cpu_baseline") else ""]
    return "\n".join(out)


def share():
    rows = [("share_p1024", "1 024 pairs, one batch in flight (the headline configuration: adaptive search, prunes)"),
            ("share_p1024_two_batches", "1 024 pairs, two batches in flight"),
            ("share_p1024_exhaustive", "1 024 pairs, `--search exhaustive` (the search the 128-pair shares run), one batch in flight"),
            ("share_p1024_exhaustive_two_batches", "the same, two batches in flight"),
            ("share_p512", "512 pairs (a 2-GPU share) as `bench.py --pairs 512` chooses: two batches in flight, pruned search + K3"),
            ("share_p256", "256 pairs (a 4-GPU share), the same choice"),
            ("share_p128_one_batch_separate", "128 pairs, one batch in flight, separate K3 (round 2's structure + the 1 024-lane K3, graph replay)"),
            ("share_p128_one_batch_fused", "128 pairs, one batch in flight, reduction in the search launch"),
            ("share_p128_two_batches_separate", "128 pairs, two batches in flight, separate K3"),
            ("share_p128", "128 pairs, `bench.py --pairs 128` as it chooses itself: two batches in flight, graph replay, exhaustive search with the reduction in the launch"),
            ("share_p128_eager", "the same, launched eagerly (`--graph off`)")]
    base = load("share_p1024")
    if not base:
        return "(not collected)"
    def fastest(a, b):
        a, b = load(a), load(b)
        if a and b:
            return a if a["ms_per_step"] <= b["ms_per_step"] else b
        return a or b
    best = fastest("share_p1024", "share_p1024_two_batches")             # the fastest way one GPU runs the 1 024 pairs
    same = fastest("share_p1024_exhaustive", "share_p1024_exhaustive_two_batches") or best   # ... with the shares' search
    out = ["", "", "| step | time per step | pairs/s on one GPU | fastest 1 024-pair step with the SAME (exhaustive) search ÷ this step | fastest 1 024-pair step (adaptive search) ÷ this step |", "|---|---|---|---|---|"]
    for tag, name in rows:
        j = load(tag)
        if not j:
            continue
        r1, r2 = same["ms_per_step"] / j["ms_per_step"], best["ms_per_step"] / j["ms_per_step"]
        big = tag.startswith("share_p1024")
        bold = tag == "share_p128"
        out.append(f"| {name} | {j['ms_per_step']*1e3:.1f} µs | {j['value']/1e6:.2f} M | {'—' if big else (f'**{r1:.2f}×**' if bold else f'{r1:.2f}×')} | "
                   f"{'—' if big else (f'**{r2:.2f}×**' if bold else f'{r2:.2f}×')} |")
    j = load("share_p128")
    if j:
        out += ["", f"Eight GPUs that each take 128 of the 1 024 pairs therefore finish a step in {j['ms_per_step']*1e3:.1f} µs.  One GPU that runs the SAME "
                    f"exhaustive search on all of them takes {same['ms_per_step']*1e3:.1f} µs at its fastest: **{same['ms_per_step']/j['ms_per_step']:.1f}×** before the gather "
                    "(16 KB per rank, asynchronous, overlapped with the next step), against the ≥ 6× `north_star` asks for.  Since round 4 one GPU with all "
                    f"1 024 pairs in one launch does better than that — its adaptive search prunes: {best['ms_per_step']*1e3:.1f} µs — and the 128-pair shares cannot "
                    f"follow (a pruning hint is carried from block to block of a wave, and 128 pairs leave the 4 096 wave slots two or three blocks each: the best pruned step measured at this size, 24.9 µs, beats the exhaustive one by 3 %): against THAT step "
                    f"eight GPUs are {best['ms_per_step']/j['ms_per_step']:.1f}× faster.  The first figure is the scaling of the sharded job (the same computation on both sides); "
                    "the second compares it with a faster single-GPU algorithm that needs launches of 256 pairs and more (a 4-GPU split of the 1 024 pairs still prunes: "
                    f"{load('share_p256')['ms_per_step']*1e3:.1f} µs per 256-pair step).  `bench.py`'s N > 1 line reports both (`configs3.vs_one_gpu_1024_exhaustive`, `.vs_one_gpu_1024`).  "
                    "With one batch in flight the 128-pair share is "
                    "launch-bound (≈ 4.7 µs of every replayed graph are launch gaps, 5 µs the reduction): the two batches in flight are what the target needs."]
    return "\n".join(out)


def put(text, tag, body):
    b, e = f"<!-- {tag}:begin -->", f"<!-- {tag}:end -->"
    block = f"{b}\n{body}\n{e}"
    if b in text:
        return re.sub(re.escape(b) + r".*?" + re.escape(e), lambda m: block, text, flags=re.S)
    return text.replace({"results": "RESULTS_TABLE_PLACEHOLDER", "share": "SHARE_PLACEHOLDER"}[tag], block)


path = os.path.join(ROOT, "DESIGN.md")
text = open(path).read()
text = put(text, "results", results())
text = put(text, "share", share())
open(path, "w").write(text)
print(results())
print(share())
