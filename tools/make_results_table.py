#!/usr/bin/env python3
"""Markdown tables of the committed bench lines profiles/<prefix>_bench_*.json, written between the
<!-- results:begin/end --> and <!-- share:begin/end --> markers of DESIGN.md (idempotent).
    python tools/make_results_table.py [profiles/r03_final]"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prefix = sys.argv[1] if len(sys.argv) > 1 else "profiles/r05_final"
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
    out = ["| workload (`bench.py --workload`) | throughput | step | whole step vs 8 TB/s | dominant kernel: time, vs 8 TB/s | beyond-L2 traffic / algorithmic | two batches in flight (`--streams 2`) | other search modes on the same batch | other inputs, default search: ±16 LSB noise; realistic (±4 LSB + half-pixel + half contrast) | CPU oracle |",
           "|---|---|---|---|---|---|---|---|---|---|"]
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
        ins = []
        for tag in (w + "_noise16", w + "_realistic"):
            k = load(tag)
            ins.append(f"{k['value']/1e6:.2f} M ({100*k['roofline']['frac_step']:.1f} %)" if k else "—")
        inp = f"{ins[0]}; {ins[1]}" if any(x != "—" for x in ins) else "—"
        out.append(f"| {names[w]} | {fmt_value(j)} | {step} | {fs} | {dom} | {tr} | {ls} | {prs} | {inp} | {cbs} |")
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
            ("share_p1024_exhaustive", "1 024 pairs, `--search exhaustive` (what every launch runs on inputs that do not prune), one batch in flight"),
            ("share_p1024_exhaustive_two_batches", "the same, two batches in flight"),
            ("share_p512", "512 pairs (a 2-GPU share) as `bench.py --pairs 512` chooses: one batch in flight, pruned search + K3"),
            ("share_p256", "256 pairs (a 4-GPU share): two batches in flight, graph replay, pruned column walk with the reduction in its launch"),
            ("share_p128", "**128 pairs, `bench.py --pairs 128` as it chooses itself**: two batches in flight, graph replay, adaptive search — since round 5 it prunes at this size — with the reduction in its launch (`k_flow_lane8_cols`)"),
            ("share_p128_separate", "the same with K3 as a kernel of its own"),
            ("share_p128_eager", "the same, launched eagerly (`--graph off`)"),
            ("share_p128_one_batch", "128 pairs, one batch in flight"),
            ("share_p128_exhaustive", "128 pairs, `--search exhaustive` with the reduction in the launch (round 4's choice at this size)"),
            ("share_p128_noise16", "128 pairs on ±16 LSB of noise (the adaptive search settles on the exhaustive kernel)"),
            ("share_p128_realistic", "128 pairs on the realistic input (the same)"),
            ("share_p64", "64 pairs (a 16-GPU share; below the pruning threshold: exhaustive)")]
    base = load("share_p1024")
    if not base:
        return "(not collected)"
    def fastest(a, b):
        a, b = load(a), load(b)
        if a and b:
            return a if a["ms_per_step"] <= b["ms_per_step"] else b
        return a or b
    best = fastest("share_p1024", "share_p1024_two_batches")             # the fastest way one GPU runs the 1 024 pairs
    same = fastest("share_p1024_exhaustive", "share_p1024_exhaustive_two_batches") or best   # ... with the exhaustive search
    out = ["", "", "| step | time per step | pairs/s on one GPU | 1 024-pair step, one batch in flight (what `configs3.vs_one_gpu_1024` compares with) ÷ this step | fastest 1 024-pair step (adaptive, two batches in flight) ÷ this step | fastest 1 024-pair step with the exhaustive search ÷ this step |", "|---|---|---|---|---|---|"]
    for tag, name in rows:
        j = load(tag)
        if not j:
            continue
        r0, r1, r2 = base["ms_per_step"] / j["ms_per_step"], best["ms_per_step"] / j["ms_per_step"], same["ms_per_step"] / j["ms_per_step"]
        big = tag.startswith("share_p1024")
        b_ = lambda r, bold: "—" if big else (f"**{r:.2f}×**" if bold else f"{r:.2f}×")
        out.append(f"| {name} | {j['ms_per_step']*1e3:.1f} µs | {j['value']/1e6:.2f} M | {b_(r0, tag == 'share_p128')} | {b_(r1, tag == 'share_p128')} | {b_(r2, tag in ('share_p128_noise16', 'share_p128_realistic', 'share_p128_exhaustive'))} |")
    j, jn = load("share_p128"), load("share_p128_noise16")
    if j:
        txt = (f"Eight GPUs that each take 128 of the 1 024 pairs finish a step in {j['ms_per_step']*1e3:.1f} µs (before the gather: 16 KB per rank, asynchronous, "
               f"overlapped with the next step).  One GPU with all 1 024 pairs in one launch takes {base['ms_per_step']*1e3:.1f} µs with one batch in flight — "
               f"**{base['ms_per_step']/j['ms_per_step']:.1f}×**, the figure `bench.py`'s N > 1 line reports as `configs3.vs_one_gpu_1024` — and "
               f"{best['ms_per_step']*1e3:.1f} µs at its fastest (two batches in flight): **{best['ms_per_step']/j['ms_per_step']:.1f}×**, against the ≥ 6× `north_star` asks for.  "
               "Round 4 stood at 5.2× here: its 128-pair shares could not prune (a wave walked two or three blocks and spent the first one judging exhaustively) while the "
               "1 024-pair launch could.  Since round 5 the first block of a walk votes for its start row instead (108 SAD instructions against 432), launches prune from "
               "2 048 chunks of 256 blocks on (112 VGA pairs), and the column walk reduces in its own launch.")
        if jn:
            txt += (f"  On inputs that do not prune both sides run the exhaustive kernel: {jn['ms_per_step']*1e3:.1f} µs per 128-pair step against "
                    f"{same['ms_per_step']*1e3:.1f} µs for the 1 024 pairs: {same['ms_per_step']/jn['ms_per_step']:.1f}×.")
        txt += ("  A 20 µs step is launch-bound with one batch in flight; the two batches in flight, the replayed graph and the reduction inside the search launch are what "
                "the target needs, and 200-step runs of it read 10–15 % slow (the clocks settle over thousands of such steps: these lines time 2 000).")
        out += ["", txt]
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
