#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (gpurun_out/pmc/<PASS>/.../*_counter_collection.csv)
for the aof kernels: per-kernel mean counter values, and the HBM traffic per launch
corrected as /opt/skills/guides/MI355X_MICROARCH.md section HBM prescribes for gfx950:
FETCH_SIZE (KB) reads exactly half of a wide coalesced 16-B/lane stream -> x2;
WRITE_SIZE (KB) is exact for 16-B/lane stores.  Writes profiles/pmc_traffic.json
entries keyed "<workload>:<pairs>" that bench.py reports as roofline.traffic."""
import collections
import csv
import glob
import json
import os
import sys


def summarise(root, workload, pairs, out_txt=None):
    """Returns the pmc_traffic.json entry of the passes under `root` (and writes the text summary / merges the entry into
    pmc_traffic.json beside out_txt when that is given)."""
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(root, "*", "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "aof::" not in k:
                continue
            k = k.split("(anonymous namespace)::", 1)[-1].split("(")[0]   # the kernel's own name, not a parameter type's
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    lines = [f"# rocprofv3 --pmc summary, workload {workload}, {pairs} pairs per launch (means over launches)"]
    traffic = {}
    for k in sorted(acc):
        lines.append(k)
        for c in sorted(acc[k]):
            v = acc[k][c]
            lines.append(f"    {c:<24} n={len(v):<3} mean={sum(v)/len(v):.6g}")
        if "FETCH_SIZE" in acc[k] and "WRITE_SIZE" in acc[k]:
            # one kernel name can serve both pyramid levels (c3): price the level-0 launches,
            # i.e. the upper cluster when the per-launch values fall into two groups
            def level0(v):
                lo, hi = min(v), max(v)
                if hi > 1.5 * lo:
                    v = [x for x in v if x > 0.5 * (lo + hi)]
                return sum(v) / len(v)
            rd = 2.0 * 1024.0 * level0(acc[k]["FETCH_SIZE"])
            wr = 1024.0 * level0(acc[k]["WRITE_SIZE"])
            traffic[k] = (rd, wr)
            lines.append(f"    => HBM read {rd/1e6:.1f} MB (FETCH_SIZE x2, gfx950 correction) + write {wr/1e6:.1f} MB per launch")
    if out_txt:
        open(out_txt, "w").write("\n".join(lines) + "\n")
        print("\n".join(lines))
    # the level-0 search is the k_search kernel that moves the most bytes (the profiled commands name their search mode,
    # so only the headline mode's kernels run: bench.py's default line would also time the other modes)
    names = [k for k in traffic if k.startswith("k_search") or k.startswith("k_flow")]
    if not names:   # workloads without a search kernel (ingest, derotate): the kernel that moves the most bytes
        names = list(traffic)
    search = sorted(names, key=lambda k: -sum(traffic[k]))
    if not search:
        return None
    rd, wr = traffic[search[0]]
    # every kernel of one step; a kernel name that serves two launches per step (split coarse path) counts once, at its
    # larger launch
    step = dict(traffic)
    entry = {"kernel": search[0], "hbm_bytes_per_launch": int(rd + wr), "read_bytes": int(rd), "write_bytes": int(wr),
             "step_bytes": int(sum(a + b for a, b in step.values())), "step_kernels": sorted(step),
             "source": os.path.basename(out_txt) if out_txt else "live"}
    if out_txt:
        path = os.path.join(os.path.dirname(os.path.abspath(out_txt)), "pmc_traffic.json")
        data = json.load(open(path)) if os.path.exists(path) else {}
        data[f"{workload}:{pairs}"] = entry
        json.dump(data, open(path, "w"), indent=1, sort_keys=True)
    return entry


if __name__ == "__main__":
    summarise(sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4])
