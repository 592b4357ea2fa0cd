#!/bin/bash
# The default (adaptive) search of a workload against +-N LSB of noise on the newer frame, with the other two modes of the
# same batch from the same bench line (one box).   tools/search_modes_vs_noise.sh <out-dir> <workload> [<workload> ...]
out=$1; shift
mkdir -p $out
for wl in "$@"; do
  for nz in ${NZS:-0 2 4 8 16 40}; do
    timeout -k 10 300 python3 ${GRAFT_REPO_ROOT:-.}/bench.py --workload $wl --noise $nz --traffic file --cpu-seconds 0 > $out/${wl}_n$nz.json 2> $out/${wl}_n$nz.err || { tail -3 $out/${wl}_n$nz.err; exit 1; }
  done
done
python - $out "$@" <<'PY'
import json, sys
out, wls = sys.argv[1], sys.argv[2:]
for wl in wls:
    print(wl, "M pairs/s: default | exhaustive | always-pruned   (K2 us default)   adaptive stats")
    for nz in (0, 2, 4, 8, 12, 16, 40):
        try:
            d = json.loads(open(f"{out}/{wl}_n{nz}.json").read().strip().splitlines()[-1])
        except Exception:
            continue
        g = lambda k: d.get(k, {}).get("per_gpu_value", 0) / 1e6
        print(f"  n{nz:<3d} {d['value'] / 1e6:7.3f} | {g('exhaustive_search'):7.3f} | {g('exact_pruned_search'):7.3f}   ({d['roofline']['kernel_ms'] * 1e3:6.1f})  "
              f"{d['config'].get('adaptive_search')}  identical={[d[k]['records_identical_to_headline'] for k in d if k.endswith('_search')]}")
PY
