#!/bin/bash
# Builds of libaof.so (ab/*.so) against noise levels, default search mode, interleaved, one box
#   tools/ab_noise.sh <workload> <out-dir> "<noise levels>" <lib> [<lib> ...]
wl=$1; out=$2; nzs=$3; shift 3
mkdir -p $out
for round in 1 2; do
  for nz in $nzs; do
    for lib in "$@"; do
      tag=$(basename $lib .so)
      AOF_LIB=$PWD/$lib timeout -k 10 200 python bench.py --workload $wl --noise $nz --traffic file --cpu-seconds 0 --steps 100 --warmup 20 > $out/${wl}_${tag}_${round}_n$nz.json 2> $out/err.txt || { echo "$lib failed"; tail -3 $out/err.txt; exit 1; }
    done
  done
done
python - $out $wl <<'PY'
import json, glob, sys, os
rows = {}
for f in sorted(glob.glob(f"{sys.argv[1]}/{sys.argv[2]}_*.json")):
    j = json.loads(open(f).read().strip().splitlines()[-1])
    name = os.path.basename(f)[len(sys.argv[2]) + 1:-5]
    tag, nz = name.rsplit("_n", 1)
    rows.setdefault(tag, {})[int(nz)] = (j["value"] / 1e6, j["roofline"]["kernel_ms"] * 1e3)
nzs = sorted({n for r in rows.values() for n in r})
print("M pairs/s (K2 us)".ljust(22) + "".join(f"n{n:<16d}" for n in nzs))
for tag, r in rows.items():
    print(tag.ljust(22) + "".join(f"{r[n][0]:6.3f} ({r[n][1]:6.1f})  " if n in r else " " * 17 for n in nzs))
PY
