#!/bin/bash
# Lab: 8x8 search modes against noise, several builds of the pruned lane8 kernel (ab/*.so), one box.
#   tools/p8_noise_sweep.sh <workload> <out-dir> <lib> [<lib> ...]
wl=$1; out=$2; shift 2
mkdir -p $out
run() {  # lib mode noise
    tag=$(basename $1 .so)
    AOF_LIB=$PWD/$1 timeout -k 10 200 python bench.py --workload $wl --search $2 --noise $3 --traffic file --cpu-seconds 0 --steps 100 --warmup 20 \
        > $out/${wl}_${tag}_$2_n$3.json 2> $out/${wl}_${tag}_$2_n$3.err || { echo "$1 $2 $3 failed"; tail -3 $out/${wl}_${tag}_$2_n$3.err; exit 1; }
}
for nz in ${NZS:-0 4 8 12 16 40}; do
    run $1 exhaustive $nz
    for lib in "$@"; do
        run $lib adaptive $nz
    done
    run $1 pruned $nz
done
python - $out $wl <<'PY'
import json, glob, sys, os
rows = {}
for f in sorted(glob.glob(f"{sys.argv[1]}/{sys.argv[2]}_*.json")):
    j = json.loads(open(f).read().strip().splitlines()[-1])
    name = os.path.basename(f)[len(sys.argv[2]) + 1:-5]
    tag, nz = name.rsplit("_n", 1)
    rows.setdefault(tag, {})[int(nz)] = j["value"] / 1e6
nzs = sorted({n for r in rows.values() for n in r})
print("M pairs/s".ljust(28) + "".join(f"n{n:<7d}" for n in nzs))
for tag, r in rows.items():
    print(tag.ljust(28) + "".join(f"{r.get(n, 0):<8.3f}" for n in nzs))
PY
