#!/bin/bash
# Same-box A/B of several builds of libaof.so (ab/*.so, see AOF_LIB in aero-optical-flow_amd/__init__.py):
#   tools/ab_bench.sh <workload> <out-prefix> <lib> [<lib> ...]      (run through gpurun)
# Interleaved, two rounds, so clock drift and box-to-box differences cancel.
wl=$1; out=$2; shift 2
mkdir -p $(dirname $out)
for round in 1 2; do
    for lib in "$@"; do
        tag=$(basename $lib .so)
        AOF_LIB=$PWD/$lib timeout -k 10 200 python bench.py --workload $wl --cpu-seconds 0 --traffic file --steps 100 > ${out}_${tag}_$round.json 2> ${out}_${tag}_$round.err || { echo "$lib failed"; tail -3 ${out}_${tag}_$round.err; exit 1; }
    done
done
python - "$out" <<'PY'
import json, glob, sys
for f in sorted(glob.glob(sys.argv[1] + "_*.json")):
    j = json.loads(open(f).read().strip().splitlines()[-1])
    print(f"{f:60s} {j['value']:>10.0f} pairs/s  step {j['ms_per_step']:.4f} ms  K2 {j['roofline']['kernel_ms']:.4f}  {j['kernels_ms']}")
PY
