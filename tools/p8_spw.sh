#!/bin/bash
# Lab: chunks per workgroup of the pruned lane8 kernel against the batch size (ab/env.so reads AOF_SPW)
out=gpurun_out/p8e; mkdir -p $out
for pairs in ${PAIRS:-1024 512 256 128}; do
  AOF_LIB=$PWD/ab/env.so python bench.py --workload c2 --pairs $pairs --search exhaustive --streams 1 --graph off --traffic file --cpu-seconds 0 --steps 100 --warmup 20 > $out/p${pairs}_exh.json 2>/dev/null
  for spw in ${SPWS:-1 2 3 4 6 8 12 16}; do
    AOF_SPW=$spw AOF_LIB=$PWD/ab/env.so python bench.py --workload c2 --pairs $pairs --search pruned --streams 1 --graph off --traffic file --cpu-seconds 0 --steps 100 --warmup 20 > $out/p${pairs}_spw$spw.json 2>$out/err.txt || { tail -3 $out/err.txt; exit 1; }
  done
done
python - <<'PY'
import json, glob, os
for pairs in map(int, os.environ.get("PAIRS", "1024 512 256 128").split()):
    row = []
    for tag in ["exh"] + [f"spw{s}" for s in map(int, os.environ.get("SPWS", "1 2 3 4 6 8 12 16").split())]:
        j = json.loads(open(f"gpurun_out/p8e/p{pairs}_{tag}.json").read().strip().splitlines()[-1])
        row.append(f"{tag} {j['value'] / 1e6:.3f}")
    print(pairs, "  ".join(row))
PY
