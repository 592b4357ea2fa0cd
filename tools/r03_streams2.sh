#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/streams2
rm -rf $O; mkdir -p $O
cd $R
run() { tag=$1; shift; timeout -k 10 200 python3 bench.py --cpu-seconds 0 --steps 200 "$@" > $O/$tag.json 2> $O/$tag.err || { echo "$tag failed"; tail -5 $O/$tag.err; exit 1; }; }
for n in 64 128 256 512 1024; do
  for s in 1 2 3; do
    for r in separate fused; do
      run p${n}_s${s}_${r} --pairs $n --streams $s --reduce $r
    done
  done
done
run p1024_s2_fused_graph --streams 2 --reduce fused --graph on
run p1024_s2_sep_graph --streams 2 --graph on
run p128_s2_fused_eager --pairs 128 --streams 2 --reduce fused --graph off
run p128_s4_fused --pairs 128 --streams 4 --reduce fused
python3 - $O <<'PY'
import json, glob, sys
rows=[]
for f in glob.glob(sys.argv[1] + "/*.json"):
    j = json.loads(open(f).read().strip().splitlines()[-1])
    rows.append((j['config']['pairs_per_gpu'], f.split('/')[-1], j))
for n,name,j in sorted(rows, key=lambda r:(r[0],r[1])):
    print(f"{name:28s} {j['value']:>10.0f} pairs/s  step {j['ms_per_step']*1e3:8.2f} us  median {j['ms_per_step_median']*1e3:8.2f}  graph {j['config']['graph_replay']}  K2 {j['roofline']['kernel_ms']*1e3:7.2f}  parity {j['parity']['oracle_pairs_bit_exact']} {j['parity']['all_pairs_return_known_shift']}")
PY
