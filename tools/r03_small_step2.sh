#!/bin/bash
# The 128-pair step (configs[3]'s per-GPU share) with the reduction inside the search launch against the
# separate K3, and one-wave against four-wave workgroups (ab/t64|t256.so), same box, interleaved.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/small_step2
rm -rf $O; mkdir -p $O
cd $R
run() {  # tag, lib, args...
  tag=$1; lib=$2; shift 2
  AOF_LIB=$R/ab/$lib.so timeout -k 10 200 python3 bench.py --cpu-seconds 0 --steps 200 "$@" > $O/$tag.json 2> $O/$tag.err || { echo "$tag failed"; tail -5 $O/$tag.err; exit 1; }
}
for rep in 1 2; do
  for lib in t64 t256; do
    run p128_${lib}_fused_graph_$rep $lib --pairs 128 --graph
    run p128_${lib}_sep_graph_$rep $lib --pairs 128 --graph --reduce separate
    run p1024_${lib}_fused_$rep $lib
  done
  run p128_t64_fused_eager_$rep t64 --pairs 128
  run p1024_t64_sep_$rep t64 --reduce separate
  run p256_t64_fused_graph_$rep t64 --pairs 256 --graph
  run p512_t64_fused_graph_$rep t64 --pairs 512 --graph
done
python3 - $O <<'PY'
import json, glob, sys
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    j = json.loads(open(f).read().strip().splitlines()[-1])
    print(f"{f.split('/')[-1]:36s} {j['value']:>10.0f} pairs/s  step {j['ms_per_step']*1e3:8.2f} us  K2 {j['roofline']['kernel_ms']*1e3:7.2f} us  {j['kernels_ms']} parity {j['parity']['oracle_pairs_bit_exact']} {j['parity']['all_pairs_return_known_shift']}")
PY
