#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/small_step3
rm -rf $O; mkdir -p $O
cd $R
run() {  # tag, lib, args...
  tag=$1; lib=$2; shift 2
  AOF_LIB=$R/ab/$lib.so timeout -k 10 200 python3 bench.py --cpu-seconds 0 --steps 200 "$@" > $O/$tag.json 2> $O/$tag.err || { echo "$tag failed"; tail -5 $O/$tag.err; exit 1; }
}
for rep in 1 2; do
  run p128_t64_fused_graph_$rep t64 --pairs 128 --graph
  run p128_t64_sep_graph_$rep t64 --pairs 128 --graph --reduce separate
  run p128_noat_fused_graph_$rep noatomics --pairs 128 --graph
  run p1024_t64_fused_$rep t64
  run p1024_noat_fused_$rep noatomics
  run p1024_t64_sep_$rep t64 --reduce separate
  run p256_t64_fused_graph_$rep t64 --pairs 256 --graph
  run p256_t64_sep_graph_$rep t64 --pairs 256 --graph --reduce separate
  run p512_t64_sep_graph_$rep t64 --pairs 512 --graph --reduce separate
done
python3 - $O <<'PY'
import json, glob, sys
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    j = json.loads(open(f).read().strip().splitlines()[-1])
    print(f"{f.split('/')[-1]:36s} {j['value']:>10.0f} pairs/s  step {j['ms_per_step']*1e3:8.2f} us  K2 {j['roofline']['kernel_ms']*1e3:7.2f} us  {j['kernels_ms']} parity {j['parity']['oracle_pairs_bit_exact']} {j['parity']['all_pairs_return_known_shift']}")
PY
cd /tmp && export TMPDIR=/tmp
for mode in auto separate; do
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$mode -- python3 $R/bench.py --pairs 128 --steps 200 --cpu-seconds 0 --graph --reduce $mode > $O/kt_$mode.log 2>&1 || { echo "kernel trace failed"; exit 1; }
python3 $R/tools/summarize_rocprof.py $(ls $O/kt_$mode/*/*kernel_stats.csv | head -1) "bench.py --pairs 128 --steps 200 --cpu-seconds 0 --graph --reduce $mode" | grep -v "at::native\|Memset\|elementwise\|Cijk\|rocprim\|vectorized"
rm -rf $O/kt_$mode
done
