#!/bin/bash
# round 5: k_tile16_probe and k_search_tile16 durations, head against new, c5 noise-free and +-40 LSB (kernel trace)
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof46
mkdir -p $O
cd $R
for lib in head new; do
  if [ $lib = head ]; then export AOF_LIB=$R/ab/libaof_head.so; else unset AOF_LIB; fi
  for nz in 0 40; do
    timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${lib}_n$nz -- python3 $R/bench.py --workload c5 --pairs 256 --noise $nz --cpu-seconds 0 --traffic file --legs none > $O/${lib}_n$nz.log 2>&1 < /dev/null || { echo "trace failed"; tail -3 $O/${lib}_n$nz.log; exit 1; }
    f=$(find $O/${lib}_n$nz -name "*kernel_stats.csv" | head -1)
    echo "== $lib noise $nz"
    if [ -n "$f" ]; then grep -E "tile16" "$f" | awk -F, '{print $1, $2, $4}' | cut -c1-140; fi
  done
done
echo done
