#!/bin/bash
# One box's lines for the box-to-box spread table (profiles/rNN_box_spread.txt): run through gpurun
# several times -- every call gets a fresh box -- and paste gpurun_out/box_spread.txt.
set -o pipefail
O=gpurun_out/box_spread.txt
run() {  # label, bench arguments
    local label=$1; shift
    timeout -k 10 200 python3 bench.py --cpu-seconds 0 --traffic file "$@" 2>/dev/null | python3 tools/box_spread_line.py "$label" >> $O || { echo "$label failed" >> $O; return 1; }
}
echo "# box $(hostname) $(date -u +%H:%M:%S)" >> $O
run c2 && run c3 --workload c3 && run "c2 128 pairs (share)" --pairs 128 && run "c2 1024, two batches" --streams 2 --reduce separate
cat $O
