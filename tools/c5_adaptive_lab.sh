#!/bin/bash
# Same-box sweep of the lab builds of the adaptive 16x16 thresholds (ab/t16_*.so): tools/c5_adaptive_lab.sh <out-file> [sweep args]
out=$1; shift
mkdir -p $(dirname $out); : > $out
for lib in ab/t16_*.so; do
  AOF_LIB=$PWD/$lib timeout -k 10 280 python3 tools/c5_adaptive_sweep.py "$@" >> $out 2>> $out.err || { echo "$lib failed"; tail -5 $out.err; exit 1; }
  echo >> $out
done
cat $out
