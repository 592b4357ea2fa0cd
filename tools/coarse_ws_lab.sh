#!/bin/bash
# Lab: the wave-specialised form of the fused coarse kernel (tools/coarse_ws_lab.hpp) against the shipped k_coarse.
#   tools/coarse_ws_lab.sh build     (here: cross-compiles lab libraries ab/ws*.so, one per role split)
#   tools/coarse_ws_lab.sh run <out> (on the GPU box, through gpurun: C3 with every library, two interleaved rounds)
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  mkdir -p ab; cd aero-optical-flow_amd/csrc && make -j8 >/dev/null || exit 1
  for cfg in "256 26 1 20 6" "384 30 1 23 5" "512 40 2 34 4"; do set -- $cfg; tag="ws$1_$2_$3_$4_$5"
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -I../../include -I. --offload-arch=gfx950 -fno-fast-math \
      -DAOF_LAB_COARSE_WS='"../../tools/coarse_ws_lab.hpp"' -DAOF_WS_SEARCH=$1 -DAOF_WS_NK=$2 -DAOF_WS_S1=$3 -DAOF_WS_S2=$4 -DAOF_WS_DEPTH=$5 \
      -c k_coarse.hip -o /tmp/kc_$tag.o || exit 1
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../ab/$tag.so $(ls *.o | grep -v k_coarse.o) /tmp/kc_$tag.o -Wl,-rpath,/opt/rocm/lib || exit 1
  done
  exit 0
fi
out=$2; mkdir -p $(dirname $out); : > $out
line() { python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', j['value'], j['ms_per_step'], j['kernels_ms'], j['parity']['oracle_pairs_bit_exact'])"; }
for round in 1 2; do
  timeout -k 10 120 python3 bench.py --workload c3 --cpu-seconds 0 --traffic file --steps 100 2>/dev/null | line shipped >> $out
  for lib in ab/ws*.so; do
    AOF_LIB=$PWD/$lib timeout -k 10 120 python3 bench.py --workload c3 --cpu-seconds 0 --traffic file --steps 100 2>/dev/null | line $(basename $lib .so) >> $out
  done
done
cat $out
