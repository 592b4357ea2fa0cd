#!/bin/bash
# Same-box A/B of the per-call latency for two builds of libaof.so: ab/<name>/libaof.so ...
mkdir -p gpurun_out
for n in "$@"; do g++ -O2 -Iinclude tools/bench_stream.cpp -Lab/$n -laof -Wl,-rpath,$PWD/ab/$n -Wl,-rpath,/opt/rocm/lib -L/opt/rocm/lib -lamdhip64 -o gpurun_out/bench_stream_$n || exit 1; done
for round in 1 2 3; do
  for cfg in "64 64 1" "128 128 2"; do
    for mode in 0 4; do
      for n in "$@"; do echo -n "$n: "; timeout -k 10 60 ./gpurun_out/bench_stream_$n $cfg $mode 20000 | cut -d: -f1,2 | sed 's/ lane8 graph=1//; s/instantiated=. //; s/on_device=.//'; done
    done
  done
done
