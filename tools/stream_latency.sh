#!/bin/bash
# Per-call latency of aof_stream_push_host (what calcFlow() costs), both facade configurations.
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
g++ -O2 -Iinclude tools/bench_stream.cpp -Laero-optical-flow_amd/csrc -laof -Wl,-rpath,$PWD/aero-optical-flow_amd/csrc -Wl,-rpath,/opt/rocm/lib -L/opt/rocm/lib -lamdhip64 -o gpurun_out/bench_stream
for cfg in "64 64 1" "128 128 1" "64 64 2" "128 128 2"; do
  timeout -k 10 60 ./gpurun_out/bench_stream $cfg
done
