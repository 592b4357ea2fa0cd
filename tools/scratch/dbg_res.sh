#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
g++ -O2 -Iinclude tools/bench_stream.cpp -Laero-optical-flow_amd/csrc -laof -Wl,-rpath,$PWD/aero-optical-flow_amd/csrc -Wl,-rpath,/opt/rocm/lib -L/opt/rocm/lib -lamdhip64 -o gpurun_out/bench_stream
for i in 1 2 3 4 5 6; do
timeout -k 5 20 ./gpurun_out/bench_stream 64 64 1 4; echo "rc=$?"
done
timeout -k 5 20 ./gpurun_out/bench_stream 128 128 2 4; echo "rc=$?"
