set -e
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "c3 or fused or many_pairs or two_level" 2>&1 | tail -2
bash tools/ab_bench.sh c3 gpurun_out/ab_fin ab/base.so ab/fin.so | awk '{print $1, $2, $5, $8, $9, $10, $11}'
