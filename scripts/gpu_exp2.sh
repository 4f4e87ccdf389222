#!/bin/bash
# Ping-pong GEMM main loop: parity (repeated for race screening) and A/B against the two-phase loop.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
export LD_LIBRARY_PATH=/opt/rocm/lib:${LD_LIBRARY_PATH:-}
for r in 1 2 3; do
  timeout 600 python -m pytest tests/test_gpu_gemm.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
done
echo "== A/B (ablate 0 = ping-pong, 8 = two-phase)"
for a in 0 8 0 8; do
  tools/abi_probe gemm $a 1 65536 3072 768 30 | tail -1
  tools/abi_probe gemm $a 2 65536 768 3072 30 | tail -1
  tools/abi_probe gemm $a 0 65536 1536 768 30 | tail -1
  tools/abi_probe gemm $a 0 8192 8192 8192 10 | tail -1
done
echo "== encoder parity + probe"
timeout 900 python -m pytest tests/test_gpu_encoder.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
tools/abi_probe encode 65536 128 12 3 | tail -1
