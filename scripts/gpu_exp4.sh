#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
export LD_LIBRARY_PATH=/opt/rocm/lib:${LD_LIBRARY_PATH:-}
for r in 1 2 3; do
  timeout 600 python -m pytest tests/test_gpu_gemm.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2
done
echo "== 0 product, 8 two-phase, 21 L2-hot K + tile00, 24 no staging, 18 no MFMA, 26 neither"
for shape in "1 65536 3072 768" "2 65536 768 3072" "0 65536 1536 768" "0 8192 8192 8192"; do
  for dm in 0 8 21 24 18 26; do
    tools/abi_probe gemm $dm $shape 10 | tail -1
  done
done
timeout 900 python -m pytest tests/test_gpu_encoder.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2
tools/abi_probe encode 65536 128 12 3 | tail -1
