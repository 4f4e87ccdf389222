#!/bin/bash
# Ping-pong GEMM: parity, ablations, PMC.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
export LD_LIBRARY_PATH=/opt/rocm/lib:${LD_LIBRARY_PATH:-}
for r in 1 2 3; do
  timeout 600 python -m pytest tests/test_gpu_gemm.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2
done
echo "== ablations: 0 product, 17 L2-hot K, 21 L2-hot K + tile00, 24 no staging, 18 no MFMA, 26 neither"
for shape in "1 65536 3072 768" "2 65536 768 3072" "0 8192 8192 8192"; do
  for dm in 0 17 21 24 18 26; do
    tools/abi_probe gemm $dm $shape 10 | tail -1
  done
done
SHAPE="0 8192 8192 8192" TAG=pp8k bash scripts/gpu_gemm_pmc.sh 2>&1 | tail -40
SHAPE="1 65536 3072 768" TAG=ppffn1 bash scripts/gpu_gemm_pmc.sh 2>&1 | tail -40
