#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
export LD_LIBRARY_PATH=/opt/rocm/lib:${LD_LIBRARY_PATH:-}
timeout 600 python -m pytest tests/test_gpu_gemm.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2
for shape in "1 65536 3072 768" "2 65536 768 3072" "2 65536 768 768" "0 65536 1536 768"; do
  tools/abi_probe gemm 0 $shape 20 | tail -1
done
timeout 900 python -m pytest tests/test_gpu_encoder.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2
cat gpurun_out/encoder_parity.jsonl 2>/dev/null | tail -4
tools/abi_probe encode 65536 128 12 3 | tail -1
