#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
export LD_LIBRARY_PATH=/opt/rocm/lib:${LD_LIBRARY_PATH:-}
for r in 1 2; do timeout 600 python -m pytest tests/test_gpu_gemm.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2; done
timeout 1500 python -m pytest tests/test_gpu_search.py -m gpu -q -x -p no:cacheprovider --timeout 900 2>&1 | tail -4
for S in 2 4; do ANCE_FAST_SPLITS=$S tools/abi_probe search 8841823 32768 200 2 | tail -1; done
tools/abi_probe search 8841823 4096 200 2 | tail -1
for shape in "1 65536 3072 768" "2 65536 768 3072" "2 65536 768 768" "0 65536 1536 768" "0 8192 8192 8192"; do
  tools/abi_probe gemm 0 $shape 20 | tail -1
  tools/abi_probe gemm 8 $shape 20 | tail -1
done
timeout 900 python -m pytest tests/test_gpu_encoder.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2
tools/abi_probe encode 65536 128 12 3 | tail -1
