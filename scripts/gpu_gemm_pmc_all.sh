#!/bin/bash
cd "$(dirname "$0")/.."
export LD_LIBRARY_PATH=/opt/rocm/lib:${LD_LIBRARY_PATH:-}
for cfg in "qk|0 65536 1536 768" "ffn1|1 65536 3072 768" "ffn2|2 65536 768 3072" "out|2 65536 768 768"; do
  tag=${cfg%%|*}; shape=${cfg#*|}
  echo "=== $tag ($shape)"; tools/abi_probe gemm 0 $shape 10
  TAG=$tag SHAPE="$shape" bash scripts/gpu_gemm_pmc.sh 2>&1 | grep -E "derived|GRBM_GUI|SQ_INSTS_LDS |TCC_"
done
