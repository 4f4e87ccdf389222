#!/bin/bash
# Timing-only sweep of LDS-DMA placement variants (ablate 32+VAR) in one process sequence.
set -u
cd "$(dirname "$0")/.."
export LD_LIBRARY_PATH=/opt/rocm/lib:${LD_LIBRARY_PATH:-}
for rep in 1 2; do
for shape in "0 8192 8192 8192" "0 65536 1536 768"; do
  for v in 32 33 34 35 36 37 38 39; do
    tools/abi_probe gemm $v $shape 10 | tail -1
  done
done
done
