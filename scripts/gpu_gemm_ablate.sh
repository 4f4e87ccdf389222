#!/bin/bash
# Ablation of the encoder GEMM on the GPU box (ance_debug_gemm ablate bits: 1 no loads, 2 no MFMA,
# 4 L2-resident operands) for the encoder's shapes.  Results go to stdout as JSON lines.
# The ablation builds of the kernel are in the measurement library only: build tools/abi_probe with -lance_amd_measure first
# (hipcc -O2 -std=c++17 tools/abi_probe.cpp -Iinclude -Lance_amd -lance_amd_measure -Wl,-rpath,'$ORIGIN/../ance_amd' -o tools/abi_probe).
cd "$(dirname "$0")/.."
export LD_LIBRARY_PATH=/opt/rocm/lib:${LD_LIBRARY_PATH:-}
for shape in "0 65536 1536 768" "1 65536 3072 768" "2 65536 768 3072" "2 65536 768 768" "0 8192 8192 8192"; do
  for dm in 0 1 2 3 4 6; do
    tools/abi_probe gemm $dm $shape 10
  done
done
