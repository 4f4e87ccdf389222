#!/bin/bash
# micro-batch size vs Infinity Cache: a layer's working set (residual pair + FFN1 activations) fits 256 MB below ~21,760 tokens,
# and 85 row tiles x 3 column tiles = 255 workgroups = one full round of the N = 768 GEMMs
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for tok in 65536 21760 43520 21504 16384 32768; do for lanes in 2 1 3; do
  ANCE_ENCODER_STREAMS=$lanes timeout 300 python scripts/encode_mode_leg.py fp16 6 16384 $tok 2>&1 | tail -1
done; done | tee gpurun_out/mall_sweep.txt
