#!/bin/bash
set -u
cd "$(dirname "$0")/.."
export LD_LIBRARY_PATH=/opt/rocm/lib:${LD_LIBRARY_PATH:-}
for S in 2 4 8 16 2 8; do
  echo "splits $S"; ANCE_FAST_SPLITS=$S tools/abi_probe search 8841823 32768 200 2 | tail -1
done
