#!/bin/bash
set -u
cd "$(dirname "$0")/.."
export LD_LIBRARY_PATH=/opt/rocm/lib:${LD_LIBRARY_PATH:-}
for ns in 1 2 3 2 3; do
  echo "lanes=$ns"
  ANCE_ENCODER_STREAMS=$ns tools/abi_probe encode 131072 128 12 3 | tail -1 | cut -c1-120
done
