#!/bin/bash
set -u
cd "$(dirname "$0")/.."
export LD_LIBRARY_PATH=/opt/rocm/lib:${LD_LIBRARY_PATH:-}
for n in 131072 262144 1048576 4194304 8841823; do
  tools/abi_probe search $n 32768 200 3 | tail -1
done
