#!/bin/bash
# Runs on the GPU box through gpurun: smoke, every GPU parity test file (one pytest process per file so a
# device fault in one file cannot take the others down), optionally a short bench; logs under gpurun_out/.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx" | head -4 > gpurun_out/device.txt
nproc >> gpurun_out/device.txt
echo "== smoke" ; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1 ; echo "smoke rc=$?" ; tail -3 gpurun_out/smoke.log
bash scripts/gpu_tests.sh
if [ "${1:-}" != "nobench" ]; then
  echo "== bench"
  timeout 900 python bench.py --steps 3 --warmup 1 ${BENCH_ARGS:-} > gpurun_out/bench.log 2> gpurun_out/bench.err
  echo "bench rc=$?"; tail -c 3000 gpurun_out/bench.log; tail -5 gpurun_out/bench.err
fi
