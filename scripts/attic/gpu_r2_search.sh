#!/bin/bash
# round 2, search: parity tests of the reworked two-precision path, then the scheduling sweep on the headline shape
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== test_gpu_search"
timeout 1500 python -m pytest tests/test_gpu_search.py -m gpu -q --timeout 900 -p no:cacheprovider -x > gpurun_out/test_gpu_search.log 2>&1
echo "rc=$?"; tail -25 gpurun_out/test_gpu_search.log
if [ "${1:-}" != "nosweep" ]; then
  echo "== sweep"
  timeout 900 python scripts/sweep_search.py ${SWEEP_ARGS:-} > gpurun_out/sweep_search.jsonl 2> gpurun_out/sweep_search.err
  echo "rc=$?"; cat gpurun_out/sweep_search.jsonl; tail -5 gpurun_out/sweep_search.err
fi
