#!/bin/bash
# Runs on the GPU box through gpurun: smoke, GPU parity tests (one pytest process per file so a
# device fault in one file cannot take the others down), a short bench, logs under gpurun_out/.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx" | head -4 > gpurun_out/device.txt
nproc >> gpurun_out/device.txt
echo "== smoke" ; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1 ; echo "smoke rc=$?" ; tail -3 gpurun_out/smoke.log
for f in test_gpu_search test_gpu_gemm test_gpu_encoder test_gpu_e2e test_gpu_dpr; do
  echo "== $f"
  timeout 1500 python -m pytest tests/$f.py -m gpu -q --timeout 900 -p no:cacheprovider > gpurun_out/$f.log 2>&1
  echo "$f rc=$?"; tail -15 gpurun_out/$f.log
done
if [ "${1:-}" != "nobench" ]; then
  echo "== bench"
  timeout 900 python bench.py --steps 3 --warmup 1 ${BENCH_ARGS:-} > gpurun_out/bench.log 2> gpurun_out/bench.err
  echo "bench rc=$?"; tail -c 3000 gpurun_out/bench.log; tail -5 gpurun_out/bench.err
fi
