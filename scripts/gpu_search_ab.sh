#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
export LD_LIBRARY_PATH=/opt/rocm/lib:${LD_LIBRARY_PATH:-}
echo "== search parity (fast path default)"
timeout 1500 python -m pytest tests/test_gpu_search.py -m gpu -q --timeout 900 -p no:cacheprovider -x > gpurun_out/search_fast.log 2>&1; echo "rc=$?"; tail -12 gpurun_out/search_fast.log | cut -c1-300
ls gpurun_out/diag_*.json 2>/dev/null | head
echo "== probe: fast vs exact"
for nq in 4096 32768; do
  tools/abi_probe search 8841823 $nq 200 2
  ANCE_SEARCH=exact tools/abi_probe search 8841823 $nq 200 1
done
tools/abi_probe search 8841823 4096 100 2
echo "== bench search leg (fast) and exact"
timeout 600 python bench.py --skip-encode --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/bench_search_fast.json 2>gpurun_out/bench_search_fast.err; python -c "
import json; d=json.load(open('gpurun_out/bench_search_fast.json')); s=d['search']; print('fast qps', s['value'], s['roofline']['achieved'], d.get('errors'))"
timeout 600 python bench.py --skip-encode --no-cpu-baseline --steps 3 --warmup 1 --query-block 32768 > gpurun_out/bench_search_fast32k.json 2>gpurun_out/bench_search_fast32k.err; python -c "
import json; d=json.load(open('gpurun_out/bench_search_fast32k.json')); s=d['search']; print('fast 32k qps', s['value'], s['roofline']['achieved'], d.get('errors'))"
