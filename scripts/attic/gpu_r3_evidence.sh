#!/bin/bash
# Round-3 evidence on the final tree (copy with scripts/collect_profiles.sh r03): default bench line, rocprofv3 kernel
# traces of the bench command and of both legs alone (encode single-stream), PMC passes restricted to the roofline
# kernels, one full refresh end to end, the other BASELINE configurations.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== bench (default flags)"
timeout 900 python bench.py > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "rc=$?"; tail -c 600 gpurun_out/bench.log; tail -3 gpurun_out/bench.err
echo "== rocprofv3 kernel trace of the bench command"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench -o kt -- python bench.py --no-cpu-baseline > gpurun_out/prof_bench.log 2>&1; echo "rc=$?"
find gpurun_out/prof_bench -name "*kernel_stats.csv" | head -2
find gpurun_out -name "*kernel_trace.csv" -size +8M -delete
bash scripts/gpu_pmc.sh
echo "== full refresh"
timeout 1500 python bench.py --full > gpurun_out/bench_full.log 2> gpurun_out/bench_full.err; echo "rc=$?"; tail -c 1200 gpurun_out/bench_full.log
rm -rf /tmp/ance_full
echo "== other BASELINE configurations"
timeout 900 python scripts/bench_configs.py > gpurun_out/bench_configs.jsonl 2> gpurun_out/bench_configs.err; echo "rc=$?"; cat gpurun_out/bench_configs.jsonl | cut -c1-400
