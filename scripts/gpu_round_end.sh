#!/bin/bash
# What the driver runs at round end, plus the profiles we commit: GPU tests, smoke, bench (default flags), rocprofv3
# kernel trace of the bench command, the PMC passes of both legs (scripts/gpu_pmc.sh), the other BASELINE configs.
# Copy the summaries you want judged from gpurun_out/ to profiles/ (scripts/collect_profiles.sh rNN).
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
bash scripts/gpu_check.sh nobench
echo "== bench (default flags)"
timeout 900 python bench.py > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "rc=$?"; tail -c 1500 gpurun_out/bench.log; tail -3 gpurun_out/bench.err
echo "== rocprofv3 kernel trace of the bench command"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench -o kt -- python bench.py --no-cpu-baseline > gpurun_out/prof_bench.log 2>&1; echo "rc=$?"
find gpurun_out/prof_bench -name "*kernel_stats.csv" | head -2
find gpurun_out -name "*kernel_trace.csv" -size +8M -delete
bash scripts/gpu_pmc.sh
echo "== other BASELINE configurations"
timeout 900 python scripts/bench_configs.py > gpurun_out/bench_configs.jsonl 2> gpurun_out/bench_configs.err; echo "rc=$?"; cat gpurun_out/bench_configs.jsonl
