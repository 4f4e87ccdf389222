#!/bin/bash
# Kernel-trace stats (no counters) of the final tree: the default bench command, and the encode leg single-stream.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/final
export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/bench -o kt -- python bench.py --no-cpu-baseline > gpurun_out/final/bench.log 2>&1; echo "rc=$?"
tail -1 gpurun_out/final/bench.log | cut -c1-300
ANCE_ENCODER_STREAMS=1 timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/encode1 -o kt -- python bench.py --skip-search --no-cpu-baseline --steps 2 --warmup 1 > gpurun_out/final/encode1.log 2>&1; echo "rc=$?"
find gpurun_out/final -name "*kernel_trace.csv" -delete
find gpurun_out/final -name "*kernel_stats.csv" | head
