#!/bin/bash
# Round-4 evidence on the final tree (copy with scripts/collect_profiles.sh r04): default bench line (three encoder modes, search,
# CPU baselines), rocprofv3 kernel traces (bench command; single-stream encode legs of the three modes), PMC passes restricted to
# the roofline kernels + the whole-step HBM traffic, one full refresh end to end (default mode) and one in split mode at 2 M rows,
# the other BASELINE configurations.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
export ANCE_ROUND=r04
echo "== bench (default flags)"
timeout 900 python bench.py > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "rc=$?"; tail -c 400 gpurun_out/bench.log; tail -3 gpurun_out/bench.err
echo "== rocprofv3 kernel trace of the bench command"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench -o kt -- python bench.py --no-cpu-baseline > gpurun_out/prof_bench.log 2>&1; echo "rc=$?"
find gpurun_out/prof_bench -name "*kernel_stats.csv" | head -2
find gpurun_out -name "*kernel_trace.csv" -size +8M -delete
PMC_LEGS="search encode encode_split encode_fp32" bash scripts/gpu_pmc.sh > gpurun_out/pmc.log 2>&1; echo "pmc rc=$?"; tail -5 gpurun_out/pmc.log
echo "== full refresh (default mode)"
timeout 1500 python bench.py --full > gpurun_out/bench_full.log 2> gpurun_out/bench_full.err; echo "rc=$?"; tail -c 1200 gpurun_out/bench_full.log
rm -rf /tmp/ance_full
echo "== full refresh, split (fp32-grade) mode, 2 M passages / 100 k queries"
ANCE_ENCODER_SPLIT=1 timeout 900 python bench.py --full --n-passages 2000000 --full-queries 100000 > gpurun_out/bench_full_split_2m.log 2> gpurun_out/bench_full_split_2m.err; echo "rc=$?"; tail -c 900 gpurun_out/bench_full_split_2m.log
rm -rf /tmp/ance_full
echo "== other BASELINE configurations"
timeout 900 python scripts/bench_configs.py > gpurun_out/bench_configs.jsonl 2> gpurun_out/bench_configs.err; echo "rc=$?"; cat gpurun_out/bench_configs.jsonl | cut -c1-300
