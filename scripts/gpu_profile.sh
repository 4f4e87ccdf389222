#!/bin/bash
# rocprofv3 evidence for the bench command: kernel-trace stats, then PMC passes (HBM bytes) in
# their own runs.  Summaries are copied to gpurun_out/prof_summary/ (commit them under profiles/).
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/prof_summary
export TMPDIR=/tmp
BENCH="python bench.py --steps 2 --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-}"
echo "== kernel trace"
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_kt -o kt -- $BENCH > gpurun_out/prof_kt.log 2>&1
echo "rc=$?"; tail -2 gpurun_out/prof_kt.log
find gpurun_out/prof_kt -name "*stats*.csv" -exec cp {} gpurun_out/prof_summary/ \;
for c in FETCH_SIZE WRITE_SIZE; do
  echo "== pmc $c"
  timeout 900 rocprofv3 --pmc $c --kernel-trace -d gpurun_out/prof_$c -o pmc -- $BENCH > gpurun_out/prof_$c.log 2>&1
  echo "rc=$?"; tail -2 gpurun_out/prof_$c.log
done
python scripts/summarize_pmc.py gpurun_out > gpurun_out/prof_summary/pmc_summary.txt 2>&1
cat gpurun_out/prof_summary/pmc_summary.txt | head -40
ls -la gpurun_out/prof_summary
# keep the merge-back under 64 MiB: drop raw traces, keep summaries
find gpurun_out -name "*kernel_trace.csv" -size +8M -delete
find gpurun_out -name "*counter_collection.csv" -size +8M -delete
