#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== bench (default flags)"
timeout 900 python bench.py > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "rc=$?"; tail -3 gpurun_out/bench.err
echo "== other BASELINE configurations, split mode"
ANCE_ENCODER_SPLIT=1 timeout 900 python scripts/bench_configs.py > gpurun_out/bench_configs_split.jsonl 2> gpurun_out/bench_configs_split.err; echo "rc=$?"; cut -c1-260 gpurun_out/bench_configs_split.jsonl
