#!/bin/bash
# Round 4, last GPU call: the whole -m gpu suite + smoke on the final tree, the default bench line (now reading this round's
# committed profiles), a 2-rank functional run of the bench on one GPU (gloo), one FULL refresh in split mode.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/encoder_parity.jsonl gpurun_out/config1_agreement.json gpurun_out/retrieval_agreement.json gpurun_out/e2e_agreement*.json
timeout 2400 python -m pytest tests/ -q -m gpu -p no:cacheprovider > gpurun_out/t_all.log 2>&1; echo "pytest -m gpu rc=$?"; tail -4 gpurun_out/t_all.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "== bench (default flags)"
timeout 900 python bench.py > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "rc=$?"; tail -3 gpurun_out/bench.err
echo "== bench, 2 ranks on one GPU over gloo (functional)"
ANCE_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 2 --warmup 1 --n-passages 2000000 --skip-precise --no-cpu-baseline > gpurun_out/bench_2rank_gloo.json 2> gpurun_out/bench_2rank_gloo.err; echo "rc=$?"; tail -c 300 gpurun_out/bench_2rank_gloo.json
echo "== full refresh, split (fp32-grade) mode, full size"
ANCE_ENCODER_SPLIT=1 timeout 1500 python bench.py --full > gpurun_out/bench_full_split.log 2> gpurun_out/bench_full_split.err; echo "rc=$?"; tail -c 900 gpurun_out/bench_full_split.log
rm -rf /tmp/ance_full
