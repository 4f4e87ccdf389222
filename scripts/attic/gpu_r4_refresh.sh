#!/bin/bash
# after the split-attention staging change: the whole GPU suite, the split leg's traces / counters, the default bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp ANCE_ROUND=r04
bash scripts/gpu_tests.sh
PMC_LEGS="encode_split" bash scripts/gpu_pmc.sh > gpurun_out/pmc_split.log 2>&1; echo "pmc rc=$?"
timeout 900 python bench.py > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench rc=$?"; tail -c 300 gpurun_out/bench.log
