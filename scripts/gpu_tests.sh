#!/bin/bash
# the whole -m gpu suite, as the driver runs it at round end
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
rm -f gpurun_out/encoder_parity.jsonl gpurun_out/config1_agreement.json gpurun_out/retrieval_agreement.json gpurun_out/e2e_agreement*.json
timeout 2400 python -m pytest tests/ -q -m gpu -p no:cacheprovider --durations=15 > gpurun_out/t_all.log 2>&1; echo "pytest -m gpu rc=$?"
tail -40 gpurun_out/t_all.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
