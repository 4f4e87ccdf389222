#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_search.py tests/test_gpu_dist.py -q -p no:cacheprovider -x 2>&1 | tail -5
timeout 600 python scripts/search_exchange_probe.py 2>&1 | grep -v amdgpu.ids | tail -40
