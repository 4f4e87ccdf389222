#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python -m pytest tests/test_gpu_config1.py tests/test_nll.py tests/test_gpu_dist.py -q -m gpu -p no:cacheprovider 2>&1 | tail -3
