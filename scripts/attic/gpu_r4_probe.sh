#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python -m pytest tests/test_gpu_dpr.py -q -p no:cacheprovider 2>&1 | tail -5
