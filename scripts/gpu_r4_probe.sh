#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 300 python scripts/split_probe3.py 2>&1 | grep -v amdgpu.ids | tail -12
