#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_config1.py tests/test_gpu_search.py tests/test_gpu_dist.py -q -p no:cacheprovider 2>&1 | tail -4
python -c "import json;d=json.load(open('gpurun_out/config1_agreement.json'));print({k:v for k,v in d.items() if k.startswith('shuffle')})"
