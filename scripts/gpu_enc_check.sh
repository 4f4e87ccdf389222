#!/bin/bash
# Encoder-side change check: parity tests, probe, short bench.
set -u
cd "$(dirname "$0")/.."
export LD_LIBRARY_PATH=/opt/rocm/lib:${LD_LIBRARY_PATH:-}
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_encoder.py tests/test_gpu_e2e.py tests/test_gpu_dpr.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
tail -4 gpurun_out/encoder_parity.jsonl | cut -c1-200
ANCE_CLS_TAIL=0 timeout 600 python -m pytest tests/test_gpu_encoder.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -1
tools/abi_probe encode 65536 128 12 3 | tail -1
timeout 600 python bench.py --no-cpu-baseline --skip-search --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); r=d['roofline']; print('pps',d['value']); print({k:(round(v['ms_per_launch']*1e3,1), v['tflops'] and round(v['tflops'])) for k,v in r['by_kernel'].items()})"
