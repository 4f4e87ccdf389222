#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
export LD_LIBRARY_PATH=/opt/rocm/lib:${LD_LIBRARY_PATH:-}
timeout 2400 python -m pytest tests -x -q -m gpu -p no:cacheprovider 2>&1 | tail -5
tools/abi_probe search 8841823 32768 200 2 | tail -1
tools/abi_probe search 8841823 4096 200 2 | tail -1
tools/abi_probe search 8841823 6980 100 2 | tail -1
tools/abi_probe encode 65536 128 12 3 | tail -1
timeout 600 python bench.py --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/bench_exp11.json 2>gpurun_out/bench_exp11.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_exp11.json'))
r=d['roofline']
print('pps',d['value'], 'search', d['search']['value'], d['search']['roofline']['achieved'], d.get('errors'))
print({k:(round(v['ms_per_launch']*1e3,1), v['tflops'] and round(v['tflops'])) for k,v in r['by_kernel'].items()})
PY
