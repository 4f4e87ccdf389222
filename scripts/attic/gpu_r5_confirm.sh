#!/bin/bash
# Round 5, last GPU call: the whole -m gpu suite + smoke + the bench line under the driver's flags on the committed tree (the
# committed traces / counters are what `frac_from_profiles` and `roofline.traffic` read).
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/ -q -m gpu -p no:cacheprovider > gpurun_out/t_all.log 2>&1; echo "pytest -m gpu rc=$?"; tail -4 gpurun_out/t_all.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "rc=$?"; tail -3 gpurun_out/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench.log').read().strip().splitlines()[-1])
print({k: d.get(k) for k in ('value','ms_per_step','errors')}, {k: d['roofline'].get(k) for k in ('achieved','frac','frac_executed','frac_from_profiles')})
print('traffic round', (d['roofline'].get('traffic') or {}).get('round'))
print('fp16', d['encode_fp16_fast']['value'], 'fp32', d['encode_fp32']['value'], 'search', d['search']['value'], 'slice', d['full_refresh_slice']['passages_per_sec'], d['full_refresh_slice']['wall_s'])
PY
