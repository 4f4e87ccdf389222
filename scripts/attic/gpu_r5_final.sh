#!/bin/bash
# Round 5, final tree (131,072-token micro-batches by default): -m gpu suite + smoke, the bench line under the driver's flags, the
# single-stream kernel traces of the three modes, the PMC passes of the split encode leg (+ whole-step HBM bytes), the kernel
# trace of the bench command, one FULL refresh in the default arithmetic, the other BASELINE configurations.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
export ANCE_ROUND=r05
rm -f gpurun_out/encoder_parity.jsonl gpurun_out/config1_agreement.json gpurun_out/retrieval_agreement.json gpurun_out/e2e_agreement*.json
rm -rf gpurun_out/pmc gpurun_out/prof_bench
t() { echo "[$(date +%H:%M:%S)] $*"; }
t "pytest -m gpu"
timeout 1500 python -m pytest tests/ -q -m gpu -p no:cacheprovider > gpurun_out/t_all.log 2>&1; echo "pytest -m gpu rc=$?"; tail -4 gpurun_out/t_all.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
t "bench (driver flags)"
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "rc=$?"; tail -3 gpurun_out/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench.log').read().strip().splitlines()[-1])
print({k: d.get(k) for k in ('value','ms_per_step','errors')}, {k: d['roofline'].get(k) for k in ('achieved','frac','frac_executed')})
print('fp16', d['encode_fp16_fast']['value'], 'fp32', d['encode_fp32']['value'], 'search', d['search']['value'], 'slice', d['full_refresh_slice']['passages_per_sec'], d['full_refresh_slice']['wall_s'])
PY
t "pmc passes + traces"
PMC_LEGS="encode_split encode encode_fp32" PMC_TRACE_ONLY="encode" bash scripts/gpu_pmc.sh > gpurun_out/pmc.log 2>&1; echo "rc=$?"; grep -c "rc=0" gpurun_out/pmc.log; grep "rc=[1-9]" gpurun_out/pmc.log | head
t "rocprofv3 kernel trace of the bench command (5 steps)"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench -o kt -- python bench.py --no-cpu-baseline --steps 5 --warmup 2 > gpurun_out/prof_bench.log 2>&1; echo "rc=$?"
find gpurun_out -name "*kernel_trace.csv" -size +8M -delete
t "full refresh, 8,841,823 passages, default (split) arithmetic"
timeout 1500 python bench.py --full > gpurun_out/bench_full_split.log 2> gpurun_out/bench_full_split.err; echo "rc=$?"; tail -c 700 gpurun_out/bench_full_split.log
rm -rf /tmp/ance_full
t "other BASELINE configurations (split)"
timeout 600 python scripts/bench_configs.py --skip-search > gpurun_out/bench_configs.jsonl 2> gpurun_out/bench_configs.err; echo "rc=$?"; cat gpurun_out/bench_configs.jsonl | cut -c1-240
t done
