#!/bin/bash
# Round 6 evidence on the final tree: instruction probes, -m gpu suite + smoke, the bench line under the driver's flags, rocprofv3
# kernel traces (bench command; single-stream encode legs of the three modes; search leg), PMC passes (split encode leg = the headline
# arithmetic incl. SQ counters, search leg, whole-step HBM bytes), one FULL refresh at 8,841,823 passages in the default (split)
# arithmetic, the search of configuration 4, and the N > 1 path of the bench on one GPU over gloo (functional).
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
export ANCE_ROUND=r06
rm -f gpurun_out/encoder_parity.jsonl gpurun_out/config1_agreement.json gpurun_out/retrieval_agreement.json gpurun_out/e2e_agreement*.json
rm -rf gpurun_out/pmc gpurun_out/prof_bench
t() { echo "[$(date +%H:%M:%S)] $*"; }
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx" | head -4 > gpurun_out/device.txt; nproc >> gpurun_out/device.txt
t "probes"; tools/tr16_probe > gpurun_out/tr16_probe.txt 2>&1; echo "probe rc=$?"; grep -E "rule|RULE" gpurun_out/tr16_probe.txt
t "pytest -m gpu"
timeout 1500 python -m pytest tests/ -q -m gpu -p no:cacheprovider --durations=10 > gpurun_out/t_all.log 2>&1; echo "pytest -m gpu rc=$?"; tail -4 gpurun_out/t_all.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
t "bench (driver flags)"
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "rc=$?"; tail -3 gpurun_out/bench.err; tail -c 400 gpurun_out/bench.log
t "rocprofv3 kernel trace of the bench command (5 steps)"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench -o kt -- python bench.py --no-cpu-baseline --steps 5 --warmup 2 > gpurun_out/prof_bench.log 2>&1; echo "rc=$?"
find gpurun_out -name "*kernel_trace.csv" -size +8M -delete
t "pmc passes"
PMC_LEGS="search encode_split encode encode_fp32" PMC_TRACE_ONLY="encode" bash scripts/gpu_pmc.sh > gpurun_out/pmc.log 2>&1; echo "rc=$?"; grep -c "rc=0" gpurun_out/pmc.log; grep "rc=[1-9]" gpurun_out/pmc.log | head
t "full refresh, 8,841,823 passages, default (split) arithmetic"
timeout 1500 python bench.py --full > gpurun_out/bench_full_split.log 2> gpurun_out/bench_full_split.err; echo "rc=$?"; tail -c 900 gpurun_out/bench_full_split.log
rm -rf /tmp/ance_full
t "search of configuration 4 + the other configurations at 2 M tokens per step"
timeout 600 python scripts/bench_configs.py > gpurun_out/bench_configs.jsonl 2> gpurun_out/bench_configs.err; echo "rc=$?"; cat gpurun_out/bench_configs.jsonl | cut -c1-260
t "bench, 2 ranks on one GPU over gloo (functional)"
ANCE_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 2 --warmup 1 --n-passages 2000000 --skip-precise --no-cpu-baseline > gpurun_out/bench_2rank_gloo.json 2> gpurun_out/bench_2rank_gloo.err; echo "rc=$?"; tail -c 400 gpurun_out/bench_2rank_gloo.json
t done
