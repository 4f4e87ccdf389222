#!/bin/bash
# Round 5: per-workgroup stamps of the split GEMMs inside the encoder (measurement library), and the micro-batch size A/B in the
# split mode (65,536 vs 131,072 tokens per micro-batch).
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python scripts/gemm_split_stamps.py > gpurun_out/gemm_split_stamps.jsonl 2> gpurun_out/gemm_split_stamps.err; echo "rc=$?"; tail -2 gpurun_out/gemm_split_stamps.err; cat gpurun_out/gemm_split_stamps.jsonl | cut -c1-700
rm -f gpurun_out/ab_max_tokens_split.jsonl
for i in 1 2; do for mt in 65536 131072; do
  timeout 300 python bench.py --skip-search --no-cpu-baseline --skip-precise --skip-slice --steps 4 --warmup 1 --max-tokens $mt > gpurun_out/ab_mt.json 2>/dev/null
  python -c "
import json; d=json.loads(open('gpurun_out/ab_mt.json').read().strip().splitlines()[-1]); bk=d['roofline']['by_kernel']
print(json.dumps({'max_tokens': $mt, 'run': $i, 'passages_per_sec': d['value'], 'isolated': d['roofline']['timing'][-90:], 'ffn1_us': round(1e3*bk['gemm_ffn1']['ms_per_launch'],1)}))" | tee -a gpurun_out/ab_max_tokens_split.jsonl
done; done
