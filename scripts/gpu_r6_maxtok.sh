#!/bin/bash
# Round 6: micro-batch size of the split encoder: 131,072 (the drivers' default) vs 262,144 tokens, 65,536-passage encode calls (the
# job's block size), same-box alternation.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/ab_maxtok.jsonl
enc() {  # max_tokens
  timeout 600 python bench.py --steps 3 --warmup 1 --encode-block 65536 --max-tokens $1 --skip-search --skip-precise --skip-slice --skip-other-configs --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']
print(json.dumps({'max_tokens': $1, 'passages_per_sec': d['value'], 'timing': r['timing'], 'by_kernel': {k: v['ms_per_launch'] for k, v in r['by_kernel'].items()}, 'all_gemm_tflops': r['all_gemm_tflops']}))" >> gpurun_out/ab_maxtok.jsonl
}
for rep in 1 2; do
  enc 131072
  enc 262144
  enc 196608
done
cat gpurun_out/ab_maxtok.jsonl
