#!/bin/bash
# Round 6: does a stagger set up ONCE survive on the encoder's two streams?  ANCE_GEMM_STAGGER=s sleeps of ~4.8 us per XCD index in every
# ANCE_GEMM_STAGGER_EVERY-th launch of a persistent GEMM (24 launches = one micro-batch); the product's two-stream passages/s is the figure.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/ab_stagger2.jsonl
one() {  # name stagger every
  ANCE_GEMM_STAGGER=$2 ANCE_GEMM_STAGGER_EVERY=$3 timeout 600 python bench.py --steps 5 --warmup 2 --skip-search --skip-precise --skip-slice --skip-other-configs --no-cpu-baseline 2>gpurun_out/ab_stagger2_$1.err | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']
print(json.dumps({'variant': '$1', 'passages_per_sec': d['value'], 'ms_per_step': d['ms_per_step'], 'timing': r['timing'][-90:], 'by_kernel': {k: round(v['ms_per_launch'], 4) for k, v in r['by_kernel'].items()}}))" >> gpurun_out/ab_stagger2.jsonl
}
for rep in 1 2; do
  one none 0 1
  one s1_every24 1 24
  one s2_every24 2 24
  one s1_every240 1 240
  one s2_every240 2 240
  one s1_every6 1 6
done
cat gpurun_out/ab_stagger2.jsonl
