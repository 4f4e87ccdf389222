#!/bin/bash
# Round 6: persistent GEMM grids smaller than the chip (ANCE_GEMM_STREAM_WGS) with the encoder's two internal streams -- does
# leaving CUs to the other micro-batch's kernels (attention: HBM-bound; the GEMMs: matrix-pipe-bound) pay?
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/ab_stream_wgs.jsonl
one() {  # wgs
  ANCE_GEMM_STREAM_WGS=$1 timeout 600 python bench.py --steps 4 --warmup 1 --skip-search --skip-precise --skip-slice --skip-other-configs --no-cpu-baseline 2>gpurun_out/ab_stream_wgs_$1.err | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']
print(json.dumps({'stream_wgs': $1, 'passages_per_sec': d['value'], 'ms_per_step': d['ms_per_step'], 'timing': r['timing'][-90:], 'by_kernel': {k: round(v['ms_per_launch'], 4) for k, v in r['by_kernel'].items()}}))" >> gpurun_out/ab_stream_wgs.jsonl
}
for rep in 1 2; do
  for n in ${WGS:-256 248 224 192 128}; do one $n; done
done
cat gpurun_out/ab_stream_wgs.jsonl
