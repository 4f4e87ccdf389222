#!/bin/bash
# Round 6: split attention with V row-major in LDS + LDS transpose reads (ANCE_ATTN_TR, default 1) and the hybrid streaming GEMM
# (ANCE_GEMM_STREAM, default 1): bit-equality test, then same-box A/B of the encode leg.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_encoder.py -q -x -p no:cacheprovider -k "transpose or split or golden" > gpurun_out/t_enc.log 2>&1; echo "enc rc=$?"; tail -5 gpurun_out/t_enc.log
rm -f gpurun_out/ab_attn.jsonl
one() {  # name stream tr
  ANCE_GEMM_STREAM=$2 ANCE_ATTN_TR=$3 timeout 600 python bench.py --steps 5 --warmup 2 --skip-search --skip-precise --skip-slice --skip-other-configs --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']
print(json.dumps({'variant': '$1', 'passages_per_sec': d['value'], 'ms_per_step': d['ms_per_step'], 'timing': r['timing'], 'by_kernel': {k: v['ms_per_launch'] for k, v in r['by_kernel'].items()}, 'all_gemm_tflops': r['all_gemm_tflops']}))" >> gpurun_out/ab_attn.jsonl
}
for rep in 1 2 3; do
  one round5_state 0 0
  one stream_qkv_ffn1 1 0
  one stream_qkv_ffn1+attn_tr 1 1
done
cat gpurun_out/ab_attn.jsonl
