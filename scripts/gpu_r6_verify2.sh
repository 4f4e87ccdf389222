#!/bin/bash
# Round 6: GEMM / encoder parity on the current tree, then same-box A/B of the encode leg: round-5 kernels' form (library
# libance_amd_epi4col.so, ANCE_GEMM_STREAM=0, ANCE_ATTN_TR=0) / product / product with the RESLN GEMMs streaming too (ANCE_GEMM_STREAM=2).
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_gemm.py -q -x -p no:cacheprovider > gpurun_out/t_gemm.log 2>&1; echo "gemm rc=$?"; tail -3 gpurun_out/t_gemm.log
timeout 900 python -m pytest tests/test_gpu_encoder.py -q -x -p no:cacheprovider > gpurun_out/t_enc.log 2>&1; echo "enc rc=$?"; tail -3 gpurun_out/t_enc.log
rm -f gpurun_out/ab_round.jsonl
enc() {  # name lib stream attn_tr
  ANCE_AMD_LIB=$2 ANCE_GEMM_STREAM=$3 ANCE_ATTN_TR=$4 timeout 600 python bench.py --steps 5 --warmup 2 --skip-search --skip-precise --skip-slice --skip-other-configs --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']
print(json.dumps({'variant': '$1', 'passages_per_sec': d['value'], 'timing': r['timing'], 'by_kernel': {k: v['ms_per_launch'] for k, v in r['by_kernel'].items()}}))" >> gpurun_out/ab_round.jsonl
}
for rep in 1 2 3; do
  enc round5_kernels $PWD/ance_amd/libance_amd_epi4col.so 0 0
  enc product "" 1 1
  enc product+resln_streaming "" 2 1
done
cat gpurun_out/ab_round.jsonl
