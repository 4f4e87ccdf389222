#!/bin/bash
# Round 6: GELU epilogue on the packed fp32 pipe (v_pk_fma_f32) + one statistics store per RESLN pass, against the library of the
# previous commit (ance_amd/libance_amd_head.so): GEMM / encoder parity, then same-box A/B of the encode leg.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_gemm.py -q -x -p no:cacheprovider > gpurun_out/t_gemm.log 2>&1; echo "gemm rc=$?"; tail -3 gpurun_out/t_gemm.log
timeout 900 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_config1.py tests/test_gpu_dist.py -q -x -p no:cacheprovider > gpurun_out/t_enc.log 2>&1; echo "enc rc=$?"; tail -3 gpurun_out/t_enc.log
rm -f gpurun_out/ab_gelu.jsonl
enc() {  # name lib
  ANCE_AMD_LIB=$2 timeout 600 python bench.py --steps 5 --warmup 2 --skip-search --skip-precise --skip-slice --skip-other-configs --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']
print(json.dumps({'variant': '$1', 'passages_per_sec': d['value'], 'timing': r['timing'], 'by_kernel': {k: v['ms_per_launch'] for k, v in r['by_kernel'].items()}}))" >> gpurun_out/ab_gelu.jsonl
}
for rep in 1 2 3; do
  enc previous_commit $PWD/ance_amd/libance_amd_head.so
  enc packed_gelu+one_stats_store ""
done
cat gpurun_out/ab_gelu.jsonl
