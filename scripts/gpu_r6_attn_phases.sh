#!/bin/bash
# Round 6: what the split attention kernel's phases cost -- the product against measurement builds without the K / V staging
# (ATTN_DIAG_NO_STAGE: no panel loads, no split, zeros written to the LDS), without the panel's global loads only (ATTN_DIAG_NO_LOAD),
# without the key-block loop (ATTN_DIAG_NO_COMPUTE).  (The padded V rows of the first transpose-read form, ANCE_ATTN_V_PAD, were a fourth variant at commit fff5260.)  Also the
# v_fma_mix_f32 probe of pair_split4 and the GEMM / encoder bit tests of the tree.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
./tools/tr16_probe > gpurun_out/tr16_probe.txt 2>&1; echo "probe rc=$?"; head -3 gpurun_out/tr16_probe.txt
timeout 900 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_encoder.py -q -x -p no:cacheprovider > gpurun_out/t_ge.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/t_ge.log
rm -f gpurun_out/ab_attn_phases.jsonl
one() {  # name lib
  ANCE_AMD_LIB=$2 timeout 600 python bench.py --steps 3 --warmup 1 --skip-search --skip-precise --skip-slice --skip-other-configs --no-cpu-baseline 2>gpurun_out/ab_attn_phases_$1.err | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']
print(json.dumps({'variant': '$1', 'passages_per_sec': d['value'], 'ms_per_step': d['ms_per_step'], 'attention_ms': r['by_kernel'].get('attention', {}).get('ms_per_launch'), 'by_kernel': {k: round(v['ms_per_launch'], 4) for k, v in r['by_kernel'].items()}}))" >> gpurun_out/ab_attn_phases.jsonl
}
for rep in 1 2; do
  one product ance_amd/libance_amd.so
  for v in ${VARIANTS:-nostage noload nocompute}; do one $v ance_amd/libance_amd_$v.so; done
done
cat gpurun_out/ab_attn_phases.jsonl
