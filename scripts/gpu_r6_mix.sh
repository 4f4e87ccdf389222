#!/bin/bash
# Round 6: pair halves through v_fma_mix_f32 (pair_split4: v - hi; the RESLN epilogue: hi + lo, then the packed fp32 pipe spelled out)
# -- the device probe of the instruction forms, the GEMM / encoder bit tests, then the tree's library against HEAD's (6980977) on one box.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
./tools/tr16_probe > gpurun_out/tr16_probe.txt 2>&1; echo "probe rc=$?"; head -2 gpurun_out/tr16_probe.txt
timeout 1200 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_encoder.py tests/test_gpu_config1.py -q -x -p no:cacheprovider > gpurun_out/t_mix.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/t_mix.log
rm -f gpurun_out/ab_mix.jsonl
one() {  # name lib
  ANCE_AMD_LIB=$2 timeout 600 python bench.py --steps 5 --warmup 2 --skip-search --skip-precise --skip-slice --skip-other-configs --no-cpu-baseline 2>gpurun_out/ab_mix_$1.err | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']
print(json.dumps({'variant': '$1', 'passages_per_sec': d['value'], 'ms_per_step': d['ms_per_step'], 'timing': r['timing'][-90:], 'by_kernel': {k: round(v['ms_per_launch'], 4) for k, v in r['by_kernel'].items()}}))" >> gpurun_out/ab_mix.jsonl
}
for rep in 1 2 3; do
  one head_6980977 ance_amd/libance_amd_head.so
  one tree ance_amd/libance_amd.so
done
cat gpurun_out/ab_mix.jsonl
