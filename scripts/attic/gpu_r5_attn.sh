#!/bin/bash
# Round 5: conflict-free V^T rows in the split attention's LDS: parity (bit-identical results expected: only LDS addresses change),
# then same-box A/B against round 4's layout (make variant NAME=vtold DEFS=-DANCE_ATTN_VT_OLD).
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/ab
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_dist.py tests/test_nll.py -q -x -p no:cacheprovider -k "split or golden or default_is or 512 or large_micro or gpu" > gpurun_out/t_attn.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/t_attn.log
rm -f gpurun_out/ab_attn_vt.jsonl
for i in 1 2 3; do
  for lib in vtold cur; do
    p=$PWD/ance_amd/libance_amd.so; [ $lib != cur ] && p=$PWD/ance_amd/libance_amd_$lib.so
    ANCE_AMD_LIB=$p timeout 600 python bench.py --skip-search --no-cpu-baseline --skip-precise --skip-slice --steps 6 --warmup 2 > gpurun_out/ab/a_${lib}_$i.json 2> gpurun_out/ab/a_${lib}_$i.err
    python -c "
import json; d=json.loads(open('gpurun_out/ab/a_${lib}_$i.json').read().strip().splitlines()[-1]); bk=d['roofline']['by_kernel']
print(json.dumps({'lib': '$lib', 'run': $i, 'passages_per_sec': d['value'], 'ms_per_step': d['ms_per_step'], 'attention_us': round(1e3*bk['attention']['ms_per_launch'],1), 'gemm_qk_us': round(1e3*bk['gemm_qk']['ms_per_launch'],1)}))" | tee -a gpurun_out/ab_attn_vt.jsonl
  done
done
