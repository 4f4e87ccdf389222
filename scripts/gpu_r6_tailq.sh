#!/bin/bash
# Round 6: the CLS-only tail projects Q for the [CLS] rows only (K | V of every token + a compact Q GEMM instead of the full QKV GEMM in
# the last layer): encoder / job tests (incl. bit equality with the full last layer), then the tree against the previous form
# (libance_amd_fullq.so = the tree before this change), three alternations; both matrix-core modes.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/t_tailq.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/t_tailq.log
rm -f gpurun_out/ab_tailq.jsonl
one() {  # name lib
  ANCE_AMD_LIB=$2 timeout 600 python bench.py --steps 5 --warmup 2 --skip-search --skip-slice --skip-other-configs --no-cpu-baseline 2>gpurun_out/ab_tailq_$1.err | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']
print(json.dumps({'variant': '$1', 'passages_per_sec': d['value'], 'ms_per_step': d['ms_per_step'], 'timing': r['timing'][-90:], 'fp16_fast_passages_per_sec': d['encode_fp16_fast']['value'], 'by_kernel': {k: round(v['ms_per_launch'], 4) for k, v in r['by_kernel'].items()}}))" >> gpurun_out/ab_tailq.jsonl
}
for rep in 1 2 3; do
  one full_qkv_in_the_last_layer ance_amd/libance_amd_fullq.so
  one compact_q_in_the_last_layer ance_amd/libance_amd.so
done
cat gpurun_out/ab_tailq.jsonl
