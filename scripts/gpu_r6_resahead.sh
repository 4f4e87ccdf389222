#!/bin/bash
# Round 6: the RESLN epilogue with the residual pair rows of pass y + 1 requested at the top of pass y (the tree) against the form that
# requests each pass's rows inside the pass (make variant NAME=nopf DEFS=-DANCE_RESLN_NO_PREFETCH): bit tests, then three alternations.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_encoder.py -q -x -p no:cacheprovider > gpurun_out/t_resahead.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/t_resahead.log
rm -f gpurun_out/ab_resahead.jsonl
one() {  # name lib
  ANCE_AMD_LIB=$2 timeout 600 python bench.py --steps 5 --warmup 2 --skip-search --skip-precise --skip-slice --skip-other-configs --no-cpu-baseline 2>gpurun_out/ab_resahead_$1.err | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']
print(json.dumps({'variant': '$1', 'passages_per_sec': d['value'], 'ms_per_step': d['ms_per_step'], 'timing': r['timing'][-90:], 'by_kernel': {k: round(v['ms_per_launch'], 4) for k, v in r['by_kernel'].items()}}))" >> gpurun_out/ab_resahead.jsonl
}
for rep in 1 2 3; do
  one residual_in_pass ance_amd/libance_amd_nopf.so
  one residual_a_pass_ahead ance_amd/libance_amd.so
done
cat gpurun_out/ab_resahead.jsonl
