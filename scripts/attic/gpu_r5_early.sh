#!/bin/bash
# Round 5: where the LDS-DMAs of the split GEMM's K-tile are issued (PIPE_PAIR3_EARLY = 0 / 1 / 2): correctness of the two early
# forms (direct GEMM tests on every element, encoder goldens), then the same-box A/B.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/ab
export TMPDIR=/tmp
for lib in early1 early2; do
  ANCE_AMD_LIB=$PWD/ance_amd/libance_amd_$lib.so timeout 600 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_encoder.py -q -x -p no:cacheprovider -k "split or golden or subnormal or default_is" > gpurun_out/t_$lib.log 2>&1; echo "$lib tests rc=$?"; tail -3 gpurun_out/t_$lib.log
done
rm -f gpurun_out/ab_early.jsonl
for i in 1 2 3; do
  for lib in cur early1 early2; do
    p=$PWD/ance_amd/libance_amd.so; [ $lib != cur ] && p=$PWD/ance_amd/libance_amd_$lib.so
    ANCE_AMD_LIB=$p timeout 600 python bench.py --skip-search --no-cpu-baseline --skip-precise --skip-slice --steps 6 --warmup 2 > gpurun_out/ab/y_${lib}_$i.json 2> gpurun_out/ab/y_${lib}_$i.err
    python -c "
import json; d=json.loads(open('gpurun_out/ab/y_${lib}_$i.json').read().strip().splitlines()[-1]); bk=d['roofline']['by_kernel']
print(json.dumps({'lib': '$lib', 'run': $i, 'passages_per_sec': d['value'], 'ms_per_step': d['ms_per_step'], 'isolated': d['roofline']['timing'][-90:], 'us_per_launch': {k: round(1e3*v['ms_per_launch'],1) for k,v in bk.items()}}))" | tee -a gpurun_out/ab_early.jsonl
  done
done
