#!/bin/bash
# Round 5: non-temporal output stores in the split GEMM epilogues (make variant NAME=ntstore DEFS=-DANCE_EPI_NT_STORE): same-box A/B.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/ab
export TMPDIR=/tmp
ANCE_AMD_LIB=$PWD/ance_amd/libance_amd_ntstore.so timeout 600 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_encoder.py -q -x -p no:cacheprovider -k "split or golden12" > gpurun_out/t_nt.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/t_nt.log
rm -f gpurun_out/ab_nt_store.jsonl
for i in 1 2 3; do
  for lib in cur ntstore; do
    p=$PWD/ance_amd/libance_amd.so; [ $lib != cur ] && p=$PWD/ance_amd/libance_amd_$lib.so
    ANCE_AMD_LIB=$p timeout 600 python bench.py --skip-search --no-cpu-baseline --skip-precise --skip-slice --steps 6 --warmup 2 > gpurun_out/ab/n_${lib}_$i.json 2> gpurun_out/ab/n_${lib}_$i.err
    python -c "
import json; d=json.loads(open('gpurun_out/ab/n_${lib}_$i.json').read().strip().splitlines()[-1]); bk=d['roofline']['by_kernel']
print(json.dumps({'lib': '$lib', 'run': $i, 'passages_per_sec': d['value'], 'ms_per_step': d['ms_per_step'], 'us_per_launch': {k: round(1e3*v['ms_per_launch'],1) for k,v in bk.items() if k.startswith('gemm') or k=='attention'}}))" | tee -a gpurun_out/ab_nt_store.jsonl
  done
done
