#!/bin/bash
# Round 5: the one-phase-per-K-tile pipeline of the split GEMM (Pipe256One, two workgroup barriers per K-tile instead of four):
# correctness (direct GEMM tests on every element, the race screen = the same tests three times, encoder goldens, the multi-rank
# byte-identity test), then the same-box A/B against the coarse two-phase schedule.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/ab
export TMPDIR=/tmp
L=$PWD/ance_amd/libance_amd_${VARIANT:-onephase}.so
for rep in 1 2 3; do
  ANCE_AMD_LIB=$L timeout 600 python -m pytest tests/test_gpu_gemm.py -q -x -p no:cacheprovider -k "split or subnormal" > gpurun_out/t_one_$rep.log 2>&1; echo "gemm tests rep $rep rc=$?"; tail -2 gpurun_out/t_one_$rep.log
done
ANCE_AMD_LIB=$L timeout 900 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_config1.py tests/test_gpu_dist.py -q -x -p no:cacheprovider -k "split or golden or default_is or config1 or 512" > gpurun_out/t_one_enc.log 2>&1; echo "encoder / config1 / dist tests rc=$?"; tail -3 gpurun_out/t_one_enc.log
rm -f gpurun_out/ab_onephase.jsonl
for i in 1 2 3; do
  for lib in cur ${VARIANT:-onephase}; do
    p=$PWD/ance_amd/libance_amd.so; [ $lib != cur ] && p=$PWD/ance_amd/libance_amd_$lib.so
    ANCE_AMD_LIB=$p timeout 600 python bench.py --skip-search --no-cpu-baseline --skip-precise --skip-slice --steps 6 --warmup 2 > gpurun_out/ab/o_${lib}_$i.json 2> gpurun_out/ab/o_${lib}_$i.err
    python -c "
import json; d=json.loads(open('gpurun_out/ab/o_${lib}_$i.json').read().strip().splitlines()[-1]); bk=d['roofline']['by_kernel']
print(json.dumps({'lib': '$lib', 'run': $i, 'passages_per_sec': d['value'], 'ms_per_step': d['ms_per_step'], 'isolated': d['roofline']['timing'][-90:], 'us_per_launch': {k: round(1e3*v['ms_per_launch'],1) for k,v in bk.items()}}))" | tee -a gpurun_out/ab_onephase.jsonl
  done
done
