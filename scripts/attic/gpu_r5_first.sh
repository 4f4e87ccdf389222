#!/bin/bash
# Round 5, first GPU call: the split GEMM v2 (blocked pair rows, three products from four staged tiles) -- its direct tests incl.
# the fp16-subnormal MFMA probe, the whole -m gpu suite with split as the library default, smoke, and the same-box A/B of the
# split encode leg: round 4's scheme (libance_amd_splitv1.so) vs this tree vs the LDS-DMA spacing variant.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/ab
export TMPDIR=/tmp
rm -f gpurun_out/encoder_parity.jsonl gpurun_out/config1_agreement.json gpurun_out/retrieval_agreement.json gpurun_out/e2e_agreement*.json gpurun_out/ab_split.jsonl
timeout 600 python -m pytest tests/test_gpu_gemm.py -q -x -p no:cacheprovider > gpurun_out/t_gemm.log 2>&1; echo "gemm tests rc=$?"; tail -5 gpurun_out/t_gemm.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
for i in 1 2; do
  for lib in splitv1 cur gap1; do
    p=$PWD/ance_amd/libance_amd.so; [ $lib != cur ] && p=$PWD/ance_amd/libance_amd_$lib.so
    ANCE_AMD_LIB=$p timeout 600 python bench.py --skip-search --no-cpu-baseline --skip-precise --skip-slice --steps 6 --warmup 2 > gpurun_out/ab/s_${lib}_$i.json 2> gpurun_out/ab/s_${lib}_$i.err
    python -c "
import json; d=json.loads(open('gpurun_out/ab/s_${lib}_$i.json').read().strip().splitlines()[-1]); bk=d['roofline']['by_kernel']
print(json.dumps({'lib': '$lib', 'run': $i, 'passages_per_sec': d['value'], 'ms_per_step': d['ms_per_step'], 'isolated': d['roofline']['timing'][-90:], 'us_per_launch': {k: round(1e3*v['ms_per_launch'],1) for k,v in bk.items()}}))" | tee -a gpurun_out/ab_split.jsonl
  done
done
timeout 2400 python -m pytest tests/ -q -m gpu -p no:cacheprovider > gpurun_out/t_all.log 2>&1; echo "pytest -m gpu rc=$?"; tail -15 gpurun_out/t_all.log
