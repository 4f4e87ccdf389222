#!/bin/bash
# same-box A/B of two builds of the library on the encode leg: ANCE_AMD_LIB=<.so> alternating
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/ab
export TMPDIR=/tmp
for i in 1 2; do
  for lib in prev cur; do
    p=$PWD/ance_amd/libance_amd.so; [ $lib = prev ] && p=$PWD/ance_amd/libance_amd_prev.so
    ANCE_AMD_LIB=$p timeout 600 python bench.py --skip-search --no-cpu-baseline --skip-precise --steps 6 --warmup 2 ${AB_ARGS:-} > gpurun_out/ab/e_${lib}_$i.json 2> gpurun_out/ab/e_${lib}_$i.err
    python -c "
import json,sys; d=json.loads(open('gpurun_out/ab/e_${lib}_$i.json').read().strip().splitlines()[-1]); bk=d['roofline']['by_kernel']
print('$lib $i: passages/s %.0f  %s' % (d['value'], d['roofline']['timing'][-60:])); print('   ' + '  '.join('%s %.1f' % (k, 1e3*v['ms_per_launch']) for k,v in bk.items()))"
  done
done
