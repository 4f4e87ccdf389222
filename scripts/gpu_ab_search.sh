#!/bin/bash
# same-box A/B of two builds of the library on the search leg: ANCE_AMD_LIB=<.so> alternating
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/ab
export TMPDIR=/tmp
for i in 1 2; do
  for lib in prev cur; do
    p=$PWD/ance_amd/libance_amd.so; [ $lib = prev ] && p=$PWD/ance_amd/libance_amd_prev.so
    ANCE_AMD_LIB=$p timeout 600 python bench.py --skip-encode --no-cpu-baseline --steps 5 --warmup 2 ${AB_ARGS:-} > gpurun_out/ab/s_${lib}_$i.json 2> gpurun_out/ab/s_${lib}_$i.err
    python -c "
import json,sys; d=json.loads(open('gpurun_out/ab/s_${lib}_$i.json').read().strip().splitlines()[-1]); s=d['search']; e=s.get('encoder_like') or {}
print('$lib $i: q/s %.0f filter %.1f ms frac %.3f rescore %.2f | encoder-like q/s %.0f filter %.1f ms' % (s['value'], s['roofline']['ms_per_launch'], s['roofline']['frac'], s['roofline']['rescore_ms_per_launch'], e.get('value',0), e.get('filter_ms_per_launch',0)))"
  done
done
