#!/bin/bash
# Round 6: what bounds the epilogue of the streaming split GEMMs (QKV, FFN1: ~13 us of a 57 us tile) -- measurement builds without
# the output stores (GS_DIAG_NO_STORE), without the GELU (GS_DIAG_NO_GELU), and with the XCDs' workgroups started GS_DIAG_STAGGER x k
# sleeps apart (are the epilogues of all CUs one HBM write burst?).
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/ab_epi_phases.jsonl
one() {  # name lib
  ANCE_AMD_LIB=$2 timeout 600 python bench.py --steps 3 --warmup 1 --skip-search --skip-precise --skip-slice --skip-other-configs --no-cpu-baseline 2>gpurun_out/ab_epi_phases_$1.err | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']
print(json.dumps({'variant': '$1', 'passages_per_sec': d['value'], 'ms_per_step': d['ms_per_step'], 'timing': r['timing'][-90:], 'by_kernel': {k: round(v['ms_per_launch'], 4) for k, v in r['by_kernel'].items()}}))" >> gpurun_out/ab_epi_phases.jsonl
}
for rep in 1 2; do
  one product ance_amd/libance_amd.so
  for v in ${VARIANTS:-nostore nogelu stag1 stag2}; do one $v ance_amd/libance_amd_$v.so; done
done
cat gpurun_out/ab_epi_phases.jsonl
