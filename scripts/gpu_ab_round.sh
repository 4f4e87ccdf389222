#!/bin/bash
# same-box A/B of this round's encoder against round 2's form of every piece (the A/B switches of include/ance_amd.h)
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/ab
export TMPDIR=/tmp
for i in 1 2; do
  for cfg in r2form default; do
    envs="A=1"; [ $cfg = r2form ] && envs="ANCE_LN_FOLD=0 ANCE_HEAD_MFMA=0 ANCE_ATTN_COAL=0"
    env $envs timeout 600 python bench.py --skip-search --no-cpu-baseline --skip-precise --steps 6 --warmup 2 > gpurun_out/ab/r_${cfg}_$i.json 2> gpurun_out/ab/r_${cfg}_$i.err
    python -c "
import json; d=json.loads(open('gpurun_out/ab/r_${cfg}_$i.json').read().strip().splitlines()[-1]); bk=d['roofline']['by_kernel']
print(json.dumps({'config': '$cfg', 'run': $i, 'passages_per_sec': d['value'], 'ms_per_step': d['ms_per_step'], 'isolated': d['roofline']['timing'][-62:], 'us_per_launch': {k: round(1e3*v['ms_per_launch'],1) for k,v in bk.items()}}))" | tee -a gpurun_out/ab_round.jsonl
  done
done
