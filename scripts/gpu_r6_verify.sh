#!/bin/bash
# Round 6: the whole GPU suite on the current tree, the instruction probes, and a same-box A/B of the encode leg against the library
# of the previous epilogue / attention form (ance_amd/libance_amd_epi4col.so).
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== probes"; tools/tr16_probe > gpurun_out/tr16_probe.txt 2>&1; echo "probe rc=$?"; grep -E "rule|RULE" gpurun_out/tr16_probe.txt
bash scripts/gpu_check.sh nobench
rm -f gpurun_out/ab_epi8.jsonl
enc() {  # name lib attn_tr
  ANCE_AMD_LIB=$2 ANCE_ATTN_TR=$3 timeout 600 python bench.py --steps 5 --warmup 2 --skip-search --skip-precise --skip-slice --skip-other-configs --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']
print(json.dumps({'variant': '$1', 'passages_per_sec': d['value'], 'timing': r['timing'], 'by_kernel': {k: v['ms_per_launch'] for k, v in r['by_kernel'].items()}}))" >> gpurun_out/ab_epi8.jsonl
}
for rep in 1 2 3; do
  enc epilogues_8_byte+attention_round5 $PWD/ance_amd/libance_amd_epi4col.so 0
  enc epilogues_16_byte+attention_round5 "" 0
  enc epilogues_16_byte+attention_tr_wide_stores "" 1
done
cat gpurun_out/ab_epi8.jsonl
