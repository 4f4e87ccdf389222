#!/bin/bash
# Round 6: LDS bank conflicts of the split attention with the unpadded, half-row-swapped V rows (the product) against the padded
# rows of the first transpose-read form (ANCE_ATTN_V_PAD build of commit fff5260, libance_amd_vpad.so): one SQ counter pass each (never combined with other trace domains).
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/attn_lds
export TMPDIR=/tmp
export ANCE_ENCODER_STREAMS=1
for v in product vpad; do
  lib=$PWD/ance_amd/libance_amd.so; [ $v = vpad ] && lib=$PWD/ance_amd/libance_amd_vpad.so
  ANCE_AMD_LIB=$lib timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --kernel-trace --kernel-include-regex "attention_split_kernel" --output-format csv -d gpurun_out/attn_lds/$v -o pmc -- python scripts/encode_mode_leg.py split 1 8192 > gpurun_out/attn_lds/$v.log 2>&1; echo "$v rc=$?"
done
python - <<'PY'
import csv, glob, collections, json
out = {}
for v in ("product", "vpad"):
    agg = collections.defaultdict(list)
    for f in glob.glob('gpurun_out/attn_lds/%s/**/*counter_collection.csv' % v, recursive=True):
        for r in csv.DictReader(open(f)):
            if 'attention_split_kernel' in r['Kernel_Name']:
                agg[r['Counter_Name']].append(float(r['Counter_Value']))
    line = {c: sum(x) / len(x) for c, x in agg.items()}
    line['dispatches'] = max((len(x) for x in agg.values()), default=0)
    if line.get('SQ_LDS_IDX_ACTIVE'):
        line['lds_bank_conflict_per_lds_active_cycle'] = round(line['SQ_LDS_BANK_CONFLICT'] / line['SQ_LDS_IDX_ACTIVE'], 3)
    if line.get('SQ_WAVE_CYCLES'):
        line['lds_wait_frac_of_wave_cycles'] = round(line.get('SQ_WAIT_INST_LDS', 0.0) / line['SQ_WAVE_CYCLES'], 4)
    out[v] = line
    print(v, json.dumps(line))
json.dump(out, open('gpurun_out/attn_lds/attention_lds_counters.json', 'w'), indent=1)
PY
find gpurun_out/attn_lds -name "*kernel_trace.csv" -delete
