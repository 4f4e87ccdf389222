#!/bin/bash
# Round 5: where the waves of the split GEMMs spend their cycles (SQ counters, one pass): parked at s_waitcnt / barriers, issue-stalled,
# issuing; LDS bank conflicts and LDS-array activity.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
export ANCE_ENCODER_STREAMS=1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --kernel-include-regex "gemm256_split_kernel|attention_split_kernel" --output-format csv -d gpurun_out/pmc/SQ_encode_split -o pmc -- python scripts/encode_mode_leg.py split > gpurun_out/pmc/SQ_encode_split.log 2>&1; echo "rc=$?"; tail -2 gpurun_out/pmc/SQ_encode_split.log | cut -c1-200
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM --kernel-trace --kernel-include-regex "gemm256_split_kernel" --output-format csv -d gpurun_out/pmc/SQ2_encode_split -o pmc -- python scripts/encode_mode_leg.py split > gpurun_out/pmc/SQ2_encode_split.log 2>&1; echo "rc=$?"; tail -2 gpurun_out/pmc/SQ2_encode_split.log | cut -c1-200
python - <<'PY'
import csv, glob, collections, re, json
out={}
for d in ('SQ_encode_split','SQ2_encode_split'):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob('gpurun_out/pmc/%s/**/*counter_collection.csv'%d, recursive=True):
        for r in csv.DictReader(open(f)):
            m=re.search(r'(gemm256_split_kernel<\d+>|attention_split_kernel)', r['Kernel_Name'])
            if not m: continue
            g=int(r.get('Grid_Size',0) or 0)
            k=m.group(1)+(' tail' if ('gemm' in k if False else False) else '')
            if 'gemm' in m.group(1) and g < 200000: continue   # skip the CLS-tail launches
            agg[m.group(1)][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items():
        out.setdefault(k,{}).update({c: sum(x)/len(x) for c,x in v.items()})
for k,v in out.items():
    wc=v.get('SQ_WAVE_CYCLES',0)
    line={c: round(x) for c,x in v.items()}
    if wc:
        line['frac_of_wave_cycles']={c: round(v[c]/wc,3) for c in ('SQ_WAIT_ANY','SQ_WAIT_INST_ANY','SQ_ACTIVE_INST_ANY','SQ_WAIT_INST_LDS') if c in v}
    print(k, json.dumps(line))
json.dump(out, open('gpurun_out/pmc/sq_split_gemms.json','w'), indent=1)
PY
