#!/bin/bash
# Round 5, second GPU call: A/B of the polynomial GELU against the library erff (same-box, alternating), MFMA-busy / clock counters
# of the split GEMMs, the whole -m gpu suite (12-layer goldens of the other towers, chunked exchange on one GPU), the default bench
# line with the refresh slice.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/ab gpurun_out/pmc
export TMPDIR=/tmp
rm -f gpurun_out/encoder_parity.jsonl gpurun_out/ab_gelu.jsonl
for i in 1 2; do
  for lib in erff cur; do
    p=$PWD/ance_amd/libance_amd.so; [ $lib != cur ] && p=$PWD/ance_amd/libance_amd_$lib.so
    ANCE_AMD_LIB=$p timeout 600 python bench.py --skip-search --no-cpu-baseline --skip-precise --skip-slice --steps 6 --warmup 2 > gpurun_out/ab/g_${lib}_$i.json 2> gpurun_out/ab/g_${lib}_$i.err
    python -c "
import json; d=json.loads(open('gpurun_out/ab/g_${lib}_$i.json').read().strip().splitlines()[-1]); bk=d['roofline']['by_kernel']
print(json.dumps({'lib': '$lib', 'run': $i, 'passages_per_sec': d['value'], 'ms_per_step': d['ms_per_step'], 'isolated': d['roofline']['timing'][-90:], 'us_per_launch': {k: round(1e3*v['ms_per_launch'],1) for k,v in bk.items()}}))" | tee -a gpurun_out/ab_gelu.jsonl
  done
done
echo "== pmc cycles, split encode leg"
ANCE_ENCODER_STREAMS=1 timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --kernel-include-regex "gemm256_split_kernel|attention_split_kernel" --output-format csv -d gpurun_out/pmc/CYCLES_encode_split -o pmc -- python scripts/encode_mode_leg.py split > gpurun_out/pmc/CYCLES_encode_split.log 2>&1; echo "rc=$?"
python - <<'PY'
import csv, glob, collections
gui=collections.defaultdict(list); busy=collections.defaultdict(list); dur=collections.defaultdict(list)
for f in glob.glob('gpurun_out/pmc/CYCLES_encode_split/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0][-40:]
        if r['Counter_Name']=='GRBM_GUI_ACTIVE': gui[k].append(float(r['Counter_Value'])/8)
        if r['Counter_Name']=='SQ_VALU_MFMA_BUSY_CYCLES': busy[k].append(float(r['Counter_Value'])/1024)
for f in glob.glob('gpurun_out/pmc/CYCLES_encode_split/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0][-40:]
        dur[k].append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
for k in gui:
    g=sum(gui[k])/len(gui[k]); b=sum(busy[k])/max(len(busy[k]),1); t=sum(dur[k])/max(len(dur[k]),1)
    print('%-42s n=%4d cycles %.0f mfma_busy %.3f clock %.2f GHz avg %.1f us' % (k, len(gui[k]), g, b/g, g/t if t else 0, t/1e3))
PY
timeout 2400 python -m pytest tests/ -q -m gpu -p no:cacheprovider > gpurun_out/t_all.log 2>&1; echo "pytest -m gpu rc=$?"; tail -12 gpurun_out/t_all.log
echo "== bench (driver flags)"
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "rc=$?"; tail -3 gpurun_out/bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench.log').read().strip().splitlines()[-1])
print({k: d.get(k) for k in ('metric','value','ms_per_step','dtype','errors')})
print('roofline', {k: d['roofline'].get(k) for k in ('kernel','achieved','frac','frac_executed','all_gemm_tflops')})
for k in ('encode_fp16_fast','encode_fp32'):
    if k in d: print(k, d[k]['value'], d[k]['ms_per_step'], d[k].get('max_abs_vs_split'))
print('search', d.get('search',{}).get('value'))
print('slice', d.get('full_refresh_slice'))
print('cpu', d.get('cpu_baseline',{}).get('value'))
PY
