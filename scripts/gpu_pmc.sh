#!/bin/bash
# rocprofv3 on the bench legs themselves (distinct random rows -- the tiled corpus of tools/abi_probe is one big duplicate
# class and no longer representative): kernel-trace stats, then one --pmc pass per counter set (never combined with
# other trace domains).  The encode leg is traced single-stream (ANCE_ENCODER_STREAMS=1) so that a kernel's duration is
# its own.  Results: gpurun_out/pmc/**, summary -> gpurun_out/pmc/pmc_traffic.json (copy to profiles/).
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
# one step, no fp32-mode sample, no second search leg, and counters only for the kernels a roofline is reported for: a pass
# of the encode leg (~6 k dispatches) otherwise does not finish inside the 900 s limit
S="python bench.py --skip-encode --no-cpu-baseline --skip-encoder-like --steps ${PMC_STEPS:-1} --warmup 1"
# legs: search | encode_split (the headline arithmetic since round 5) | encode (= the fp16 fast mode) | encode_fp32 (kernel trace only)
for what in ${PMC_LEGS:-search encode_split}; do
  cmd="$S"
  if [ $what = encode ]; then cmd="python scripts/encode_mode_leg.py fp16"; export ANCE_ENCODER_STREAMS=1; fi
  if [ $what = encode_split ]; then cmd="python scripts/encode_mode_leg.py split"; export ANCE_ENCODER_STREAMS=1; fi
  if [ $what = encode_fp32 ]; then cmd="python scripts/encode_mode_leg.py fp32 1 4096"; export ANCE_ENCODER_STREAMS=1; fi
  rx="ip_topk_fast_kernel|rescore_kernel"; [ $what = encode ] && rx="gemm256_f16_desc_kernel|attention_kernel"
  [ $what = encode_split ] && rx="gemm256_split_kernel|gemm256_split_stream_kernel|attention_split_kernel"
  echo "== kernel-trace $what"
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pmc/kt_$what -o kt -- $cmd > gpurun_out/pmc/kt_$what.log 2>&1; echo "rc=$?"
  # the audit path and (PMC_TRACE_ONLY) the fp16 fast mode: kernel trace only
  if [ $what = encode_fp32 ] || [[ " ${PMC_TRACE_ONLY:-} " == *" $what "* ]]; then unset ANCE_ENCODER_STREAMS; continue; fi
  echo "== pmc cycles $what (GRBM_GUI_ACTIVE = shader clocks of the dispatch: clock-independent cost, and the clock itself)"
  timeout 900 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --kernel-include-regex "$rx" --output-format csv -d gpurun_out/pmc/CYCLES_$what -o pmc -- $cmd > gpurun_out/pmc/CYCLES_$what.log 2>&1; echo "rc=$?"
  echo "== pmc L2 hit/miss $what"
  timeout 900 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --kernel-include-regex "$rx" --output-format csv -d gpurun_out/pmc/L2_$what -o pmc -- $cmd > gpurun_out/pmc/L2_$what.log 2>&1; echo "rc=$?"
  for c in FETCH_SIZE WRITE_SIZE; do
    echo "== pmc $c $what"
    timeout 900 rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "$rx" --output-format csv -d gpurun_out/pmc/${c}_$what -o pmc -- $cmd > gpurun_out/pmc/${c}_$what.log 2>&1; echo "rc=$?"
    tail -2 gpurun_out/pmc/${c}_$what.log | cut -c1-200
  done
  unset ANCE_ENCODER_STREAMS
done
# ---- HBM traffic of the WHOLE encode step by counters (north_star: "rocprof evidences achieved HBM GB/s on the encode sweep"): every
# kernel of the leg, a 4,096-passage block so that a counter pass (~500 dispatches) finishes, the same block timed untraced
if [[ " ${PMC_LEGS:-search encode_split} " == *" encode_split "* ]]; then
  export ANCE_ENCODER_STREAMS=1
  EA="python bench.py --skip-search --no-cpu-baseline --skip-precise --skip-slice --skip-other-configs --encode-block 4096 --steps 1 --warmup 1"
  echo "== untraced step, 4,096-passage block"
  timeout 300 $EA > gpurun_out/pmc/encode_all_plain.json 2> gpurun_out/pmc/encode_all_plain.err; echo "rc=$?"
  for c in FETCH_SIZE WRITE_SIZE; do
    echo "== pmc $c encode_all"
    timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc/${c}_encodeall -o pmc -- $EA > gpurun_out/pmc/${c}_encodeall.log 2>&1; echo "rc=$?"
  done
  unset ANCE_ENCODER_STREAMS
fi
# ---- SQ counters of the split encoder's kernels (one pass): where the waves wait, LDS bank conflicts against LDS activity
if [[ " ${PMC_LEGS:-search encode_split} " == *" encode_split "* ]]; then
  export ANCE_ENCODER_STREAMS=1
  echo "== pmc SQ encode_split"
  timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --kernel-include-regex "gemm256_split_kernel|gemm256_split_stream_kernel|attention_split_kernel" --output-format csv -d gpurun_out/pmc/SQ_encode_split -o pmc -- python scripts/encode_mode_leg.py split > gpurun_out/pmc/SQ_encode_split.log 2>&1; echo "rc=$?"
  unset ANCE_ENCODER_STREAMS
  python - <<'PY'
import csv, glob, collections, re, json
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pmc/SQ_encode_split/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r'(gemm256_split(?:_stream)?_kernel<\d+>|attention_split_kernel)', r['Kernel_Name'])
        if not m:
            continue
        if 'gemm' in m.group(1) and int(r.get('Grid_Size', 0) or 0) < 100000:
            continue  # the CLS-tail launches of the last layer
        agg[m.group(1)][r['Counter_Name']].append(float(r['Counter_Value']))
out = {}
for k, v in agg.items():
    line = {c: sum(x) / len(x) for c, x in v.items()}
    wc = line.get('SQ_WAVE_CYCLES', 0)
    if wc:
        line['frac_of_wave_cycles'] = {c: round(line[c] / wc, 3) for c in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_WAIT_INST_LDS') if c in line}
    if line.get('SQ_LDS_IDX_ACTIVE'):
        line['lds_bank_conflict_per_lds_active_cycle'] = round(line.get('SQ_LDS_BANK_CONFLICT', 0.0) / line['SQ_LDS_IDX_ACTIVE'], 3)
    out[k] = line
    print(k, json.dumps({c: (round(x) if isinstance(x, float) and x > 10 else x) for c, x in line.items()}))
json.dump(out, open('gpurun_out/pmc/sq_counters_split.json', 'w'), indent=1)
PY
fi
find gpurun_out/pmc -name "*kernel_trace.csv" -size +8M -delete
python scripts/summarize_pmc.py gpurun_out/pmc gpurun_out/pmc/pmc_traffic.json 2>&1 | tail -120
