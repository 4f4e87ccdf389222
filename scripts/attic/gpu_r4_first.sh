#!/bin/bash
# round 4, first GPU call: new encoder tests (split mode, wide-mean second pass, 12-layer golden), config-1 job tests,
# encode bench with the N-split tile order on / off (same box A/B)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
rm -f gpurun_out/encoder_parity.jsonl gpurun_out/config1_agreement.json
timeout 1200 python -m pytest tests/test_gpu_encoder.py -q -p no:cacheprovider > gpurun_out/t_encoder.log 2>&1; echo "encoder rc=$?" | tee -a gpurun_out/summary.txt
tail -25 gpurun_out/t_encoder.log
timeout 900 python -m pytest tests/test_gpu_config1.py -q -p no:cacheprovider > gpurun_out/t_config1.log 2>&1; echo "config1 rc=$?" | tee -a gpurun_out/summary.txt
tail -25 gpurun_out/t_config1.log
for i in 1 2; do
  ANCE_GEMM_NSPLIT=0 timeout 300 python bench.py --steps 5 --warmup 2 --skip-search --skip-precise --no-cpu-baseline > gpurun_out/bench_nsplit0_$i.json 2> gpurun_out/bench_nsplit0_$i.err
  ANCE_GEMM_NSPLIT=1 timeout 300 python bench.py --steps 5 --warmup 2 --skip-search --skip-precise --no-cpu-baseline > gpurun_out/bench_nsplit1_$i.json 2> gpurun_out/bench_nsplit1_$i.err
done
timeout 600 python bench.py --steps 5 --warmup 2 --skip-search --no-cpu-baseline > gpurun_out/bench_modes.json 2> gpurun_out/bench_modes.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/bench_nsplit*.json'))+['gpurun_out/bench_modes.json']:
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        bk=d['roofline']['by_kernel']
        print(f, round(d['value']), {k:round(v['ms_per_launch']*1e3,1) for k,v in bk.items() if k.startswith('gemm')}, d.get('errors'))
        for m in ('encode_split','encode_fp32'):
            if m in d: print('  ',m, round(d[m]['value']), d[m]['max_abs_vs_default'], d[m]['roofline']['kernel'], round(d[m]['roofline']['algorithmic'],1), d[m].get('max_abs_vs_fp32_mode'))
    except Exception as e:
        print(f,'ERR',e)
PY
