#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
ANCE_GEMM_TILE128=1 timeout 600 python -m pytest tests/test_gpu_gemm.py -q -p no:cacheprovider -k "not split" 2>&1 | tail -4
ANCE_GEMM_TILE128=1 ANCE_LN_FOLD=0 timeout 600 python -m pytest tests/test_gpu_encoder.py -q -p no:cacheprovider -k "full_depth_against_oracle or long_sequences or firstp_golden or maxp_golden or bert_golden" 2>&1 | tail -4
for i in 1 2; do
  ANCE_LN_FOLD=0 timeout 300 python bench.py --steps 5 --warmup 2 --skip-search --skip-precise --no-cpu-baseline > gpurun_out/bench_t256_$i.json 2> gpurun_out/bench_t256_$i.err
  ANCE_LN_FOLD=0 ANCE_GEMM_TILE128=1 timeout 300 python bench.py --steps 5 --warmup 2 --skip-search --skip-precise --no-cpu-baseline > gpurun_out/bench_t128_$i.json 2> gpurun_out/bench_t128_$i.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/bench_t*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        bk=d['roofline']['by_kernel']
        print(f, round(d['value']), {k:round(v['ms_per_launch']*1e3,1) for k,v in bk.items() if k.startswith('gemm') or k in ('layernorm','attention')}, d.get('errors'))
    except Exception as e:
        print(f,'ERR',e, open(f.replace('.json','.err')).read()[-300:])
PY
