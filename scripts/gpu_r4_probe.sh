#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_gemm.py -q -p no:cacheprovider -x 2>&1 | tail -3
for i in 1 2; do
  for pf in 0 256 512; do
    ANCE_GEMM_PREFETCH=$pf timeout 300 python bench.py --steps 5 --warmup 2 --skip-search --skip-precise --no-cpu-baseline > gpurun_out/bench_pf${pf}_$i.json 2> gpurun_out/bench_pf${pf}_$i.err
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/bench_pf*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        bk=d['roofline']['by_kernel']
        print(f, round(d['value']), {k:round(v['ms_per_launch']*1e3,1) for k,v in bk.items() if k.startswith('gemm')}, d.get('errors'))
    except Exception as e:
        print(f,'ERR',e)
PY
