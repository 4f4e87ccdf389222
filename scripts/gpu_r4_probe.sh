#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
rm -f gpurun_out/encoder_parity.jsonl gpurun_out/config1_agreement.json
timeout 1200 python -m pytest tests/test_gpu_encoder.py -q -p no:cacheprovider -k "split or golden" > gpurun_out/t_encoder.log 2>&1; echo "encoder rc=$?"; tail -8 gpurun_out/t_encoder.log
timeout 300 python scripts/split_probe3.py 2>&1 | grep -v amdgpu.ids | grep SPLIT
timeout 900 python -m pytest tests/test_gpu_config1.py -q -p no:cacheprovider -k "split" > gpurun_out/t_config1.log 2>&1; echo "config1 rc=$?"; tail -5 gpurun_out/t_config1.log
timeout 600 python bench.py --steps 5 --warmup 2 --skip-search --no-cpu-baseline > gpurun_out/bench_modes.json 2> gpurun_out/bench_modes.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_modes.json').read().strip().splitlines()[-1])
print(round(d['value']), d.get('errors'))
for m in ('encode_split','encode_fp32'):
    print(m, round(d[m]['value']), d[m]['ms_per_step'], d[m]['max_abs_vs_default'], d[m].get('max_abs_vs_fp32_mode'))
    for k,v in d[m]['roofline']['by_kernel'].items(): print('   ',k,v)
PY
