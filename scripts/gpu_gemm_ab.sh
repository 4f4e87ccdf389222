export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_encoder.py tests/test_gpu_e2e.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -3
for v in 1 0 1 0; do
  ANCE_GEMM_DESC=$v timeout 300 python bench.py --skip-search --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); b=d['roofline']['by_kernel']
print('desc=$v pps', round(d['value']), 'all_gemm_tf', round(d['roofline']['all_gemm_tflops']), {k:round(1e3*v['ms_per_launch'],1) for k,v in b.items() if k.startswith('gemm')})"
done
