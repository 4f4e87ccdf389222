#!/bin/bash
# Round 6: persistent streaming split GEMM -- bit-equality with the launch-per-tile kernel, encoder parity, same-box A/B of the encode leg:
#   stream 0 = launch-per-tile kernel (ANCE_GEMM_STREAM=0), 1 = streaming (product), tight = streaming with steady-state waits on
#   K-tile 0 (make variant NAME=tight DEFS=-DANCE_STREAM_LOOSE_FIRST=0)
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== tr16 probe"; tools/tr16_probe > gpurun_out/tr16_probe.txt 2>&1; echo "probe rc=$?"; tail -2 gpurun_out/tr16_probe.txt
timeout 600 python -m pytest tests/test_gpu_gemm.py -q -x -p no:cacheprovider -k "streaming or split" > gpurun_out/t_gemm.log 2>&1; echo "gemm rc=$?"; tail -5 gpurun_out/t_gemm.log
timeout 900 python -m pytest tests/test_gpu_encoder.py -q -x -p no:cacheprovider -k "split or golden or default or large_micro" > gpurun_out/t_enc.log 2>&1; echo "enc rc=$?"; tail -5 gpurun_out/t_enc.log
rm -f gpurun_out/ab_stream.jsonl
one() {  # name stream lanes lib
  ANCE_AMD_LIB=$4 ANCE_GEMM_STREAM=$2 ANCE_ENCODER_STREAMS=$3 timeout 600 python bench.py --steps 5 --warmup 2 --skip-search --skip-precise --skip-slice --skip-other-configs --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']
print(json.dumps({'variant': '$1', 'lanes': $3, 'passages_per_sec': d['value'], 'ms_per_step': d['ms_per_step'], 'timing': r['timing'], 'by_kernel': {k: v['ms_per_launch'] for k, v in r['by_kernel'].items()}, 'all_gemm_tflops': r['all_gemm_tflops']}))" >> gpurun_out/ab_stream.jsonl
}
for rep in 1 2; do
  one per_tile 0 2 ""
  one stream 1 2 ""
  one stream 1 1 ""
  [ -f ance_amd/libance_amd_tight.so ] && one stream_tight 1 2 $PWD/ance_amd/libance_amd_tight.so
done
cat gpurun_out/ab_stream.jsonl
