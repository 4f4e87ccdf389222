#!/bin/bash
# Round 6: search filter with the LDS-DMA offsets / corpus descriptor precomputed per K-tile (pipe256.h: PRECOMPUTE) -- parity suite, then
# same-box A/B of the search leg against the previous form (make variant NAME=noprep DEFS=-DANCE_FAST_NO_PRECOMPUTE); plus the
# diagnosis of the streaming RESLN regression (make variant NAME=epi32 DEFS=-DANCE_EPI32_IN_PER_TILE: the 32 x 32-pass epilogue inside
# the launch-per-tile kernel).
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_search.py -q -x -p no:cacheprovider > gpurun_out/t_search.log 2>&1; echo "search rc=$?"; tail -4 gpurun_out/t_search.log
rm -f gpurun_out/ab_search.jsonl gpurun_out/ab_epi32.jsonl
srch() {  # name lib
  ANCE_AMD_LIB=$2 timeout 600 python bench.py --steps 5 --warmup 2 --skip-encode --skip-slice --skip-other-configs --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read())['search']
r=d['roofline']
print(json.dumps({'variant': '$1', 'queries_per_sec': d['value'], 'ms_per_step': d['ms_per_step'], 'filter_ms': r['ms_per_launch'], 'filter_frac': r['frac'], 'rescore_ms': r['rescore_ms_per_launch'], 'exact': d['exact_check']['identical'], 'encoder_like_qps': d['encoder_like']['value'], 'encoder_like_filter_ms': d['encoder_like']['filter_ms_per_launch'], 'encoder_like_exact': d['encoder_like']['exact_check']['identical']}))" >> gpurun_out/ab_search.jsonl
}
enc() {  # name lib stream
  ANCE_AMD_LIB=$2 ANCE_GEMM_STREAM=$3 timeout 600 python bench.py --steps 5 --warmup 2 --skip-search --skip-precise --skip-slice --skip-other-configs --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']
print(json.dumps({'variant': '$1', 'passages_per_sec': d['value'], 'timing': r['timing'], 'by_kernel': {k: v['ms_per_launch'] for k, v in r['by_kernel'].items()}}))" >> gpurun_out/ab_epi32.jsonl
}
for rep in 1 2 3; do
  srch previous $PWD/ance_amd/libance_amd_noprep.so
  srch precompute ""
done
for rep in 1 2; do
  enc per_tile "" 0
  enc per_tile_epi32 $PWD/ance_amd/libance_amd_epi32.so 0
  enc stream_all "" 2
done
cat gpurun_out/ab_search.jsonl gpurun_out/ab_epi32.jsonl
