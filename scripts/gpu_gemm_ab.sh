#!/bin/bash
# A/B of the encoder GEMM variants on the GPU box: parity tests per variant, then encode-only bench.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
for v in ${VARIANTS:-256reg 256glds}; do
  echo "== parity ANCE_GEMM=$v"
  rm -f gpurun_out/encoder_parity.jsonl
  ANCE_GEMM=$v timeout 900 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_e2e.py -m gpu -q --timeout 600 -p no:cacheprovider > gpurun_out/enc_$v.log 2>&1
  echo "rc=$?"; tail -4 gpurun_out/enc_$v.log; cp gpurun_out/encoder_parity.jsonl gpurun_out/encoder_parity_$v.jsonl 2>/dev/null
done
for v in ${BENCH_VARIANTS:-128 256reg 256glds}; do
  for mt in ${MAXTOK:-32768 65536}; do
    echo "== bench ANCE_GEMM=$v max_tokens=$mt"
    ANCE_GEMM=$v timeout 600 python bench.py --skip-search --no-cpu-baseline --steps 3 --warmup 1 --max-tokens $mt ${BENCH_ARGS:-} > gpurun_out/bench_${v}_${mt}.json 2> gpurun_out/bench_${v}_${mt}.err
    python - <<PY
import json
try:
    d = json.load(open("gpurun_out/bench_${v}_${mt}.json"))
    r = d.get("roofline", {})
    print("  pps %.0f  alg TF %.0f  allgemm TF %.0f" % (d["value"], d["encode"]["algorithmic_tflops"], r.get("all_gemm_tflops") or 0))
    for k, x in r.get("by_kernel", {}).items():
        print("    %-14s %8.1f us/launch  %s" % (k, 1e3 * x["ms_per_launch"], ("%.0f TF" % x["tflops"]) if x.get("tflops") else ""))
    if d.get("errors"): print("  ERRORS", d["errors"])
except Exception as e:
    print("  failed:", e)
PY
  done
done
