#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== parity"
rm -f gpurun_out/encoder_parity.jsonl
timeout 900 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_e2e.py tests/test_gpu_gemm.py -m gpu -q --timeout 600 -p no:cacheprovider > gpurun_out/enc.log 2>&1; echo "rc=$?"; tail -4 gpurun_out/enc.log
for ns in 2 1; do
  echo "== bench ANCE_ENCODER_STREAMS=$ns"
  ANCE_ENCODER_STREAMS=$ns timeout 600 python bench.py --skip-search --no-cpu-baseline --steps 3 --warmup 1 ${BENCH_ARGS:-} > gpurun_out/bench_ns$ns.json 2> gpurun_out/bench_ns$ns.err
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/bench_ns$ns.json"))
    r = d.get("roofline", {})
    print("  pps %.0f  ms/step %.1f  alg TF %.0f  allgemm TF %.0f" % (d["value"], d["ms_per_step"], d["encode"]["algorithmic_tflops"], r.get("all_gemm_tflops") or 0))
    tot = 0
    for k, x in r.get("by_kernel", {}).items():
        tot += x["total_ms"]
        print("    %-14s %8.1f us/launch n=%5d tot %8.1f ms %s" % (k, 1e3 * x["ms_per_launch"], x["launches"], x["total_ms"], ("%.0f TF" % x["tflops"]) if x.get("tflops") else ""))
    print("    sum of kernel ms / steps = %.1f" % (tot / d["steps"]))
    if d.get("errors"): print("  ERRORS", d["errors"])
except Exception as e:
    print("  failed:", e); print(open("gpurun_out/bench_ns$ns.err").read()[-2000:])
PY
done
