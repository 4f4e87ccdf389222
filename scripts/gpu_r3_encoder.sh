#!/bin/bash
# Round 3, encoder work: parity of the new default path (register-resident attention, folded LayerNorm, MFMA head), then
# the same parity tests with each piece switched back (isolates a broken piece in one call), then the encode leg of the
# bench per configuration.  Logs under gpurun_out/r3enc/.
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r3enc
mkdir -p $O
export TMPDIR=/tmp
run_tests() {  # name, env...
  name=$1; shift
  env "$@" timeout 900 python -m pytest tests/test_gpu_encoder.py -m gpu -q --timeout 600 -p no:cacheprovider > $O/test_$name.log 2>&1
  echo "tests[$name] rc=$?"; tail -4 $O/test_$name.log
  [ -f gpurun_out/encoder_parity.jsonl ] && mv gpurun_out/encoder_parity.jsonl $O/parity_$name.jsonl
}
run_bench() {
  name=$1; shift
  env "$@" timeout 600 python bench.py --skip-search --no-cpu-baseline --steps ${STEPS:-6} --warmup 2 > $O/bench_$name.json 2> $O/bench_$name.err
  echo "bench[$name] rc=$?"
  python - $O/bench_$name.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    bk = d["roofline"]["by_kernel"]
    print("  passages/s %.0f  ms/step %.1f  iso: %s" % (d["value"], d["ms_per_step"], d["roofline"]["timing"][-60:]))
    print("  " + "  ".join("%s %.1f" % (k, 1e3 * v["ms_per_launch"]) for k, v in bk.items()))
    print("  all-gemm TF %.0f" % d["roofline"]["all_gemm_tflops"])
except Exception as e:
    print("  (no line)", e)
PY
}
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
run_tests default A=1
run_tests no_attn_reg ANCE_ATTN_REG=0
run_tests no_fold ANCE_LN_FOLD=0
run_tests no_head ANCE_HEAD_MFMA=0
run_tests all_old ANCE_ATTN_REG=0 ANCE_LN_FOLD=0 ANCE_HEAD_MFMA=0
run_bench default A=1
run_bench all_old ANCE_ATTN_REG=0 ANCE_LN_FOLD=0 ANCE_HEAD_MFMA=0
run_bench no_attn_reg ANCE_ATTN_REG=0
run_bench no_fold ANCE_LN_FOLD=0
run_bench no_head ANCE_HEAD_MFMA=0
