#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r3attn2
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_encoder.py -m gpu -q --timeout 600 -p no:cacheprovider > $O/test_default.log 2>&1; echo "tests rc=$?"; tail -4 $O/test_default.log
run_bench() {
  name=$1; shift
  env "$@" timeout 600 python bench.py --skip-search --no-cpu-baseline --steps ${STEPS:-6} --warmup 2 > $O/bench_$name.json 2> $O/bench_$name.err
  rc=$?
  python - $O/bench_$name.json $name $rc <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    bk = d["roofline"]["by_kernel"]
    print("bench[%s] rc=%s passages/s %.0f  ms/step %.1f  %s" % (sys.argv[2], sys.argv[3], d["value"], d["ms_per_step"], d["roofline"]["timing"][-62:]))
    print("  " + "  ".join("%s %.1f" % (k, 1e3 * v["ms_per_launch"]) for k, v in bk.items()))
except Exception as e:
    print("bench[%s] rc=%s (no line) %r" % (sys.argv[2], sys.argv[3], e))
PY
}
run_bench default A=1
run_bench no_coal ANCE_ATTN_COAL=0
run_bench reg ANCE_ATTN_REG=1
run_bench default2 A=1
