#!/bin/bash
# Round 3: (1) GEMM phase timeline on the encoder shapes; (2) CU-partitioned lanes (ANCE_CU_SPLIT) x lane count on the encode leg.
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r3split
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python scripts/gemm_timeline.py > $O/gemm_timeline.jsonl 2> $O/gemm_timeline.err; echo "timeline rc=$?"; cat $O/gemm_timeline.jsonl | cut -c1-900
run_bench() {
  name=$1; shift
  env "$@" timeout 600 python bench.py --skip-search --no-cpu-baseline --steps ${STEPS:-6} --warmup 2 > $O/bench_$name.json 2> $O/bench_$name.err
  rc=$?
  python - $O/bench_$name.json $name $rc <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("bench[%s] rc=%s passages/s %.0f  ms/step %.1f  %s" % (sys.argv[2], sys.argv[3], d["value"], d["ms_per_step"], d["roofline"]["timing"][-62:]))
except Exception as e:
    print("bench[%s] rc=%s (no line) %r" % (sys.argv[2], sys.argv[3], e))
PY
}
run_bench base2 A=1
run_bench split1_2 ANCE_CU_SPLIT=1
run_bench split2_2 ANCE_CU_SPLIT=2
run_bench split3_2 ANCE_CU_SPLIT=3
run_bench base3 ANCE_ENCODER_STREAMS=3
run_bench base4 ANCE_ENCODER_STREAMS=4
run_bench split1_4 ANCE_CU_SPLIT=1 ANCE_ENCODER_STREAMS=4
run_bench split2_4 ANCE_CU_SPLIT=2 ANCE_ENCODER_STREAMS=4
run_bench split3_4 ANCE_CU_SPLIT=3 ANCE_ENCODER_STREAMS=4
run_bench base2_again A=1
