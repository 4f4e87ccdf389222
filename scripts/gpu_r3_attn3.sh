#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r3attn3
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
CMD="python bench.py --skip-search --no-cpu-baseline --steps 1 --warmup 1"
pass() {
  tag=$1; shift
  ANCE_ENCODER_STREAMS=1 timeout 300 rocprofv3 --pmc "$@" --kernel-trace --kernel-include-regex "attention" --output-format csv -d $O/$tag -o pmc -- $CMD > $O/$tag.log 2>&1
  echo "pass[$tag] rc=$?"
}



find $O -name "*kernel_trace.csv" -size +4M -delete
python - <<'PY'
import csv, glob, os, collections
O = "gpurun_out/r3attn3"
for d in sorted(glob.glob(O + "/*/")):
    tag = os.path.basename(d.rstrip("/"))
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            a = acc[(r.get("Kernel_Name", "")[:60], r["Counter_Name"])]
            a[0] += float(r["Counter_Value"]); a[1] += 1
    for (k, c), (v, n) in sorted(acc.items()):
        print("%-6s %-36s per-launch %.4g  (n=%d)" % (tag, c, v / max(n, 1), n))
PY
