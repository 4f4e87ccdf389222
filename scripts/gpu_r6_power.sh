#!/bin/bash
# Round 6: is the split encoder at the chip's power budget?  rocm-smi samples (socket power, sclk, temperature) while the encode leg runs
# in each arithmetic mode and while the search leg runs; idle samples first.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/power
export TMPDIR=/tmp
sample() {  # tag seconds
  for i in $(seq 1 $2); do
    echo "== $1 $i $(date +%s.%N)"; rocm-smi --showpower --showclocks --showtemp --showperflevel 2>&1 | grep -E "Power|sclk|mclk|Temperature \(Sensor (edge|junction|memory)|Performance Level|W\b" | head -12
    sleep 0.5
  done
}
rocm-smi --showmaxpower 2>&1 | grep -iE "power|W" | head -4 > gpurun_out/power/max.txt
sample idle 4 > gpurun_out/power/idle.txt 2>&1
for mode in split fp16 fp32; do
  steps=70; [ $mode = fp32 ] && steps=12; [ $mode = fp16 ] && steps=140
  python scripts/encode_mode_leg.py $mode $steps 16384 > gpurun_out/power/leg_$mode.log 2>&1 &
  pid=$!
  sleep 14   # import + weights + warm-up; the leg runs ~35 s
  sample $mode 16 > gpurun_out/power/$mode.txt 2>&1
  wait $pid; tail -1 gpurun_out/power/leg_$mode.log
done
python - <<'PY'
import re, json, glob
out = {}
for f in sorted(glob.glob('gpurun_out/power/*.txt')):
    t = open(f).read()
    pw = [float(x) for x in re.findall(r'Power[^:\n]*:\s*([0-9.]+)', t)]
    sc = [int(x) for x in re.findall(r'sclk[^\n]*\((\d+)Mhz\)', t)]
    out[f.split('/')[-1][:-4]] = {'power_w': pw, 'sclk_mhz': sc}
    print(f, 'power', pw[:12], 'sclk', sc[:12])
json.dump(out, open('gpurun_out/power/power_samples.json', 'w'), indent=1)
PY
cat gpurun_out/power/max.txt; head -20 gpurun_out/power/split.txt
