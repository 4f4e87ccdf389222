#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/var
export TMPDIR=/tmp
export LD_LIBRARY_PATH=/opt/rocm/lib:${LD_LIBRARY_PATH:-}
for n in 1105228 2210456 4420912; do
  python bench.py --skip-encode --no-cpu-baseline --steps 3 --warmup 1 --n-passages $n 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); s=d['search']; print('n',s['rows_total'],'qps',round(s['value']),'ms/step',round(s['ms_per_step'],1),'kernel ms',round(s['roofline']['ms_per_launch'],1),'finalize',round(s['roofline']['finalize_ms_per_launch'],2))"
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/var/kt_small -o kt -- python bench.py --skip-encode --no-cpu-baseline --steps 3 --warmup 1 --n-passages 1105228 > gpurun_out/var/kt_small.log 2>&1
cat gpurun_out/var/kt_small/kt_kernel_stats.csv | cut -c1-160 | head -12
