#!/bin/bash
# split-mode attention: parity subset + kernel-trace of one split encode leg (single stream) + untraced timing
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_encoder.py -q -m gpu -p no:cacheprovider -k "split" 2>&1 | tail -3
ANCE_ENCODER_STREAMS=1 timeout 300 python scripts/encode_mode_leg.py split 5 2>&1 | tail -1
rm -rf gpurun_out/attn_trace
ANCE_ENCODER_STREAMS=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/attn_trace -o t -- python scripts/encode_mode_leg.py split 2 > gpurun_out/attn_trace.log 2>&1
f=$(find gpurun_out/attn_trace -name "*kernel_stats.csv" | head -1)
python - "$f" <<'P'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:6]:
    print("%-60s calls %5s avg %8.1f us %5s%%"%(r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
P
echo "--- buckets off"
ANCE_SPLIT_ATTN_BUCKETS=0 ANCE_ENCODER_STREAMS=1 timeout 300 python scripts/encode_mode_leg.py split 5 2>&1 | tail -1
ANCE_SPLIT_ATTN_BUCKETS=0 timeout 300 python scripts/encode_mode_leg.py split 5 2>&1 | tail -1
echo "--- buckets on, two lanes"
timeout 300 python scripts/encode_mode_leg.py split 5 2>&1 | tail -1
