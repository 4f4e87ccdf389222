#!/bin/bash
# RESLN epilogue with 16-byte accesses: GEMM / encoder parity, step time, kernel trace (gemm<4> against gemm<6> of the same box)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_encoder.py -q -m gpu -p no:cacheprovider -x 2>&1 | tail -3
for rep in 1 2; do timeout 300 python scripts/encode_mode_leg.py fp16 8 2>&1 | tail -1; done
rm -rf gpurun_out/resln
ANCE_ENCODER_STREAMS=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/resln -o t -- python scripts/encode_mode_leg.py fp16 2 > gpurun_out/resln.log 2>&1
python - gpurun_out/resln/t_kernel_stats.csv <<'P'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:5]:
    print("%-60s calls %5s avg %8.1f us %5s%%"%(r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
P
find gpurun_out/resln -name "*kernel_trace.csv" -delete
